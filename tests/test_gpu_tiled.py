"""GPU parity tests of the LDS-tiled encoder-shape MSDA kernel (csrc/msda_tile.hip, bf16, Nq == S, L == 4), called
through the C ABI (rdetr_msda_forward[_fused]_tiled_bf16) -- against the C oracle on bf16-rounded value and against the
direct query-run kernel on the same inputs.  The kernel copies, per 16x16 query tile and level, the window of the value
plane the tile samples into LDS; samples outside the window are fetched from global memory.  The cases below therefore
sweep the offset spread from "everything inside the window" to "nothing inside" -- the result must not depend on it.

Tolerance: |err| <= 2^-8 |ref| + 1e-3 vs the fp32 oracle (one bf16 output rounding, fp32 accumulation; the reference op
has no bf16 -- SURVEY.md Appendix B item 12); vs the direct kernel 2^-7 |ref| + 1e-3 (two independently rounded outputs).
"""
import numpy as np
import pytest
import torch

from helpers import pyramid

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
R50 = [(100, 168), (50, 84), (25, 42), (13, 21)]


@pytest.fixture(scope="module")
def ops():
    from relation_detr_amd import _lib, ops
    _lib.load()
    return ops


def _pixel_refs(shapes):
    refs = []
    for h, w in shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    return torch.cat(refs, 0)


def _encoder_inputs(shapes, B, spread_px, seed, scatter=0.0, poison=False):
    shp, start, S = pyramid(shapes)
    L = len(shapes)
    g = torch.Generator().manual_seed(seed)
    value = torch.randn(B, S, 8, 32, generator=g).to(torch.bfloat16)
    wh = shp.flip(-1).float()
    k = torch.arange(1, 5, dtype=torch.float32).view(1, 1, 1, 1, 4, 1) / 4.0
    off = torch.randn(B, S, 8, L, 4, 2, generator=g) * k * spread_px / wh.view(1, 1, 1, L, 1, 2)
    loc = _pixel_refs(shapes)[None, :, None, None, None, :] + off
    if scatter > 0:           # a fraction of the queries samples anywhere (and beyond the border), like decoder queries
        pick = torch.rand(B, S, 1, 1, 1, 1, generator=g) < scatter
        loc = torch.where(pick, torch.rand(B, S, 8, L, 4, 2, generator=g) * 1.4 - 0.2, loc)
    if poison:                # NaN / huge / exactly-on-the-border locations
        loc[0, 5, 0, 0, 0, 0] = float("nan")
        loc[0, 7, 1, 1, 2, 1] = 1e9
        loc[0, 9, 2, 2, 1, :] = 0.0
        loc[0, 11, 3, 3, 3, :] = 1.0
        loc[0, 13, 4, 0, 0, 0] = -1e-7
    attn = torch.softmax(torch.randn(B, S, 8, L * 4, generator=g), -1).view(B, S, 8, L, 4)
    return value, shp, start, loc.contiguous(), attn.contiguous(), S, L


def _check(out, ref, direct=None):
    assert np.isfinite(out).all()
    bad = np.abs(out - ref) > 2.0 ** -8 * np.abs(ref) + 1e-3
    assert not bad.any(), f"{bad.sum()} outputs off, max err {np.abs(out - ref).max()} at rows {np.unique(np.argwhere(bad)[:, 1])[:8]}"
    if direct is not None:
        assert (np.abs(out - direct) <= 2.0 ** -7 * np.abs(ref) + 1e-3).all()


@pytest.mark.parametrize("shapes,B,spread_px,scatter,poison", [
    (R50, 1, 3.0, 0.0, False),                                   # every sample inside its tile's window
    (R50, 2, 4.0, 0.0, True),                                    # BASELINE spread (sigma up to 4 px) + NaN / border cases
    (R50, 1, 30.0, 0.0, False),                                  # windows clipped: most fine-level samples come from global memory
    ([(64, 96), (32, 48), (16, 24), (8, 12)], 2, 4.0, 0.05, False),      # multiples of the tile; 5 % scattered queries
    ([(75, 61), (38, 31), (19, 16), (10, 8)], 3, 6.0, 0.0, True),        # ragged tiles on every level
    ([(70, 70), (35, 35), (18, 18), (9, 9)], 1, 2.0, 1.0, False),        # all queries scattered: no locality at all
])
def test_tiled_matches_oracle_and_direct(ops, shapes, B, spread_px, scatter, poison):
    from oracle import c_oracle
    value, shp, start, loc, attn, S, L = _encoder_inputs(shapes, B, spread_px, seed=int(spread_px * 7) + B, scatter=scatter,
                                                         poison=poison)
    dev = (value.to(DEV), shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    out = ops.ms_deform_attn_forward_strategy("tiled", *dev).float().cpu().numpy()
    direct = ops.ms_deform_attn_forward_strategy("direct", *dev).float().cpu().numpy()
    ref = c_oracle.msda_forward(value.float().numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
    _check(out, ref, direct)
    # the plain operator (direct kernel unless RDETR_MSDA_ALGO=lds) agrees with the explicit direct entry point bit for bit
    auto = ops.ms_deform_attn_forward(*dev, 64).float().cpu().numpy()
    assert np.array_equal(auto, direct) or np.array_equal(auto, out)


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_tiled_fused_producer(ops, ref_dim):
    """raw offsets / logits + reference points in, softmax and location arithmetic inside the kernel
    (ms_deform_attn.py:326-349): against the oracle's materialised sequence."""
    from oracle import torch_ref
    shapes = [(72, 100), (36, 50), (18, 25), (9, 13)]
    shp, start, S = pyramid(shapes)
    g = torch.Generator().manual_seed(40 + ref_dim)
    B, L = 2, 4
    value = torch.randn(B, S, 8, 32, generator=g).to(torch.bfloat16)
    off = (torch.randn(B, S, 8, L, 4, 2, generator=g) * 3).to(torch.bfloat16)
    logits = (torch.randn(B, S, 8, L * 4, generator=g) * 2).to(torch.bfloat16)
    ref = _pixel_refs(shapes)[None, :, None, :].expand(B, S, L, 2)
    if ref_dim == 4:
        ref = torch.cat([ref, torch.rand(B, S, L, 2, generator=g) * 0.2 + 0.02], -1)
    ref = ref.contiguous()
    dev = (value.to(DEV), shp.to(DEV), start.to(DEV), off.to(DEV), logits.to(DEV), ref.to(DEV))
    out = ops.ms_deform_attn_forward_strategy("tiled", *dev).float().cpu().numpy()
    direct = ops.ms_deform_attn_forward_strategy("direct", *dev).float().cpu().numpy()
    loc = torch_ref.sampling_locations_from_reference(ref, off.float(), shp, 4)
    w = logits.float().softmax(-1).view(B, S, 8, L, 4)
    expect = torch_ref.msda_core(value.float(), shp, loc, w).numpy()
    _check(out, expect, direct)
    auto = ops.ms_deform_attn_forward_fused(*dev).float().cpu().numpy()
    assert np.array_equal(auto, direct) or np.array_equal(auto, out)


def test_tiled_unsupported_shapes(ops):
    from relation_detr_amd import _lib
    # five levels
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)], 1, 2.0, 1)
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward_strategy("tiled", v.to(DEV), shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    # Nq != S
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 96), (32, 48), (16, 24), (8, 12)], 1, 2.0, 2)
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward_strategy("tiled", v.to(DEV), shp.to(DEV), start.to(DEV), loc[:, :900].contiguous().to(DEV),
                                            attn[:, :900].contiguous().to(DEV))
    # ... while the plain operator serves both through the direct kernel
    out = ops.ms_deform_attn_forward(v.to(DEV), shp.to(DEV), start.to(DEV), loc[:, :900].contiguous().to(DEV),
                                     attn[:, :900].contiguous().to(DEV), 64)
    assert out.shape == (1, 900, 256)


def test_tiled_full_size_properties(ops):
    """BASELINE.json configs[1] size (B = 4, S = Nq = 22,323): a constant value map returns the constant wherever all
    samples fall inside the levels (weights sum to one), and two launches on the same inputs are bit-identical (no
    atomics, fixed summation order)."""
    value, shp, start, loc, attn, S, L = _encoder_inputs(R50, 4, 4.0, seed=11)
    wh = shp.flip(-1).float().view(1, 1, 1, L, 1, 2)
    loc = torch.minimum(torch.maximum(loc, 1.0 / wh), 1.0 - 1.0 / wh).contiguous()
    dev = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    const = (torch.arange(256, dtype=torch.float32).view(1, 1, 8, 32) / 64).expand(4, S, 8, 32).contiguous().to(torch.bfloat16)
    oc = ops.ms_deform_attn_forward_strategy("tiled", const.to(DEV), *dev).float().cpu()
    assert (oc - const[:, :1].reshape(4, 1, 256).float()).abs().max().item() <= 2.0 ** -7 * 4
    v = value.to(DEV)
    o1 = ops.ms_deform_attn_forward_strategy("tiled", v, *dev)
    o2 = ops.ms_deform_attn_forward_strategy("tiled", v, *dev)
    assert torch.equal(o1, o2)
    d = ops.ms_deform_attn_forward_strategy("direct", v, *dev).float()
    assert ((o1.float() - d).abs() <= 2.0 ** -7 * d.abs() + 1e-3).all()
