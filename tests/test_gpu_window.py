"""GPU parity tests of the LDS-window encoder-shape MSDA kernel (csrc/msda_win.hip, bf16, Nq == S, L == 4), called
through the C ABI (rdetr_msda_forward[_fused]_opt_bf16 with algo = RDETR_MSDA_WINDOW) in BOTH value layouts ([B,S,H,D], the
reference operator's, and head-major [B,H,S,D] as rdetr_value_to_head_major_bf16 writes it) -- against the C oracle on
bf16-rounded value and against the direct query-run kernel on the same inputs.  The kernel copies, per 16x12 query tile
and level, a window of the value plane into LDS (pixels outside the level arrive as zeros through the range-checked
DMA); samples outside the window are fetched from global memory into a patch.  The cases below therefore sweep the
offset spread from "everything inside the window" to "nothing inside" -- the result must not depend on it -- and put
samples on / beyond every border.

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


def _head_major(value):
    """[B,S,8,32] -> [B,8,S,32] by torch (the layout under test, not the repack kernel)."""
    return value.permute(0, 2, 1, 3).contiguous()


def _check(out, ref, direct=None):
    assert np.isfinite(out).all()
    bad = np.abs(out - ref) > 2.0 ** -8 * np.abs(ref) + 1e-3
    assert not bad.any(), f"{bad.sum()} outputs off, max err {np.abs(out - ref).max()} at rows {np.unique(np.argwhere(bad)[:, 1])[:8]}"
    if direct is not None:
        assert (np.abs(out - direct) <= 2.0 ** -7 * np.abs(ref) + 1e-3).all()


@pytest.mark.parametrize("shapes,B,spread_px,scatter,poison", [
    (R50, 1, 3.0, 0.0, False),                                   # (nearly) every sample inside its tile's window
    (R50, 2, 4.0, 0.0, True),                                    # BASELINE spread (sigma up to 4 px) + NaN / border cases
    (R50, 1, 30.0, 0.0, False),                                  # most fine-level samples come from global memory (> 4 per wave: extra steps)
    ([(64, 96), (32, 48), (16, 24), (8, 12)], 2, 4.0, 0.05, False),      # multiples of the tile; 5 % scattered queries
    ([(75, 61), (38, 31), (19, 16), (10, 8)], 3, 6.0, 0.0, True),        # ragged tiles on every level
    ([(70, 70), (35, 35), (18, 18), (9, 9)], 1, 2.0, 1.0, False),        # all queries scattered: no locality at all
    ([(160, 24), (80, 12), (40, 6), (20, 3)], 2, 3.0, 0.02, True),        # levels narrower than a window: zero columns on both sides
])
def test_window_matches_oracle_and_direct(ops, shapes, B, spread_px, scatter, poison):
    from oracle import c_oracle
    value, shp, start, loc, attn, S, L = _encoder_inputs(shapes, B, spread_px, seed=int(spread_px * 7) + B, scatter=scatter,
                                                         poison=poison)
    rest = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    v = value.to(DEV)
    ref = c_oracle.msda_forward(value.float().numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
    direct = ops.ms_deform_attn_forward(v, *rest, algo="direct").float().cpu().numpy()
    out = ops.ms_deform_attn_forward(v, *rest, algo="window").float().cpu().numpy()
    _check(out, ref, direct)
    # head-major value: both kernels
    vh = _head_major(v)
    out_h = ops.ms_deform_attn_forward(vh, *rest, value_layout="bhsd", algo="window").float().cpu().numpy()
    assert np.array_equal(out_h, out)                     # same arithmetic, only the fill addresses differ
    direct_h = ops.ms_deform_attn_forward(vh, *rest, value_layout="bhsd", algo="direct").float().cpu().numpy()
    assert np.array_equal(direct_h, direct)
    # the plain operator picks one of the two
    auto = ops.ms_deform_attn_forward(v, *rest, 64).float().cpu().numpy()
    assert np.array_equal(auto, direct) or np.array_equal(auto, out)


@pytest.mark.parametrize("case", ["gap", "unordered", "short", "level1_larger"])
def test_auto_never_takes_the_window_kernel_on_levels_that_do_not_tile_the_value(ops, case):
    """The window kernel enumerates its queries as the pixels of the levels, so it is only correct when the levels tile [0, S)
    (include/relation_detr_amd.h, RDETR_MSDA_WINDOW precondition).  The reference operator accepts any table whose levels fit
    (ms_deform_im2col_cuda.cuh:263-267 only indexes with level_start_index): gaps between levels, levels stored out of order,
    a value tensor longer than the levels, a level larger than level 0.  With Nq == S >= 4096, L == 4, bf16 -- the shape where
    'auto' would otherwise pick the window kernel -- the result must still be the oracle's, every output row written."""
    from oracle import c_oracle
    from relation_detr_amd import _lib
    shapes = [(64, 96), (32, 48), (16, 24), (8, 12)]
    areas = [h * w for h, w in shapes]
    if case == "gap":                                    # 100 unused positions between level 0 and level 1
        starts = [0, areas[0] + 100, areas[0] + 100 + areas[1], areas[0] + 100 + areas[1] + areas[2]]
        S = starts[3] + areas[3]
    elif case == "unordered":                            # levels stored coarsest first
        starts = [areas[3] + areas[2] + areas[1], areas[3] + areas[2], areas[3], 0]
        S = sum(areas)
    elif case == "short":                                # value longer than the levels
        starts = [0, areas[0], areas[0] + areas[1], areas[0] + areas[1] + areas[2]]
        S = sum(areas) + 333
    else:                                                # cumulative and complete, but level 1 outgrows level 0
        shapes = [(32, 48), (64, 96), (16, 24), (8, 12)]
        areas = [h * w for h, w in shapes]
        starts = [0, areas[0], areas[0] + areas[1], areas[0] + areas[1] + areas[2]]
        S = sum(areas)
    shp = torch.tensor(shapes, dtype=torch.int64)
    start = torch.tensor(starts, dtype=torch.int64)
    assert not ops.levels_window_ok(shp.to(DEV), start.to(DEV), S)
    g = torch.Generator().manual_seed(len(case))
    B, L = 2, 4
    value = torch.randn(B, S, 8, 32, generator=g).to(torch.bfloat16)
    loc = torch.rand(B, S, 8, L, 4, 2, generator=g) * 1.2 - 0.1
    attn = torch.softmax(torch.randn(B, S, 8, L * 4, generator=g), -1).view(B, S, 8, L, 4)
    ref = c_oracle.msda_forward(value.float().numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
    rest = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    for layout, v in (("bshd", value.to(DEV)), ("bhsd", _head_major(value.to(DEV)))):
        out = ops.ms_deform_attn_forward(v, *rest, value_layout=layout).float().cpu().numpy()        # algo = "auto"
        _check(out, ref)
        with pytest.raises(_lib.RdetrError):
            ops.ms_deform_attn_forward(v, *rest, value_layout=layout, algo="window")
    # the plain C entry point (the `_C` contract, no algo argument) is the direct kernel: safe for any table
    lib = _lib.load()
    v = value.to(DEV)
    out = torch.full((B, S, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
    st = lib.rdetr_msda_forward_bf16(v.data_ptr(), rest[0].data_ptr(), rest[1].data_ptr(), rest[2].data_ptr(), rest[3].data_ptr(),
                                     B, S, 8, 32, L, S, 4, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert st == 0
    _check(out.float().cpu().numpy(), ref)


def test_value_to_head_major(ops):
    g = torch.Generator().manual_seed(3)
    B, S = 3, 1000 + 37
    wide = torch.randn(B, S, 3 * 256, generator=g).to(torch.bfloat16).to(DEV)
    v = wide[..., 256:512]                                                   # a column slice: strided rows
    mask = (torch.rand(B, S, generator=g) < 0.2).to(DEV)
    got = ops.value_to_head_major(v, mask)
    want = v.masked_fill(mask[..., None], 0).view(B, S, 8, 32).permute(0, 2, 1, 3)
    assert got.shape == (B, 8, S, 32) and torch.equal(got, want)
    assert torch.equal(ops.value_to_head_major(v.contiguous()), v.reshape(B, S, 8, 32).permute(0, 2, 1, 3))


def test_value_projection_into_head_major(ops):
    """rdetr_linear_k256_hm_bf16: value_proj + padding zero-fill + re-layout in one kernel == the hand-written projection followed
    by the re-layout kernel, bit for bit; within one bf16 rounding of the fp32 product."""
    g = torch.Generator().manual_seed(8)
    B, S = 2, 5000 + 13
    wide = torch.randn(B, S, 7 * 256, generator=g).to(torch.bfloat16).to(DEV)
    x = wide[..., 256:512]                                                   # the encoder's column-slice input
    w = (torch.randn(256, 256, generator=g) * 0.06).to(torch.bfloat16).to(DEV)
    b = (torch.randn(256, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    mask = (torch.rand(B, S, generator=g) < 0.15).to(DEV)
    got = ops.value_proj_head_major(x, w, b, mask)
    two_step = ops.value_to_head_major(ops.linear_k256(x, w, b), mask)
    assert got.shape == (B, 8, S, 32) and torch.equal(got, two_step)
    exact = torch.nn.functional.linear(x.float(), w.float(), b.float()).masked_fill(mask[..., None], 0)
    exact = exact.view(B, S, 8, 32).permute(0, 2, 1, 3)                    # fp32 GEMM of the same bf16 operands
    assert ((got.float() - exact).abs() <= 2.0 ** -8 * exact.abs() + 1e-3).all()      # one bf16 rounding of the result
    assert torch.equal(ops.value_proj_head_major(x, w, b), ops.value_to_head_major(ops.linear_k256(x, w, b)))


@pytest.mark.parametrize("ref_dim,strided", [(2, False), (4, False), (2, True)])
def test_window_fused_producer(ops, ref_dim, strided):
    """raw offsets / logits + reference points in, softmax and location arithmetic inside the kernel
    (ms_deform_attn.py:326-349): against the oracle's materialised sequence; `strided`: the two producer tensors are
    column slices of one [rows, 384] projection output, as the module passes them."""
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
    off_d, lg_d = off.to(DEV), logits.to(DEV)
    if strided:
        both = torch.cat([off_d.view(B, S, 256), lg_d.view(B, S, 128)], -1)
        off_d, lg_d = both[..., :256].view(B, S, 8, L, 4, 2), both[..., 256:].view(B, S, 8, L * 4)
        assert not off_d.is_contiguous()
    v = value.to(DEV)
    args = (shp.to(DEV), start.to(DEV), off_d, lg_d, ref.to(DEV))
    out = ops.ms_deform_attn_forward_fused(v, *args, algo="window").float().cpu().numpy()
    direct = ops.ms_deform_attn_forward_fused(v, *args, algo="direct").float().cpu().numpy()
    loc = torch_ref.sampling_locations_from_reference(ref, off.float(), shp, 4)
    w = logits.float().softmax(-1).view(B, S, 8, L, 4)
    expect = torch_ref.msda_core(value.float(), shp, loc, w).numpy()
    _check(out, expect, direct)
    out_h = ops.ms_deform_attn_forward_fused(_head_major(v), *args, value_layout="bhsd", algo="window").float().cpu().numpy()
    assert np.array_equal(out_h, out)
    auto = ops.ms_deform_attn_forward_fused(v, *args).float().cpu().numpy()
    assert np.array_equal(auto, direct) or np.array_equal(auto, out)


def test_window_unsupported_shapes(ops):
    from relation_detr_amd import _lib
    # five levels
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)], 1, 2.0, 1)
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward(v.to(DEV), shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV), algo="window")
    # Nq != S
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 96), (32, 48), (16, 24), (8, 12)], 1, 2.0, 2)
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward(v.to(DEV), shp.to(DEV), start.to(DEV), loc[:, :900].contiguous().to(DEV),
                                   attn[:, :900].contiguous().to(DEV), algo="window")
    # ... while the plain operator serves both through the direct kernel, in either layout
    out = ops.ms_deform_attn_forward(v.to(DEV), shp.to(DEV), start.to(DEV), loc[:, :900].contiguous().to(DEV),
                                     attn[:, :900].contiguous().to(DEV), 64)
    out_h = ops.ms_deform_attn_forward(_head_major(v.to(DEV)), shp.to(DEV), start.to(DEV), loc[:, :900].contiguous().to(DEV),
                                       attn[:, :900].contiguous().to(DEV), 64, value_layout="bhsd")
    assert out.shape == (1, 900, 256) and torch.equal(out, out_h)
    with pytest.raises(ValueError):
        ops.ms_deform_attn_forward(v.to(DEV), shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV), algo="fastest")


def test_window_full_size_properties(ops):
    """BASELINE.json configs[1] size (B = 4, S = Nq = 22,323): a constant value map returns the constant wherever all
    samples fall inside the levels (weights sum to one), and two launches on the same inputs are bit-identical (no
    atomics, fixed summation order)."""
    value, shp, start, loc, attn, S, L = _encoder_inputs(R50, 4, 4.0, seed=11)
    wh = shp.flip(-1).float().view(1, 1, 1, L, 1, 2)
    loc = torch.minimum(torch.maximum(loc, 1.0 / wh), 1.0 - 1.0 / wh).contiguous()
    dev = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    const = (torch.arange(256, dtype=torch.float32).view(1, 1, 8, 32) / 64).expand(4, S, 8, 32).contiguous().to(torch.bfloat16)
    oc = ops.ms_deform_attn_forward(const.to(DEV), *dev, algo="window").float().cpu()
    assert (oc - const[:, :1].reshape(4, 1, 256).float()).abs().max().item() <= 2.0 ** -7 * 4
    v = value.to(DEV)
    o1 = ops.ms_deform_attn_forward(v, *dev, algo="window")
    o2 = ops.ms_deform_attn_forward(v, *dev, algo="window")
    assert torch.equal(o1, o2)
    d = ops.ms_deform_attn_forward(v, *dev, algo="direct").float()
    assert ((o1.float() - d).abs() <= 2.0 ** -7 * d.abs() + 1e-3).all()
    oh = ops.ms_deform_attn_forward(_head_major(v), *dev, value_layout="bhsd", algo="window")
    assert torch.equal(oh, o1)


@pytest.mark.parametrize("shapes", [[(56, 72), (28, 36), (14, 18), (7, 9)],                       # 4 levels
                                    [(64, 80), (32, 40), (16, 20), (8, 10), (4, 5)]])              # 5 levels (FocalNet-style pyramid)
def test_module_head_major_route_matches_operator_layout_route_and_oracle(shapes, monkeypatch):
    """MultiScaleDeformableAttention in bf16 eval at the encoder shape (queries = the pyramid's pixels): value_proj writes the
    head-major layout and the gather reads it (default) -- against the same module with RDETR_VALUE_HEAD_MAJOR=0 (the `_C`
    layout [B,S,H,D]) and against the fp32 oracle restatement of the module (ms_deform_attn.py:286-377)."""
    from oracle import torch_ref
    from relation_detr_amd import MultiScaleDeformableAttention
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    assert S >= 4096
    torch.manual_seed(L)
    m = MultiScaleDeformableAttention(256, L, 8, 4).eval()
    B = 2
    g = torch.Generator().manual_seed(5)
    src = torch.randn(B, S, 256, generator=g) * 0.5
    pos = torch.randn(B, S, 256, generator=g) * 0.1
    ref = torch.rand(B, S, L, 2, generator=g)
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[1, S // 2::7] = True
    shp = torch.tensor(shapes, dtype=torch.int64)
    start = torch.cat([shp.new_zeros(1), (shp[:, 0] * shp[:, 1]).cumsum(0)[:-1]])
    md = MultiScaleDeformableAttention(256, L, 8, 4).to(DEV).to(torch.bfloat16).eval()
    md.load_state_dict({k: v.to(torch.bfloat16) for k, v in m.state_dict().items()})
    args = dict(query=(src + pos).to(DEV).to(torch.bfloat16), reference_points=ref.to(DEV), value=src.to(DEV).to(torch.bfloat16),
                spatial_shapes=shp.to(DEV), level_start_index=start.to(DEV), key_padding_mask=mask.to(DEV))
    with torch.no_grad():
        hm = md(**args).float().cpu()
        from relation_detr_amd import options
        options.apply(md, value_head_major=False)
        plain = md(**args).float().cpu()
        params = {k: v.to(torch.bfloat16).float() for k, v in m.state_dict().items()}          # the bf16-rounded weights, fp32 math
        want = torch_ref.msda_module_forward(params, (src + pos).to(torch.bfloat16).float(), ref, src.to(torch.bfloat16).float(),
                                             shp, start, mask, 8, L, 4)
    scale = want.abs().max().item()
    assert (hm - plain).abs().max().item() <= 2.0 ** -6 * scale                 # two bf16 GEMM routes for value_proj
    assert (hm - want).abs().max().item() <= 2.0 ** -5 * scale                  # 4 bf16 GEMMs + bf16 value / output storage
    assert ((hm - want).abs().mean() / want.abs().mean()).item() <= 2.0 ** -7


def test_head_major_full_size_five_levels(ops):
    """BASELINE.json configs[4] (FocalNet-L, 1216 x 2016 padded, 5 levels, S = Nq = 204,098, one image): the direct kernel on the
    head-major value gives the bits it gives on the operator's layout (same arithmetic, other addressing), and a sample of rows
    agrees with the C oracle."""
    from oracle import c_oracle
    shapes = [(304, 504), (152, 252), (76, 126), (38, 63), (19, 32)]
    shp, start, S = pyramid(shapes)
    assert S == 204098
    g = torch.Generator().manual_seed(9)
    value = torch.randn(1, S, 8, 32, generator=g).to(torch.bfloat16)
    L = len(shapes)
    ys = torch.cat([(torch.arange(h * w) // w + 0.5) / h for h, w in shapes])
    xs = torch.cat([(torch.arange(h * w) % w + 0.5) / w for h, w in shapes])
    ref = torch.stack([xs, ys], -1)                                                       # every pixel's own centre
    wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    off = torch.randn(1, S, 8, L, 4, 2, generator=g) * 2.0 / wh.view(1, 1, 1, L, 1, 2)
    loc = (ref.view(1, S, 1, 1, 1, 2) + off).contiguous()
    attn = torch.softmax(torch.randn(1, S, 8, L * 4, generator=g), -1).view(1, S, 8, L, 4).contiguous()
    dv, dl, da = value.to(DEV), loc.to(DEV), attn.to(DEV)
    a = ops.ms_deform_attn_forward(dv, shp.to(DEV), start.to(DEV), dl, da, algo="direct")
    b = ops.ms_deform_attn_forward(dv.permute(0, 2, 1, 3).contiguous(), shp.to(DEV), start.to(DEV), dl, da, value_layout="bhsd")
    assert torch.equal(a, b)
    rows = torch.cat([torch.arange(0, 64), torch.arange(153216 - 32, 153216 + 32), torch.arange(S - 64, S)])
    want = c_oracle.msda_forward(value.float().numpy(), shp.numpy(), start.numpy(), loc[:, rows].contiguous().numpy(),
                                 attn[:, rows].contiguous().numpy())
    got = b[:, rows.to(DEV)].float().cpu().numpy()
    assert (np.abs(got - want) <= 2.0 ** -8 * np.abs(want) + 1e-3).all()


def test_new_entry_points_accept_empty_inputs(ops):
    """Degenerate sizes of the round-2 entry points: no queries / no rows is a no-op, no keys is refused."""
    from relation_detr_amd import _lib
    z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=DEV)
    boxes0, boxes5 = torch.zeros(2, 0, 4, device=DEV), torch.rand(2, 5, 4, device=DEV)
    w, b = torch.zeros(8, 64, device=DEV), torch.zeros(8, device=DEV)
    assert ops.relation_attention_boxes(z(2, 0, 256), z(2, 5, 256), z(2, 5, 256), 8, boxes0, boxes5, w, b).shape == (2, 0, 256)
    with pytest.raises(_lib.RdetrError):
        ops.relation_attention_boxes(z(2, 5, 256), z(2, 0, 256), z(2, 0, 256), 8, boxes5, boxes0, w, b)
    assert ops.value_to_head_major(z(0, 64, 256)).shape == (0, 8, 64, 32)
