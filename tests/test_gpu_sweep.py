"""GPU parity tests of the sweep kernel (csrc/msda_sweep.hip: bf16, Nq == S, L == 4; persistent workgroups sliding ring-buffer
windows of all four levels, LDS-sourced MFMA gather), called through the C ABI (rdetr_msda_forward_sweep_bf16, HOST level
table) in BOTH value layouts -- against the C oracle on bf16-rounded value and against the direct query-run kernel on the
same inputs.  The ring windows keep 8 pixels of margin around a step's footprint; samples outside are fetched from global
memory and added in fp32.  The cases therefore sweep the offset spread from "everything inside the windows" to "nothing
inside" -- the result must not depend on it -- and put samples on / beyond every border.

Tolerance: |err| <= 2^-8 |ref| + 1e-3 vs the fp32 oracle (one bf16 output rounding, fp32 accumulation; the reference op has
no bf16 -- SURVEY.md Appendix B item 12); vs the direct kernel 2^-7 |ref| + 1e-3 (two independently rounded outputs).
"""
import numpy as np
import pytest
import torch

from test_gpu_window import DEV, R50, _check, _encoder_inputs, _head_major

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from relation_detr_amd import _lib, ops
    _lib.load()
    return ops


@pytest.mark.parametrize("shapes,B,spread_px,scatter,poison", [
    (R50, 1, 3.0, 0.0, False),                                   # (nearly) every sample inside the ring windows
    (R50, 2, 4.0, 0.0, True),                                    # BASELINE spread (sigma up to 4 px) + NaN / border cases
    (R50, 1, 30.0, 0.0, False),                                  # most fine-level samples come from global memory (several trips per wave)
    ([(64, 96), (32, 48), (16, 24), (8, 12)], 2, 4.0, 0.05, False),      # multiples of the step; 5 % scattered queries
    ([(70, 70), (35, 35), (18, 18), (9, 9)], 1, 2.0, 1.0, False),        # all queries scattered: no locality at all
    ([(160, 24), (80, 12), (40, 6), (20, 3)], 2, 3.0, 0.02, True),        # levels narrower than a ring: zero columns on both sides
    ([(12, 20), (6, 10), (3, 5), (2, 3)], 2, 2.0, 0.1, True),            # the golden fixtures' tiny pyramid: one step per workgroup
    ([(37, 53), (19, 27), (10, 14), (5, 7)], 3, 5.0, 0.0, True),         # odd sizes: ragged last band and last step
])
def test_sweep_matches_oracle_and_direct(ops, shapes, B, spread_px, scatter, poison):
    from oracle import c_oracle
    value, shp, start, loc, attn, S, L = _encoder_inputs(shapes, B, spread_px, seed=int(spread_px * 7) + B, scatter=scatter,
                                                         poison=poison)
    rest = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    v = value.to(DEV)
    ref = c_oracle.msda_forward(value.float().numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
    direct = ops.ms_deform_attn_forward(v, *rest, algo="direct").float().cpu().numpy()
    out = ops.ms_deform_attn_forward(v, *rest, algo="sweep").float().cpu().numpy()
    _check(out, ref, direct)
    out_h = ops.ms_deform_attn_forward(_head_major(v), *rest, value_layout="bhsd", algo="sweep").float().cpu().numpy()
    assert np.array_equal(out_h, out)                     # same arithmetic, only the fill / patch addresses differ


def test_sweep_refuses_what_it_cannot_serve(ops):
    from relation_detr_amd import _lib
    # five levels
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)], 1, 2.0, 1)
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward(v.to(DEV), shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV), algo="sweep")
    # Nq != S
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 96), (32, 48), (16, 24), (8, 12)], 1, 2.0, 2)
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward(v.to(DEV), shp.to(DEV), start.to(DEV), loc[:, :900].contiguous().to(DEV),
                                   attn[:, :900].contiguous().to(DEV), algo="sweep")
    # a pyramid whose levels are not about half the finer one: two steps' columns do not fit the rings
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 96), (48, 80), (16, 24), (8, 12)], 1, 2.0, 3)
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward(v.to(DEV), shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV), algo="sweep")
    # levels that do not tile [0, S)
    shp2 = torch.tensor([(64, 96), (32, 48), (16, 24), (8, 12)], dtype=torch.int64)
    areas = [h * w for h, w in shp2.tolist()]
    start2 = torch.tensor([0, areas[0] + 100, areas[0] + 100 + areas[1], areas[0] + 100 + areas[1] + areas[2]], dtype=torch.int64)
    S2 = int(start2[3]) + areas[3]
    g = torch.Generator().manual_seed(4)
    v2 = torch.randn(1, S2, 8, 32, generator=g).to(torch.bfloat16).to(DEV)
    loc2 = torch.rand(1, S2, 8, 4, 4, 2, generator=g).to(DEV)
    attn2 = torch.softmax(torch.randn(1, S2, 8, 16, generator=g), -1).view(1, S2, 8, 4, 4).to(DEV)
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward(v2, shp2.to(DEV), start2.to(DEV), loc2, attn2, algo="sweep")


def test_sweep_full_size_properties(ops):
    """BASELINE.json configs[1] size (B = 4, S = Nq = 22,323): a constant value map returns the constant wherever all samples
    fall inside the levels (weights sum to one), and the result agrees with the direct kernel's in both value layouts."""
    value, shp, start, loc, attn, S, L = _encoder_inputs(R50, 4, 4.0, seed=11)
    wh = shp.flip(-1).float().view(1, 1, 1, L, 1, 2)
    loc = torch.minimum(torch.maximum(loc, 1.0 / wh), 1.0 - 1.0 / wh).contiguous()
    dev = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    const = (torch.arange(256, dtype=torch.float32).view(1, 1, 8, 32) / 64).expand(4, S, 8, 32).contiguous().to(torch.bfloat16)
    oc = ops.ms_deform_attn_forward(const.to(DEV), *dev, algo="sweep").float().cpu()
    assert (oc - const[:, :1].reshape(4, 1, 256).float()).abs().max().item() <= 2.0 ** -7 * 4
    v = value.to(DEV)
    o1 = ops.ms_deform_attn_forward(v, *dev, algo="sweep")
    o2 = ops.ms_deform_attn_forward(v, *dev, algo="sweep")
    d = ops.ms_deform_attn_forward(v, *dev, algo="direct").float()
    assert ((o1.float() - d).abs() <= 2.0 ** -7 * d.abs() + 1e-3).all()
    assert ((o2.float() - d).abs() <= 2.0 ** -7 * d.abs() + 1e-3).all()
    oh = ops.ms_deform_attn_forward(_head_major(v), *dev, value_layout="bhsd", algo="sweep")
    assert ((oh.float() - d).abs() <= 2.0 ** -7 * d.abs() + 1e-3).all()
