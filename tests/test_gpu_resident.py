"""GPU parity tests of the resident-levels kernel (csrc/msda_res.hip: bf16, head-major value, 4 or 5 levels; persistent
workgroups keep the coarse levels of an (image, head) plane in LDS and read those levels' corner rows from there, the fine
levels through the buffer descriptor), called through the C ABI (rdetr_msda_forward_resident_bf16 /
rdetr_msda_forward_fused_resident_bf16, HOST level table) -- against the C oracle on bf16-rounded value and against the
query-run kernel on the same inputs.

Tolerance: |err| <= 2^-8 |ref| + 1e-3 vs the fp32 oracle (one bf16 output rounding, fp32 accumulation; the reference op has
no bf16 -- SURVEY.md Appendix B item 12); vs the query-run kernel 2^-7 |ref| + 1e-3 (same arithmetic per point, the points
accumulated in another order: two independently rounded outputs).
"""
import numpy as np
import pytest
import torch

from helpers import pyramid
from test_gpu_window import DEV, R50, _check, _encoder_inputs, _head_major, _pixel_refs

pytestmark = pytest.mark.gpu

FOCAL_SMALL = [(76, 126), (38, 63), (19, 32), (10, 16), (5, 8)]


@pytest.fixture(scope="module")
def ops():
    from relation_detr_amd import _lib, ops
    _lib.load()
    return ops


@pytest.mark.parametrize("shapes,B,spread_px,scatter,poison", [
    (R50, 1, 3.0, 0.0, False),                                           # levels 2 and 3 resident (the benchmark's pyramid)
    (R50, 3, 4.0, 0.0, True),                                            # 24 planes on 8 XCDs: workgroups that idle; NaN / border cases
    (R50, 2, 30.0, 0.3, False),                                          # wide offsets + scattered queries: corners beyond every border
    ([(160, 160), (80, 80), (48, 48), (20, 20)], 1, 4.0, 0.05, True),    # levels 2 + 3 = 173 KB: only level 3 is resident
    ([(12, 20), (6, 10), (3, 5), (2, 3)], 2, 2.0, 0.1, True),            # the golden fixtures' tiny pyramid: most waves have no run
    ([(37, 53), (19, 27), (10, 14), (5, 7)], 5, 5.0, 0.0, True),         # odd sizes, 40 planes: five per XCD
    (FOCAL_SMALL, 2, 4.0, 0.02, True),                                   # five levels: levels 3 and 4 resident
    ([(300, 500), (150, 250), (75, 125), (38, 63), (19, 32)], 1, 4.0, 0.0, False),   # FocalNet-L 1200 x 2000: only level 4 fits
])
def test_resident_matches_oracle_and_direct(ops, shapes, B, spread_px, scatter, poison):
    from oracle import c_oracle
    value, shp, start, loc, attn, S, L = _encoder_inputs(shapes, B, spread_px, seed=int(spread_px * 5) + B, scatter=scatter,
                                                         poison=poison)
    rest = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    vh = _head_major(value.to(DEV))
    ref = c_oracle.msda_forward(value.float().numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
    direct = ops.ms_deform_attn_forward(vh, *rest, value_layout="bhsd", algo="direct").float().cpu().numpy()
    out = ops.ms_deform_attn_forward(vh, *rest, value_layout="bhsd", algo="resident").float().cpu().numpy()
    _check(out, ref, direct)
    again = ops.ms_deform_attn_forward(vh, *rest, value_layout="bhsd", algo="resident").float().cpu().numpy()
    assert np.array_equal(again, out)                     # no atomics, fixed summation order


def test_resident_queries_that_are_not_the_pixels(ops):
    """Nq != S (decoder-like queries, any reference points): the kernel only needs the levels to tile the value."""
    from oracle import c_oracle
    shp, start, S = pyramid(R50)
    g = torch.Generator().manual_seed(77)
    B, Nq, L = 2, 1000, 4
    value = torch.randn(B, S, 8, 32, generator=g).to(torch.bfloat16)
    loc = (torch.rand(B, Nq, 8, L, 4, 2, generator=g) * 1.2 - 0.1).contiguous()
    attn = torch.softmax(torch.randn(B, Nq, 8, L * 4, generator=g), -1).view(B, Nq, 8, L, 4).contiguous()
    ref = c_oracle.msda_forward(value.float().numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
    out = ops.ms_deform_attn_forward(_head_major(value.to(DEV)), shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV),
                                     value_layout="bhsd", algo="resident").float().cpu().numpy()
    _check(out, ref)


@pytest.mark.parametrize("shapes,ref_dim,strided", [(R50, 2, False), (R50, 4, False), (R50, 2, True), (FOCAL_SMALL, 2, True), (FOCAL_SMALL, 2, False)])
def test_resident_fused_producer(ops, shapes, ref_dim, strided):
    """raw offsets / logits + reference points in, softmax and location arithmetic inside the kernel
    (ms_deform_attn.py:326-349): against the oracle's materialised sequence; `strided`: the two producer tensors are
    column slices of one projection output, as the module passes them."""
    from oracle import torch_ref
    shp, start, S = pyramid(shapes)
    g = torch.Generator().manual_seed(50 + ref_dim + len(shapes))
    B, L = 2, len(shapes)
    value = torch.randn(B, S, 8, 32, generator=g).to(torch.bfloat16)
    off = (torch.randn(B, S, 8, L, 4, 2, generator=g) * 3).to(torch.bfloat16)
    logits = (torch.randn(B, S, 8, L * 4, generator=g) * 2).to(torch.bfloat16)
    ref = _pixel_refs(shapes)[None, :, None, :].expand(B, S, L, 2)
    if ref_dim == 4:
        ref = torch.cat([ref, torch.rand(B, S, L, 2, generator=g) * 0.2 + 0.02], -1)
    ref = ref.contiguous()
    off_d, lg_d = off.to(DEV), logits.to(DEV)
    if strided:
        both = torch.cat([off_d.view(B, S, 64 * L), lg_d.view(B, S, 32 * L)], -1)
        off_d, lg_d = both[..., :64 * L].view(B, S, 8, L, 4, 2), both[..., 64 * L:].view(B, S, 8, L * 4)
        assert not off_d.is_contiguous()
    vh = _head_major(value.to(DEV))
    args = (shp.to(DEV), start.to(DEV), off_d, lg_d, ref.to(DEV))
    if ref_dim == 4:                      # 4-d reference points (the decoder's form): not this kernel's; 'auto' falls back
        from relation_detr_amd import _lib
        with pytest.raises(_lib.RdetrError, match="not supported"):
            ops.ms_deform_attn_forward_fused(vh, *args, value_layout="bhsd", algo="resident")
        out = ops.ms_deform_attn_forward_fused(vh, *args, value_layout="bhsd").float().cpu().numpy()
    else:
        out = ops.ms_deform_attn_forward_fused(vh, *args, value_layout="bhsd", algo="resident").float().cpu().numpy()
    direct = ops.ms_deform_attn_forward_fused(vh, *args, value_layout="bhsd", algo="direct").float().cpu().numpy()
    loc = torch_ref.sampling_locations_from_reference(ref, off.float(), shp, 4)
    w = logits.float().softmax(-1).view(B, S, 8, L, 4)
    expect = torch_ref.msda_core(value.float(), shp, loc, w).numpy()
    _check(out, expect, direct)


def test_resident_refuses_what_it_cannot_serve(ops):
    from relation_detr_amd import _lib
    # three levels
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 64), (32, 32), (16, 16)], 1, 2.0, 1)
    rest = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward(_head_major(v.to(DEV)), *rest, value_layout="bhsd", algo="resident")
    # ... which 'auto' serves through the query-run kernel
    assert ops.ms_deform_attn_forward(_head_major(v.to(DEV)), *rest, value_layout="bhsd").shape == (1, S, 256)
    # the reference operator's layout
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 96), (32, 48), (16, 24), (8, 12)], 1, 2.0, 2)
    rest = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    with pytest.raises(_lib.RdetrError, match="head-major"):
        ops.ms_deform_attn_forward(v.to(DEV), *rest, algo="resident")
    # a coarsest level that does not fit the CU's LDS beside the staging area
    v, shp, start, loc, attn, S, L = _encoder_inputs([(64, 64), (56, 56), (52, 52), (48, 48)], 1, 2.0, 3)
    rest = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward(_head_major(v.to(DEV)), *rest, value_layout="bhsd", algo="resident")
    # levels that do not tile [0, S): check_levels accepts gaps, the resident copy cannot
    shp2 = torch.tensor([(64, 96), (32, 48), (16, 24), (8, 12)], dtype=torch.int64)
    areas = [h * w for h, w in shp2.tolist()]
    start2 = torch.tensor([0, areas[0] + 100, areas[0] + 100 + areas[1], areas[0] + 100 + areas[1] + areas[2]], dtype=torch.int64)
    S2 = int(start2[3]) + areas[3]
    g = torch.Generator().manual_seed(4)
    v2 = torch.randn(1, 8, S2, 32, generator=g).to(torch.bfloat16).to(DEV)
    loc2 = torch.rand(1, S2, 8, 4, 4, 2, generator=g).to(DEV)
    attn2 = torch.softmax(torch.randn(1, S2, 8, 16, generator=g), -1).view(1, S2, 8, 4, 4).to(DEV)
    with pytest.raises(_lib.RdetrError, match="not supported"):
        ops.ms_deform_attn_forward(v2, shp2.to(DEV), start2.to(DEV), loc2, attn2, value_layout="bhsd", algo="resident")
    assert ops.ms_deform_attn_forward(v2, shp2.to(DEV), start2.to(DEV), loc2, attn2, value_layout="bhsd").shape == (1, S2, 256)


def test_resident_full_size_properties(ops):
    """BASELINE.json configs[1] size (B = 4, S = Nq = 22,323), where 'auto' on a head-major value IS this kernel: a constant value
    map returns the constant wherever all samples fall inside the levels (weights sum to one); two launches are bit-identical;
    the result agrees with the query-run kernel's."""
    value, shp, start, loc, attn, S, L = _encoder_inputs(R50, 4, 4.0, seed=11)
    wh = shp.flip(-1).float().view(1, 1, 1, L, 1, 2)
    loc = torch.minimum(torch.maximum(loc, 1.0 / wh), 1.0 - 1.0 / wh).contiguous()
    dev = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
    const = (torch.arange(256, dtype=torch.float32).view(1, 1, 8, 32) / 64).expand(4, S, 8, 32).contiguous().to(torch.bfloat16)
    oc = ops.ms_deform_attn_forward(_head_major(const.to(DEV)), *dev, value_layout="bhsd", algo="resident").float().cpu()
    assert (oc - const[:, :1].reshape(4, 1, 256).float()).abs().max().item() <= 2.0 ** -7 * 4
    vh = _head_major(value.to(DEV))
    o1 = ops.ms_deform_attn_forward(vh, *dev, value_layout="bhsd", algo="resident")
    o2 = ops.ms_deform_attn_forward(vh, *dev, value_layout="bhsd", algo="resident")
    assert torch.equal(o1, o2)
    d = ops.ms_deform_attn_forward(vh, *dev, value_layout="bhsd", algo="direct").float()
    assert ((o1.float() - d).abs() <= 2.0 ** -7 * d.abs() + 1e-3).all()
    assert torch.equal(ops.ms_deform_attn_forward(vh, *dev, value_layout="bhsd"), o1)           # auto


def test_resident_kernel_in_a_captured_graph(ops):
    """The kernel only enqueues on the current stream (its LDS attribute and the CU count are looked up in the warm-up): a HIP-graph
    replay reproduces the eager launch bit for bit, also with new inputs copied into the captured buffers."""
    from relation_detr_amd.graph import GraphedCall
    value, shp, start, loc, attn, S, L = _encoder_inputs(R50, 1, 4.0, seed=3)
    shp_d, start_d = shp.to(DEV), start.to(DEV)
    vh, loc_d, attn_d = _head_major(value.to(DEV)), loc.to(DEV), attn.to(DEV)
    fn = lambda v, lo, at: ops.ms_deform_attn_forward(v, shp_d, start_d, lo, at, value_layout="bhsd", algo="resident")
    eager = fn(vh, loc_d, attn_d).clone()
    run = GraphedCall(fn, [vh.clone(), loc_d.clone(), attn_d.clone()])
    assert torch.equal(run(vh, loc_d, attn_d), eager)
    value2, _, _, loc2, attn2, _, _ = _encoder_inputs(R50, 1, 6.0, seed=4)
    vh2, loc2_d, attn2_d = _head_major(value2.to(DEV)), loc2.to(DEV), attn2.to(DEV)
    assert torch.equal(run(vh2, loc2_d, attn2_d), fn(vh2, loc2_d, attn2_d))
