"""GPU parity tests (run with ``-m gpu`` on an MI355X): HIP kernels, called through the C ABI via
relation_detr_amd.ops / the nn.Modules, against (1) the committed golden vectors produced by the
reference itself and (2) the CPU oracle on seeded inputs.

Tolerances (north_star: <= 1e-4 abs fp32 vs the pure-PyTorch fallback):
  fp32 MSDA forward / module / relation bias / self-attention   atol 1e-4
  bf16 MSDA forward: |err| <= 2^-8 * |ref| + 1e-3  vs the fp32 oracle on bf16-rounded value (one
       output rounding to bf16, fp32 accumulation) -- an extension, the reference op has no bf16.
  backward: grad_value / grad_attn atol 1e-4; grad_loc rtol 1e-4 + atol 5e-4 away from interpolation
       kinks (helpers.kink_mask) -- it is scaled by W_l/H_l and sums 32 channels.
"""
import numpy as np
import pytest
import torch

from helpers import kink_mask, make_msda_inputs, pyramid

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
T = torch.from_numpy


@pytest.fixture(scope="module")
def rd():
    import relation_detr_amd
    from relation_detr_amd import _lib
    _lib.load()                      # fail loudly if the HIP library is missing
    return relation_detr_amd


def _core(rd, value, shapes, start, loc, attn):
    out = rd.ms_deform_attn_forward(value.to(DEV).contiguous(), shapes.to(DEV), start.to(DEV),
                                    loc.to(DEV).contiguous(), attn.to(DEV).contiguous(), 64)
    torch.cuda.synchronize()
    return out.cpu()


# ------------------------------------------------------------------------------------------ MSDA forward
@pytest.mark.parametrize("name", ["g1_msda_core.npz", "g3_msda_core_l5.npz"])
def test_msda_forward_golden(rd, golden, name):
    g = golden(name)
    out = _core(rd, T(g["value"]), T(g["shapes"]), T(g["level_start"]), T(g["loc"]), T(g["attn"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=1e-4)
    if "out_f64" in g:
        np.testing.assert_allclose(out.numpy(), g["out_f64"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("shapes,B,Nq", [
    ([(25, 42), (13, 21), (7, 11), (4, 6)], 3, 257),          # L=4 fast path, ragged Nq (not a multiple of 16)
    ([(19, 31), (10, 16), (5, 8), (3, 4), (2, 2)], 2, 130),   # L=5 fast path
    ([(9, 14), (5, 7)], 2, 65),                               # run-time L path
    ([(6, 6)], 1, 1),                                         # single level, single query
])
def test_msda_forward_vs_oracle(rd, shapes, B, Nq):
    from oracle import c_oracle, torch_ref
    value, shp, start, loc, attn = make_msda_inputs(B, Nq, shapes, seed=11)
    out = _core(rd, value, shp, start, loc, attn).numpy()
    ref_c = c_oracle.msda_forward(value.numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
    ref_t = torch_ref.msda_core(value, shp, loc, attn).numpy()
    np.testing.assert_allclose(out, ref_c, rtol=0, atol=1e-4)
    np.testing.assert_allclose(out, ref_t, rtol=0, atol=1e-4)


def test_msda_forward_generic_shapes(rd):
    """(H, D, P) outside the wave-per-query kernel run the generic kernel."""
    from oracle import c_oracle
    for H, D, P in [(4, 16, 2), (8, 32, 3), (2, 64, 4)]:
        assert rd._lib.load().rdetr_msda_fast_path(H, D, 3, P) == 0
        value, shp, start, loc, attn = make_msda_inputs(2, 33, [(8, 12), (4, 6), (2, 3)], H=H, D=D, P=P, seed=5)
        out = _core(rd, value, shp, start, loc, attn).numpy()
        ref = c_oracle.msda_forward(value.numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-4)


def test_msda_forward_bf16(rd):
    from oracle import c_oracle
    value, shp, start, loc, attn = make_msda_inputs(2, 300, [(25, 42), (13, 21), (7, 11), (4, 6)], seed=3)
    vb = value.to(torch.bfloat16)
    out = _core(rd, vb, shp, start, loc, attn)
    assert out.dtype == torch.bfloat16
    ref = c_oracle.msda_forward(vb.float().numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
    err = np.abs(out.float().numpy() - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 1e-3).all(), err.max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_msda_forward_inputs_not_16_byte_aligned(rd, dtype):
    """The 4-level kernel reads a lane's share of the locations / weights as 16-byte vectors; tensors that start 8 bytes into a
    16-byte line (legal: the operator asks for natural alignment only) take the run-time-level instantiation -- same result."""
    value, shp, start, loc, attn = make_msda_inputs(2, 301, [(25, 42), (13, 21), (7, 11), (4, 6)], seed=11)
    v = value.to(dtype).to(DEV)
    args = (shp.to(DEV), start.to(DEV))
    want = rd.ms_deform_attn_forward(v, *args, loc.to(DEV), attn.to(DEV), 64)

    def shifted(t, n):                      # the same values, starting n elements into a fresh buffer
        buf = torch.empty(t.numel() + n, dtype=t.dtype, device=DEV)
        view = buf[n:].view(t.shape)
        view.copy_(t)
        return view
    loc8, attn8 = shifted(loc, 2), shifted(attn, 2)
    assert loc8.data_ptr() % 16 == 8 and attn8.data_ptr() % 16 == 8 and loc8.is_contiguous()
    got = rd.ms_deform_attn_forward(v, *args, loc8, attn8, 64)
    assert torch.equal(got, want)            # same per-corner arithmetic, same order of summation
    got = rd.ms_deform_attn_forward(v, *args, loc8, attn.to(DEV), 64)
    assert torch.equal(got, want)


def test_msda_forward_nan_and_far_locations(rd):
    """NaN / huge locations contribute zero (CUDA-op guard) and never fault."""
    value, shp, start, loc, attn = make_msda_inputs(1, 16, [(8, 12), (4, 6), (2, 3), (1, 2)], seed=9)
    loc2 = loc.clone()
    loc2[0, 0] = float("nan")
    loc2[0, 1] = 1e30
    loc2[0, 2] = -1e30
    loc2[0, 3] = float("inf")
    out = _core(rd, value, shp, start, loc2, attn)
    assert torch.isfinite(out).all()
    assert (out[0, :4] == 0).all()
    ref = _core(rd, value, shp, start, loc, attn)
    assert torch.equal(out[0, 4:], ref[0, 4:])


def test_msda_empty_and_errors(rd):
    value, shp, start, loc, attn = make_msda_inputs(1, 4, [(4, 4)], seed=1)
    out = rd.ms_deform_attn_forward(value.to(DEV), shp.to(DEV), start.to(DEV), loc[:, :0].to(DEV).contiguous(),
                                    attn[:, :0].to(DEV).contiguous(), 64)
    assert out.shape == (1, 0, 256)
    with pytest.raises(RuntimeError):                               # CPU tensors: no fallback
        rd.ms_deform_attn_forward(value, shp, start, loc, attn, 64)
    with pytest.raises(RuntimeError):                               # non-contiguous, as AT_ASSERTM in the reference
        rd.ms_deform_attn_forward(value.to(DEV).transpose(2, 3), shp.to(DEV), start.to(DEV), loc.to(DEV),
                                  attn.to(DEV), 64)
    bad = shp.clone()
    bad[0, 0] = 99                                                  # level does not fit S: refused on the host
    with pytest.raises(RuntimeError):
        rd.ms_deform_attn_forward(value.to(DEV), bad.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV), 64)


# ------------------------------------------------------------------------------------------ MSDA backward
def test_msda_backward_golden(rd, golden):
    g, gb = golden("g1_msda_core.npz"), golden("g2_msda_core_bwd.npz")
    args = [T(g[k]).to(DEV) for k in ("value", "shapes", "level_start", "loc", "attn")]
    gv, gl, ga = rd.ms_deform_attn_backward(*args, T(gb["grad_out"]).to(DEV), 64)
    torch.cuda.synchronize()
    np.testing.assert_allclose(gv.cpu().numpy(), gb["grad_value"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ga.cpu().numpy(), gb["grad_attn"], rtol=0, atol=1e-4)
    ok = kink_mask(g["loc"], g["shapes"])
    np.testing.assert_allclose(np.where(ok, gl.cpu().numpy(), 0), np.where(ok, gb["grad_loc"], 0), rtol=1e-4, atol=5e-4)


def test_msda_autograd_vs_oracle(rd):
    from oracle import c_oracle
    value, shp, start, loc, attn = make_msda_inputs(2, 70, [(12, 18), (6, 9), (3, 5), (2, 3)], seed=21)
    go = torch.randn(2, 70, 256, generator=torch.Generator().manual_seed(2))
    v, l, a = (t.to(DEV).requires_grad_() for t in (value, loc, attn))
    out = rd.MultiScaleDeformableAttnFunction.apply(v, shp.to(DEV), start.to(DEV), l, a, 64)
    out.backward(go.to(DEV))
    gv, gl, ga = c_oracle.msda_backward(value.numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy(), go.numpy())
    np.testing.assert_allclose(v.grad.cpu().numpy(), gv, rtol=0, atol=1e-4)
    np.testing.assert_allclose(a.grad.cpu().numpy(), ga, rtol=0, atol=1e-4)
    ok = kink_mask(loc.numpy(), shp.numpy())
    np.testing.assert_allclose(np.where(ok, l.grad.cpu().numpy(), 0), np.where(ok, gl, 0), rtol=1e-4, atol=5e-4)


@pytest.mark.parametrize("shapes,B,Nq", [([(12, 20), (6, 10), (3, 5), (2, 3)], 2, 37), ([(40, 56), (20, 28), (10, 14), (5, 7)], 2, 3010),
                                         ([(10, 16), (5, 8), (3, 4), (2, 2), (1, 1)], 1, 29)])
def test_msda_backward_deterministic_mode(rd, shapes, B, Nq):
    """rdetr_msda_backward_det_f32: grad_value through sorted per-row sums instead of float atomics.  Same
    gradients as the C oracle (double accumulation) within the bounds of the atomic kernel, the SAME BITS on a second run, every
    row written without zero-initialisation, and the switch follows torch.use_deterministic_algorithms."""
    from oracle import c_oracle
    from relation_detr_amd import ops
    value, shp, start, loc, attn = make_msda_inputs(B, Nq, shapes, seed=Nq, spread=0.2)
    loc[0, 0, 0, 0, 0, 0] = float("nan")                                          # contributes nothing
    g = torch.Generator().manual_seed(Nq + 1)
    go = torch.randn(B, Nq, 256, generator=g)
    dev = [t.to(DEV) for t in (value, shp, start, loc, attn, go)]
    gv, gl, ga = ops.ms_deform_attn_backward(*dev, deterministic=True)
    gv2, gl2, ga2 = ops.ms_deform_attn_backward(*dev, deterministic=True)
    assert torch.equal(gv, gv2) and torch.equal(gl, gl2) and torch.equal(ga, ga2)
    av, al, aa = ops.ms_deform_attn_backward(*dev, deterministic=False)          # the atomic kernel: same sums, another order
    assert torch.equal(gl, al) and torch.equal(ga, aa)
    np.testing.assert_allclose(gv.cpu().numpy(), av.cpu().numpy(), rtol=1e-5, atol=1e-4)
    loc_ref = loc.clone()
    loc_ref[0, 0, 0, 0, 0, 0] = -5.0                                             # the oracle: a far-outside point instead of NaN
    rv, rl, ra = c_oracle.msda_backward(value.numpy(), shp.numpy(), start.numpy(), loc_ref.numpy(), attn.numpy(), go.numpy())
    np.testing.assert_allclose(gv.cpu().numpy(), rv, rtol=0, atol=1e-4 * max(1.0, float(np.abs(rv).max())))
    np.testing.assert_allclose(ga.cpu().numpy(), ra, rtol=0, atol=1e-4 * max(1.0, float(np.abs(ra).max())))
    was = torch.are_deterministic_algorithms_enabled()
    try:
        torch.use_deterministic_algorithms(True, warn_only=True)
        v, l_, a_ = (t.clone().requires_grad_() for t in (dev[0], dev[3], dev[4]))
        rd.MultiScaleDeformableAttnFunction.apply(v, dev[1], dev[2], l_, a_, 64).backward(dev[5])
        assert torch.equal(v.grad, gv)
    finally:
        torch.use_deterministic_algorithms(was)


def test_msda_backward_generic_shapes(rd):
    from oracle import c_oracle
    value, shp, start, loc, attn = make_msda_inputs(2, 21, [(8, 12), (4, 6)], H=4, D=16, P=2, seed=8)
    go = torch.randn(2, 21, 64, generator=torch.Generator().manual_seed(4))
    gv, gl, ga = rd.ms_deform_attn_backward(value.to(DEV), shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV),
                                            go.to(DEV), 64)
    rv, rl, ra = c_oracle.msda_backward(value.numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy(), go.numpy())
    np.testing.assert_allclose(gv.cpu().numpy(), rv, rtol=0, atol=1e-4)
    np.testing.assert_allclose(ga.cpu().numpy(), ra, rtol=0, atol=1e-4)
    ok = kink_mask(loc.numpy(), shp.numpy())
    np.testing.assert_allclose(np.where(ok, gl.cpu().numpy(), 0), np.where(ok, rl, 0), rtol=1e-4, atol=5e-4)


# ------------------------------------------------------------------------------------------ module
def test_msda_module_golden(rd, golden):
    g = golden("g4_msda_module.npz")
    mod = rd.MultiScaleDeformableAttention(256, 4, 8, 4)
    mod.load_state_dict({k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")})
    mod = mod.to(DEV).eval()
    shapes, start = T(g["shapes"]).to(DEV), T(g["level_start"]).to(DEV)
    with torch.no_grad():
        enc = mod(query=T(g["q_enc"]).to(DEV), reference_points=T(g["ref2"]).to(DEV), value=T(g["feat"]).to(DEV),
                  spatial_shapes=shapes, level_start_index=start, key_padding_mask=T(g["mask"]).to(DEV))
        dec = mod(query=T(g["q_dec"]).to(DEV), reference_points=T(g["ref4"]).to(DEV), value=T(g["feat"]).to(DEV),
                  spatial_shapes=shapes, level_start_index=start, key_padding_mask=None)
    np.testing.assert_allclose(enc.cpu().numpy(), g["out_enc"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(dec.cpu().numpy(), g["out_dec"], rtol=0, atol=1e-4)


def test_msda_module_bf16_autocast(rd, golden):
    g = golden("g4_msda_module.npz")
    mod = rd.MultiScaleDeformableAttention(256, 4, 8, 4)
    mod.load_state_dict({k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")})
    mod = mod.to(DEV).eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        dec = mod(query=T(g["q_dec"]).to(DEV), reference_points=T(g["ref4"]).to(DEV), value=T(g["feat"]).to(DEV),
                  spatial_shapes=T(g["shapes"]).to(DEV), level_start_index=T(g["level_start"]).to(DEV),
                  key_padding_mask=None)
    assert dec.dtype == torch.bfloat16
    # bf16 GEMMs dominate the error here; this only checks that the bf16 kernel path is wired correctly
    np.testing.assert_allclose(dec.float().cpu().numpy(), g["out_dec"], rtol=0, atol=6e-2)


# ------------------------------------------------------------------------------------------ relation bias
def test_relation_bias_golden(rd, golden):
    g = golden("g5_relation.npz")
    rel = rd.PositionRelationEmbedding(16, 8)
    rel.load_state_dict({"pos_proj.0.weight": T(g["proj_weight"]), "pos_proj.0.bias": T(g["proj_bias"])})
    rel = rel.to(DEV).eval()
    with torch.no_grad():
        bias = rel(T(g["src"]).to(DEV), T(g["tgt"]).to(DEV))
        bias_self = rel(T(g["src"]).to(DEV))
        bias_tiny = rel(T(g["tiny_src"]).to(DEV), T(g["tiny_tgt"]).to(DEV))
    assert bias.shape == (2, 8, 23, 31)
    np.testing.assert_allclose(bias.cpu().numpy(), g["bias"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(bias.cpu().numpy(), g["bias_f64"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(bias_self.cpu().numpy(), g["bias_self"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(bias_tiny.cpu().numpy(), g["bias_tiny"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(bias_tiny.cpu().numpy(), g["bias_tiny_f64"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("B,N1,N2", [(1, 300, 300), (2, 129, 67), (1, 1, 1), (3, 5, 900)])
def test_relation_bias_vs_oracle(rd, B, N1, N2):
    from oracle import torch_ref
    g = torch.Generator().manual_seed(N1 * 7 + N2)
    src = torch.cat([torch.rand(B, N1, 2, generator=g), torch.rand(B, N1, 2, generator=g) * 0.49 + 0.01], -1)
    tgt = torch.cat([torch.rand(B, N2, 2, generator=g), torch.rand(B, N2, 2, generator=g) * 0.49 + 0.01], -1)
    w = (torch.rand(8, 64, 1, 1, generator=g) - 0.5) * 0.25
    b = (torch.rand(8, generator=g) - 0.5) * 0.25
    out = rd.relation_bias(src.to(DEV), tgt.to(DEV), w.to(DEV), b.to(DEV))
    ref64 = torch_ref.relation_bias(src.double(), tgt.double(), w.double(), b.double()).float()
    ref32 = torch_ref.relation_bias(src, tgt, w, b)
    np.testing.assert_allclose(out.cpu().numpy(), ref32.numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.cpu().numpy(), ref64.numpy(), rtol=0, atol=1e-4)


def test_relation_bias_generic_heads(rd):
    from oracle import torch_ref
    g = torch.Generator().manual_seed(77)
    src = torch.cat([torch.rand(2, 40, 2, generator=g), torch.rand(2, 40, 2, generator=g) * 0.4 + 0.02], -1)
    w = (torch.rand(4, 32, 1, 1, generator=g) - 0.5) * 0.3            # Hh = 4, F = 8
    b = (torch.rand(4, generator=g) - 0.5) * 0.3
    out = rd.relation_bias(src.to(DEV), src.to(DEV), w.to(DEV), b.to(DEV), num_pos_feats=8)
    ref = torch_ref.relation_bias(src, src, w, b, num_pos_feats=8)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-4)


def test_relation_bias_weight_grad(rd):
    from oracle import torch_ref
    g = torch.Generator().manual_seed(5)
    src = torch.cat([torch.rand(2, 33, 2, generator=g), torch.rand(2, 33, 2, generator=g) * 0.4 + 0.02], -1)
    tgt = torch.cat([torch.rand(2, 29, 2, generator=g), torch.rand(2, 29, 2, generator=g) * 0.4 + 0.02], -1)
    rel = rd.PositionRelationEmbedding(16, 8).to(DEV)
    go = torch.randn(2, 8, 33, 29, generator=g)
    # ReLU is only piecewise differentiable: where the pre-activation is within 1e-3 of zero, which side an implementation
    # lands on is rounding (the kernel and the fp32 reference differ by ~3e-5 there), so those entries get no gradient
    # in this comparison -- the analogue of helpers.kink_mask for the bilinear kinks
    with torch.no_grad():
        w0, b0 = rel.pos_proj[0].weight.detach().cpu().double(), rel.pos_proj[0].bias.detach().cpu().double()
        feat = torch_ref.sine_embed(torch_ref.box_rel_encoding(src.double(), tgt.double()))       # [B, N1, N2, 64]
        pre = torch.einsum("bijc,hc->bhij", feat, w0.view(8, -1)) + b0.view(1, 8, 1, 1)
        go = go * (pre.abs() > 1e-3)
    rel(src.to(DEV), tgt.to(DEV)).backward(go.to(DEV))
    w = rel.pos_proj[0].weight.detach().cpu().clone().requires_grad_()
    b = rel.pos_proj[0].bias.detach().cpu().clone().requires_grad_()
    torch_ref.relation_bias(src, tgt, w, b).backward(go)
    np.testing.assert_allclose(rel.pos_proj[0].weight.grad.cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(rel.pos_proj[0].bias.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=2e-3)


def test_relation_bias_backward_after_caller_mutates_output(rd):
    """The decoder fills the returned bias with -inf in place wherever attn_mask is set (relation_transformer.py:372-374,
    denoising training) and back-propagates through it: the autograd function must not have saved the tensor it handed out
    ('modified by an inplace operation'), and masked entries must get no gradient."""
    from oracle import torch_ref
    g = torch.Generator().manual_seed(11)
    boxes = torch.cat([torch.rand(2, 40, 2, generator=g), torch.rand(2, 40, 2, generator=g) * 0.4 + 0.02], -1)
    mask = torch.rand(40, 40, generator=g) < 0.3
    go = torch.randn(16, 40, 40, generator=g)
    rel = rd.PositionRelationEmbedding(16, 8).to(DEV)
    with torch.no_grad():                                 # no gradient through entries within 1e-3 of the ReLU kink (see the test above)
        w0, b0 = rel.pos_proj[0].weight.detach().cpu().double(), rel.pos_proj[0].bias.detach().cpu().double()
        feat = torch_ref.sine_embed(torch_ref.box_rel_encoding(boxes.double(), boxes.double()))
        pre = torch.einsum("bijc,hc->bhij", feat, w0.view(8, -1)) + b0.view(1, 8, 1, 1)
        go = go * (pre.abs() > 1e-3).flatten(0, 1)
    bias = rel(boxes.to(DEV)).flatten(0, 1)
    bias.masked_fill_(mask.to(DEV), float("-inf"))
    torch.where(torch.isinf(bias), torch.zeros_like(bias), bias).mul(go.to(DEV)).sum().backward()
    w = rel.pos_proj[0].weight.detach().cpu().clone().requires_grad_()
    b = rel.pos_proj[0].bias.detach().cpu().clone().requires_grad_()
    ref = torch_ref.relation_bias(boxes, boxes, w, b).flatten(0, 1).masked_fill(mask, float("-inf"))
    torch.where(torch.isinf(ref), torch.zeros_like(ref), ref).mul(go).sum().backward()
    np.testing.assert_allclose(rel.pos_proj[0].weight.grad.cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(rel.pos_proj[0].bias.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=2e-3)
    # masked entries got no gradient: the same call without the fill gives different gradients
    assert torch.isfinite(rel.pos_proj[0].weight.grad).all()


@pytest.mark.parametrize("B,N1,N2", [(1, 300, 300), (2, 97, 515), (3, 33, 257)])
def test_relation_bias_backward_kernel(rd, B, N1, N2):
    """rdetr_relation_bias_backward_f32 (csrc/relation_bwd.hip) against autograd through the oracle's materialised features
    (what the reference differentiates, relation_transformer.py:527-532) in DOUBLE precision: row / column tails on every
    tile edge, a masked share of g, the same bits on a second run (the reduction is a fixed-order tree, no atomics)."""
    from oracle import torch_ref
    from relation_detr_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + N1)
    src = torch.cat([torch.rand(B, N1, 2, generator=g), torch.rand(B, N1, 2, generator=g) * 0.4 + 0.02], -1)
    tgt = torch.cat([torch.rand(B, N2, 2, generator=g), torch.rand(B, N2, 2, generator=g) * 0.4 + 0.02], -1)
    go = torch.randn(B, 8, N1, N2, generator=g)
    active = torch.rand(B, 8, N1, N2, generator=g) < 0.6                       # the ReLU mask is an INPUT of the operator
    gw, gb = ops.relation_bias_backward(src.to(DEV), tgt.to(DEV), go.to(DEV), active.to(DEV))
    gw2, gb2 = ops.relation_bias_backward(src.to(DEV), tgt.to(DEV), go.to(DEV), active.to(DEV))
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    gm = (go * active).double()
    want_w = torch.zeros(8, 64, dtype=torch.float64)
    for b in range(B):                                                          # one image at a time bounds the feature tensor
        feat = torch_ref.sine_embed(torch_ref.box_rel_encoding(src[b:b + 1].double(), tgt[b:b + 1].double()))[0]      # [N1, N2, 64]
        want_w += torch.einsum("hij,ijc->hc", gm[b], feat)
    want_b = gm.sum(dim=(0, 2, 3))
    # fp32 features (angles up to 1e3 rad carry ~3e-5 rad of rounding) summed over up to 1e5 pairs per entry: 2e-3 of the
    # root-sum-square of the terms, i.e. of sqrt(pairs), is the bound the module-level tests use as well
    scale = float(np.sqrt(B * N1 * N2))
    np.testing.assert_allclose(gw.cpu().numpy(), want_w.numpy(), rtol=1e-4, atol=2e-3 * scale / 30)
    np.testing.assert_allclose(gb.cpu().numpy(), want_b.numpy(), rtol=1e-5, atol=1e-3)


def test_relation_bias_backward_kernel_empty_and_errors(rd):
    from relation_detr_amd import _lib, ops
    z = torch.zeros(2, 8, 0, 5, device=DEV)
    gw, gb = ops.relation_bias_backward(torch.zeros(2, 0, 4, device=DEV), torch.rand(2, 5, 4, device=DEV), z, z > 0)
    assert gw.shape == (8, 64) and not gw.any() and not gb.any()
    with pytest.raises(_lib.RdetrError):
        ops.relation_bias_backward(torch.rand(1, 4, 4, device=DEV), torch.rand(1, 4, 4, device=DEV), torch.zeros(1, 4, 4, 4, device=DEV),
                                   torch.zeros(1, 4, 4, 4, dtype=torch.bool, device=DEV))


# ------------------------------------------------------------------------------------------ bias softmax / self-attention
@pytest.mark.parametrize("BH,N1,N2", [(8, 50, 50), (16, 300, 300), (8, 37, 901), (2, 5, 1100), (1, 3, 5000), (4, 900, 900)])
def test_bias_softmax_vs_torch(rd, BH, N1, N2):
    g = torch.Generator().manual_seed(BH + N2)
    s = torch.randn(BH, N1, N2, generator=g) * 4
    b = torch.rand(BH, N1, N2, generator=g) * 3
    m = torch.rand(N1, N2, generator=g) < 0.2
    m[:, 0] = False
    for bias, mask in ((b, None), (None, None), (b, m), (None, m)):
        ref = s if bias is None else s + bias
        if mask is not None:
            ref = ref.masked_fill(mask, float("-inf"))
        ref = torch.softmax(ref, -1)
        out = rd.bias_softmax_(s.clone().to(DEV), None if bias is None else bias.to(DEV),
                               None if mask is None else mask.to(DEV))
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)


def test_bias_softmax_inf_bias_rows(rd):
    s = torch.randn(2, 4, 64)
    b = torch.zeros(2, 4, 64)
    b[:, :, 10:] = float("-inf")
    out = rd.bias_softmax_(s.clone().to(DEV), b.to(DEV)).cpu()
    np.testing.assert_allclose(out.numpy(), torch.softmax(s + b, -1).numpy(), rtol=0, atol=2e-6)
    assert (out[..., 10:] == 0).all()


def test_self_attention_golden(rd, golden):
    g = golden("g6_self_attn.npz")
    att = rd.RelationSelfAttention(256, 8)
    att.load_state_dict({"in_proj_weight": T(g["in_proj_weight"]), "in_proj_bias": T(g["in_proj_bias"]),
                         "out_proj.weight": T(g["out_proj_weight"]), "out_proj.bias": T(g["out_proj_bias"])})
    att = att.to(DEV).eval()
    qp, vv = T(g["qp"]).to(DEV), T(g["vv"]).to(DEV)
    rb, bm = T(g["rel_bias"]).to(DEV), T(g["bool_mask"]).to(DEV)
    cases = {"out_bias": rb, "out_none": None, "out_bool": bm, "out_bias_inf": rb.masked_fill(bm, float("-inf"))}
    with torch.no_grad():
        for key, mask in cases.items():
            out = att(query=qp, key=qp, value=vv, attn_mask=mask, need_weights=False)[0]
            np.testing.assert_allclose(out.cpu().numpy(), g[key], rtol=0, atol=1e-4, err_msg=key)


# ------------------------------------------------------------------------------------------ full-size checks
R50 = [(100, 168), (50, 84), (25, 42), (13, 21)]


def test_full_size_encoder_vs_oracle(rd):
    """BASELINE.json config 1/2 encoder shape (S = Nq = 22,323) against the torch oracle on the host."""
    from oracle import torch_ref
    value, shp, start, loc, attn = make_msda_inputs(1, 22323, R50, seed=0, spread=0.02)
    out = _core(rd, value, shp, start, loc, attn)
    ref = torch_ref.msda_core(value, shp, loc, attn)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=0, atol=1e-4)


def test_full_size_properties(rd):
    """Size-independent properties at B=4 (BASELINE.json config 2): linearity in value, and a constant
    value map sampled strictly inside every level returns that constant (weights sum to one)."""
    shp, start, S = pyramid(R50)
    g = torch.Generator().manual_seed(1)
    B, Nq = 4, 22323
    v1 = torch.randn(B, S, 8, 32, generator=g).to(DEV)
    v2 = torch.randn(B, S, 8, 32, generator=g).to(DEV)
    wh = shp.flip(-1).float()
    loc = (torch.rand(B, Nq, 8, 4, 4, 2, generator=g) * (1 - 2.0 / wh.view(1, 1, 1, 4, 1, 2)) + 1.0 / wh.view(1, 1, 1, 4, 1, 2)).to(DEV)
    attn = torch.softmax(torch.randn(B, Nq, 8, 16, generator=g), -1).view(B, Nq, 8, 4, 4).to(DEV)
    shp_d, start_d = shp.to(DEV), start.to(DEV)
    o1 = rd.ms_deform_attn_forward(v1, shp_d, start_d, loc, attn, 64)
    o2 = rd.ms_deform_attn_forward(v2, shp_d, start_d, loc, attn, 64)
    o12 = rd.ms_deform_attn_forward(v1 + 0.5 * v2, shp_d, start_d, loc, attn, 64)
    assert (o12 - (o1 + 0.5 * o2)).abs().max().item() < 2e-5
    const = torch.arange(256, dtype=torch.float32, device=DEV).view(1, 1, 8, 32).expand(B, S, 8, 32).contiguous() / 64
    oc = rd.ms_deform_attn_forward(const, shp_d, start_d, loc, attn, 64)
    assert (oc - const[:, :1].reshape(B, 1, 256)).abs().max().item() < 2e-5


# ------------------------------------------------------------------------------------------ fused producer
@pytest.mark.parametrize("L,ref_dim,dtype", [(4, 2, torch.float32), (4, 4, torch.float32), (5, 2, torch.float32),
                                             (3, 4, torch.float32), (4, 2, torch.bfloat16), (4, 4, torch.bfloat16),
                                             (5, 2, torch.bfloat16), (5, 4, torch.bfloat16), (5, 4, torch.float32)])
def test_msda_fused_producer_matches_unfused(rd, L, ref_dim, dtype):
    """softmax + location arithmetic inside the kernel == the reference's materialised sequence
    (ms_deform_attn.py:322-349) followed by the plain operator / the oracle."""
    from oracle import torch_ref
    shapes_l = [(23, 37), (12, 19), (6, 10), (3, 5), (2, 3)][:L]
    shp, start, S = pyramid(shapes_l)
    g = torch.Generator().manual_seed(100 + L * 10 + ref_dim)
    B, Nq = 2, 75
    value = torch.randn(B, S, 8, 32, generator=g)
    off = torch.randn(B, Nq, 8, L, 4, 2, generator=g) * 3
    logits = torch.randn(B, Nq, 8, L * 4, generator=g) * 2
    if ref_dim == 2:
        ref = torch.rand(B, Nq, L, 2, generator=g)
    else:
        ref = torch.cat([torch.rand(B, Nq, L, 2, generator=g), torch.rand(B, Nq, L, 2, generator=g) * 0.5 + 0.02], -1)
    vq, oq, lq = value.to(dtype), off.to(dtype), logits.to(dtype)
    out = rd.ms_deform_attn_forward_fused(vq.to(DEV), shp.to(DEV), start.to(DEV), oq.to(DEV), lq.to(DEV), ref.to(DEV))
    loc = torch_ref.sampling_locations_from_reference(ref, oq.float(), shp, 4)
    w = lq.float().softmax(-1).view(B, Nq, 8, L, 4)
    ref_out = torch_ref.msda_core(vq.float(), shp, loc, w)
    if dtype == torch.float32:
        np.testing.assert_allclose(out.cpu().numpy(), ref_out.numpy(), rtol=0, atol=1e-4)
        plain = rd.ms_deform_attn_forward(vq.to(DEV), shp.to(DEV), start.to(DEV), loc.to(DEV).contiguous(), w.to(DEV).contiguous(), 64)
        assert (out - plain).abs().max().item() < 2e-5
    else:
        err = (out.float().cpu() - ref_out).abs()
        assert (err <= 2.0 ** -8 * ref_out.abs() + 1e-3).all(), err.max()


def test_msda_module_train_and_eval_paths_agree(rd, golden):
    """eval (no grad) takes the fused kernel, training the autograd Function: same numbers."""
    g = golden("g4_msda_module.npz")
    mod = rd.MultiScaleDeformableAttention(256, 4, 8, 4)
    mod.load_state_dict({k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")})
    mod = mod.to(DEV)
    kw = dict(query=T(g["q_dec"]).to(DEV), reference_points=T(g["ref4"]).to(DEV), value=T(g["feat"]).to(DEV),
              spatial_shapes=T(g["shapes"]).to(DEV), level_start_index=T(g["level_start"]).to(DEV), key_padding_mask=None)
    train_out = mod(**kw)
    assert train_out.requires_grad
    train_out.sum().backward()
    assert mod.sampling_offsets.weight.grad is not None and torch.isfinite(mod.value_proj.weight.grad).all()
    with torch.no_grad():
        eval_out = mod(**kw)
    assert (train_out.detach() - eval_out).abs().max().item() < 2e-5
    np.testing.assert_allclose(eval_out.cpu().numpy(), g["out_dec"], rtol=0, atol=1e-4)


# ------------------------------------------------------------------------------------------ encoder shape
def _pixel_refs(shapes):
    refs = []
    for h, w in shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    return torch.cat(refs, 0)


@pytest.mark.parametrize("shapes,B,spread_px,dtype", [
    ([(100, 168), (50, 84), (25, 42), (13, 21)], 1, 3.0, torch.float32),     # R50 pyramid, offsets inside the band
    ([(100, 168), (50, 84), (25, 42), (13, 21)], 1, 30.0, torch.float32),    # most samples take the fallback path
    ([(37, 61), (19, 31), (10, 16), (5, 8), (3, 4)], 2, 4.0, torch.float32),  # odd 5-level pyramid
    ([(9, 14), (5, 7), (3, 4)], 3, 2.0, torch.float32),                       # tiny: windows clamp to the level
    ([(40, 33)], 1, 5.0, torch.float32),                                      # single level
    ([(64, 96), (32, 48), (16, 24), (8, 12)], 2, 4.0, torch.bfloat16),
])
def test_encoder_entry_matches_oracle_and_plain_entry(rd, shapes, B, spread_px, dtype, monkeypatch):
    """Encoder-shape calls (Nq == S): the Python op must agree with the oracle and, to rounding, with the plain C entry
    point called directly through ctypes, whatever the offsets and the pyramid (5 levels, 1 level, tiny levels).  Where the
    LDS-window kernel applies (bf16, 4 levels, S >= 4096) both go through it; tests/test_gpu_window.py pins it against the
    direct kernel case by case."""
    from oracle import c_oracle
    shp, start, S = pyramid(shapes)
    L = len(shapes)
    g = torch.Generator().manual_seed(int(spread_px * 10) + L)
    value = torch.randn(B, S, 8, 32, generator=g).to(dtype)
    wh = shp.flip(-1).float()
    off = torch.randn(B, S, 8, L, 4, 2, generator=g) * spread_px / wh.view(1, 1, 1, L, 1, 2)
    loc = (_pixel_refs(shapes)[None, :, None, None, None, :] + off).contiguous()
    attn = torch.softmax(torch.randn(B, S, 8, L * 4, generator=g), -1).view(B, S, 8, L, 4)
    args = (value.to(DEV), shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV), 64)
    out = rd.ms_deform_attn_forward(*args).float().cpu().numpy()
    lib = rd._lib.load()                                                   # the C ABI, called directly
    direct_t = torch.empty(B, S, 256, dtype=dtype, device=DEV)
    fn = lib.rdetr_msda_forward_f32 if dtype == torch.float32 else lib.rdetr_msda_forward_bf16
    assert fn(args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(), args[3].data_ptr(), args[4].data_ptr(), B, S, 8, 32,
              L, S, 4, direct_t.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    direct = direct_t.float().cpu().numpy()
    ref = c_oracle.msda_forward(value.float().numpy(), shp.numpy(), start.numpy(), loc.numpy(), attn.numpy())
    if dtype == torch.float32:
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-4)
        np.testing.assert_allclose(out, direct, rtol=0, atol=2e-5)
    else:
        assert (np.abs(out - ref) <= 2.0 ** -8 * np.abs(ref) + 1e-3).all()
        assert (np.abs(out - direct) <= 2.0 ** -7 * np.abs(ref) + 1e-3).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_entry_fused_matches_unfused(rd, dtype, monkeypatch):
    from oracle import torch_ref
    shapes = [(30, 50), (15, 25), (8, 13), (4, 7)]
    shp, start, S = pyramid(shapes)
    g = torch.Generator().manual_seed(4)
    B, L = 2, 4
    value = torch.randn(B, S, 8, 32, generator=g).to(dtype)
    off = (torch.randn(B, S, 8, L, 4, 2, generator=g) * 3).to(dtype)
    logits = (torch.randn(B, S, 8, L * 4, generator=g) * 2).to(dtype)
    ref = _pixel_refs(shapes)[None, :, None, :].expand(B, S, L, 2).contiguous()
    out = rd.ms_deform_attn_forward_fused(value.to(DEV), shp.to(DEV), start.to(DEV), off.to(DEV), logits.to(DEV), ref.to(DEV))
    loc = torch_ref.sampling_locations_from_reference(ref, off.float(), shp, 4)
    w = logits.float().softmax(-1).view(B, S, 8, L, 4)
    expect = torch_ref.msda_core(value.float(), shp, loc, w)
    if dtype == torch.float32:
        np.testing.assert_allclose(out.cpu().numpy(), expect.numpy(), rtol=0, atol=1e-4)
    else:
        err = (out.float().cpu() - expect).abs()
        assert (err <= 2.0 ** -8 * expect.abs() + 1e-3).all(), err.max()


# ------------------------------------------------------------------------------------------ FocalNet-size pyramid
FOCAL = [(304, 504), (152, 252), (76, 126), (38, 63), (19, 32)]      # BASELINE.json configs[4]: 1216 x 2016 padded, 5 levels


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_focalnet_size_pyramid(rd, dtype):
    """S = 204,098 positions, L = 5 (relation_detr_focalnet_large_lrf_fl4_1200_2000): a 5,000-query launch against the C
    oracle, and the full encoder-shape launch (Nq = S) must reproduce those rows bit for bit (same kernel, same inputs per
    row) and stay finite everywhere."""
    from oracle import c_oracle
    shp, start, S = pyramid(FOCAL)
    assert S == 204098
    g = torch.Generator().manual_seed(77)
    value = torch.randn(1, S, 8, 32, generator=g).to(dtype)
    loc = (torch.rand(1, S, 8, 5, 4, 2, generator=g) * 1.1 - 0.05)
    attn = torch.softmax(torch.randn(1, S, 8, 20, generator=g), -1).view(1, S, 8, 5, 4)
    pick = torch.randperm(S, generator=g)[:5000].sort().values
    v_d, shp_d, start_d = value.to(DEV), shp.to(DEV), start.to(DEV)
    sub = rd.ms_deform_attn_forward(v_d, shp_d, start_d, loc[:, pick].contiguous().to(DEV), attn[:, pick].contiguous().to(DEV), 64)
    ref = c_oracle.msda_forward(value.float().numpy(), shp.numpy(), start.numpy(), loc[:, pick].contiguous().numpy(),
                                attn[:, pick].contiguous().numpy())
    got = sub.float().cpu().numpy()
    if dtype == torch.float32:
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-4)
    else:
        assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 1e-3).all()
    full = rd.ms_deform_attn_forward(v_d, shp_d, start_d, loc.to(DEV), attn.to(DEV), 64)
    assert torch.isfinite(full.float()).all()
    assert torch.equal(full[:, pick.to(DEV)], sub)


@pytest.mark.parametrize("dtype,ref_dim", [(torch.float32, 2), (torch.bfloat16, 2), (torch.float32, 4)])
def test_msda_fused_padding_mask_inside_kernel(rd, dtype, ref_dim):
    """key_padding_mask applied in the gather (padded rows count as zero) == zero-filling the projected value first
    (ms_deform_attn.py:316-319): bit-identical, since a zero row and a skipped row add the same 0."""
    shapes_l = [(23, 37), (12, 19), (6, 10), (3, 5)]
    shp, start, S = pyramid(shapes_l)
    g = torch.Generator().manual_seed(31 + ref_dim)
    B, Nq, L = 2, 200, 4
    value = torch.randn(B, S, 8, 32, generator=g).to(dtype)
    off = (torch.randn(B, Nq, 8, L, 4, 2, generator=g) * 3).to(dtype)
    logits = (torch.randn(B, Nq, 8, L * 4, generator=g) * 2).to(dtype)
    ref = torch.rand(B, Nq, L, 2, generator=g)
    if ref_dim == 4:
        ref = torch.cat([ref, torch.rand(B, Nq, L, 2, generator=g) * 0.5 + 0.02], -1)
    mask = torch.rand(B, S, generator=g) < 0.3                       # 30 % of the positions padded
    dev = lambda t: t.to(DEV).contiguous()
    got = rd.ms_deform_attn_forward_fused(dev(value), dev(shp), dev(start), dev(off), dev(logits), dev(ref), dev(mask))
    filled = value.masked_fill(mask[:, :, None, None], 0)
    want = rd.ms_deform_attn_forward_fused(dev(filled), dev(shp), dev(start), dev(off), dev(logits), dev(ref))
    assert torch.equal(got, want)
    none = rd.ms_deform_attn_forward_fused(dev(value), dev(shp), dev(start), dev(off), dev(logits), dev(ref))
    assert not torch.equal(got, none)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_mask", [False, True])
def test_msda_fused_strided_producer_rows(dtype, with_mask):
    """offsets / logits as column slices of ONE [rows, 3*H*L*P] projection output (rdetr_msda_forward_fused_ex_*):
    bit-identical to the same numbers handed over as two contiguous tensors."""
    from relation_detr_amd import ops
    torch.manual_seed(11)
    dev = "cuda"
    B, H, D, L, P = 2, 8, 32, 4, 4
    shapes = torch.tensor([[20, 27], [10, 14], [5, 7], [3, 4]], device=dev)
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    for Nq in (S, 37):
        value = torch.randn(B, S, H, D, device=dev).to(dtype)
        both = (torch.randn(B, Nq, 3 * H * L * P, device=dev) * 2).to(dtype)
        n_off = H * L * P * 2
        off_v = both[..., :n_off].view(B, Nq, H, L, P, 2)
        lg_v = both[..., n_off:].view(B, Nq, H, L * P)
        assert not off_v.is_contiguous() and ops._producer_row_stride(off_v) == 3 * H * L * P
        ref = torch.rand(B, Nq, L, 4, device=dev) * 0.8 + 0.1
        mask = (torch.rand(B, S, device=dev) < 0.2) if with_mask else None
        got = ops.ms_deform_attn_forward_fused(value, shapes, start, off_v, lg_v, ref, mask)
        want = ops.ms_deform_attn_forward_fused(value, shapes, start, off_v.contiguous(), lg_v.contiguous(), ref, mask)
        assert torch.equal(got, want)
    # a slice whose rows are not evenly strided falls back to a copy, not to a wrong read
    odd = torch.randn(B, 37, H, L, P, 4, device=dev).to(dtype)[..., ::2]
    assert ops._producer_row_stride(odd) is None


@pytest.mark.parametrize("with_mask", [False, True])
def test_msda_fused_row_strided_value(with_mask):
    """rdetr_msda_forward_fused_strided_bf16: the value as a 256-column slice of a wider [B, S, 6*256] buffer (the decoder's six
    cross-attention value projections as one GEMM) gives the bits of the same call on a contiguous copy, for decoder-shaped
    queries with 4-d reference points, with and without the in-kernel padding mask."""
    from relation_detr_amd import ops
    shapes = [(20, 34), (10, 17), (5, 9), (3, 5)]
    shp, start, S = pyramid(shapes)
    g = torch.Generator().manual_seed(17)
    B, Nq, L = 2, 70, 4
    wide = torch.randn(B, S, 6 * 256, generator=g).to(torch.bfloat16).to(DEV)
    off = (torch.randn(B, Nq, 8, L, 4, 2, generator=g) * 2).to(torch.bfloat16).to(DEV)
    lg = torch.randn(B, Nq, 8, L * 4, generator=g).to(torch.bfloat16).to(DEV)
    ref = torch.cat([torch.rand(B, Nq, L, 2, generator=g), torch.rand(B, Nq, L, 2, generator=g) * 0.4 + 0.05], -1).to(DEV)
    mask = (torch.rand(B, S, generator=g) < 0.2).to(DEV) if with_mask else None
    for l in (0, 3, 5):
        view = wide[..., 256 * l:256 * (l + 1)].view(B, S, 8, 32)
        assert not view.is_contiguous()
        got = ops.ms_deform_attn_forward_fused(view, shp.to(DEV), start.to(DEV), off, lg, ref, mask)
        want = ops.ms_deform_attn_forward_fused(view.contiguous(), shp.to(DEV), start.to(DEV), off, lg, ref, mask)
        assert torch.equal(got, want)


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_msda_five_level_bf16_input_paths_agree(rd, ref_dim):
    """The 5-level bf16 kernel loads a lane's five consecutive points with vector loads (a lane's points span two levels: two sets
    of level constants and reference points per lane, csrc/msda_fwd.hip kSpan); inputs that miss its alignment take the run-time-L
    kernel with narrow loads.  Contiguous producer tensors and column slices of one packed projection output give the same bits
    (with and without the in-kernel padding mask, Nq not a multiple of the wave's 16 queries); the run-time-L kernel agrees to a
    bf16 rounding (it sums the softmax denominator in another order), and so does the generic kernel for the operator form."""
    from relation_detr_amd import ops
    shapes_l = [(23, 37), (12, 19), (6, 10), (3, 5), (2, 3)]
    shp, start, S = pyramid(shapes_l)
    shp, start = shp.to(DEV), start.to(DEV)
    g = torch.Generator().manual_seed(500 + ref_dim)
    B, Nq, L, H = 2, 77, 5, 8
    value = torch.randn(B, S, H, 32, generator=g).to(torch.bfloat16).to(DEV)
    packed = (torch.randn(B, Nq, 3 * H * L * 4, generator=g) * 2).to(torch.bfloat16).to(DEV)
    off = packed[..., :H * L * 8].view(B, Nq, H, L, 4, 2)
    lg = packed[..., H * L * 8:].view(B, Nq, H, L * 4)
    ref = torch.rand(B, Nq, L, 2, generator=g)
    if ref_dim == 4:
        ref = torch.cat([ref, torch.rand(B, Nq, L, 2, generator=g) * 0.5 + 0.02], -1)
    ref = ref.to(DEV)
    mask = (torch.rand(B, S, generator=g) < 0.25).to(DEV)

    def misaligned(t):                       # same numbers at an address that is 4 (mod 16): the run-time-L kernel
        buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
        view = buf[1:].view(t.shape)
        view.copy_(t)
        assert view.data_ptr() % 16 == 4 and view.is_contiguous()
        return view
    for m in (None, mask):
        fast = ops.ms_deform_attn_forward_fused(value, shp, start, off, lg, ref, m)
        dense = ops.ms_deform_attn_forward_fused(value, shp, start, off.contiguous(), lg.contiguous(), ref, m)
        slow = ops.ms_deform_attn_forward_fused(value, shp, start, off.contiguous(), lg.contiguous(), misaligned(ref), m)
        assert torch.equal(fast, dense)
        # the run-time-L kernel deals the points to the lanes differently: the softmax denominator is summed in another order
        err = (fast.float() - slow.float()).abs()
        assert (err <= 2.0 ** -8 * fast.float().abs() + 1e-3).all() and (err > 0).float().mean().item() < 0.05
    # operator form (materialised fp32 locations / weights): vector loads vs the generic kernel (location tensor at 4 mod 8)
    loc = torch.rand(B, Nq, H, L, 4, 2, generator=g).to(DEV) * 1.2 - 0.1
    w = torch.softmax(torch.randn(B, Nq, H, L * 4, generator=g), -1).view(B, Nq, H, L, 4).to(DEV).contiguous()
    a = rd.ms_deform_attn_forward(value, shp, start, loc, w, 64)
    b = rd.ms_deform_attn_forward(value, shp, start, misaligned(loc), w, 64)
    err = (a.float() - b.float()).abs()
    assert (err <= 2.0 ** -8 * a.float().abs() + 1e-3).all()          # the generic kernel sums in another order
