"""Transformer-level harness (relation_detr_amd/transformer.py) against the reference's RelationTransformer
eval forward frozen in tests/golden/g7_transformer.npz (4 levels) and g9_transformer_l5.npz (5 levels: the level count of
BASELINE.json configs[4], relation_detr_focalnet_large_lrf_fl4_1200_2000) -- weights = helpers.synthetic_state_dict, identical
on both sides.  CPU: harness glue with the oracle's operators; GPU: the same harness with the HIP-backed modules."""
import numpy as np
import pytest
import torch

from helpers import synthetic_state_dict

T = torch.from_numpy


FIXTURES = {"g7_transformer.npz": 4, "g9_transformer_l5.npz": 5}


def _build(golden, fixture="g7_transformer.npz", **kw):
    from relation_detr_amd.transformer import build_relation_transformer
    g = golden(fixture)
    nlev = FIXTURES[fixture]
    assert g["shapes"].shape == (nlev, 2)
    net = build_relation_transformer(num_classes=11, d_ffn=64, enc_layers=2, dec_layers=3, num_queries=24,
                                     hybrid_num_proposals=30, num_levels=nlev, **kw).eval()
    names = [str(n) for n in g["param_names"]]
    shapes = [tuple(int(v) for v in s.split(";")) if s else () for s in g["param_shapes"]]
    sd = net.state_dict()
    assert list(sd.keys()) == names                         # the reference's parameter names, in its order
    assert [tuple(v.shape) for v in sd.values()] == shapes
    net.load_state_dict(synthetic_state_dict(sd))
    feats = [T(g[f"feat{i}"]) for i in range(nlev)]
    masks = [T(g[f"mask{i}"]) for i in range(nlev)]
    pos = [T(g[f"pos{i}"]) for i in range(nlev)]
    return g, net, feats, masks, pos


def _check(g, outs, atol):
    assert len(outs) == 8 and all(o is None for o in outs[4:])          # eval: no hybrid branch (relation_transformer.py:147-148)
    oc, ob, ec, eb = [o.float().cpu().numpy() for o in outs[:4]]
    np.testing.assert_allclose(ec, g["enc_classes"], rtol=0, atol=atol)
    np.testing.assert_allclose(eb, g["enc_coords"], rtol=0, atol=atol)
    np.testing.assert_allclose(oc, g["out_classes"], rtol=0, atol=atol)
    np.testing.assert_allclose(ob, g["out_coords"], rtol=0, atol=atol)


@pytest.mark.parametrize("fixture", list(FIXTURES))
def test_harness_glue_matches_reference_on_cpu(golden, fixture):
    from oracle.cpu_modules import OracleMSDA, OracleRelation, OracleSelfAttention
    g, net, feats, masks, pos = _build(golden, fixture, msda_cls=OracleMSDA, self_attn_cls=OracleSelfAttention,
                                       relation_cls=OracleRelation)
    with torch.no_grad():
        outs = net(feats, masks, pos)
    assert outs[0].shape == (3, 3, 24, 11) and outs[1].shape == (3, 3, 24, 4)
    _check(g, outs, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fixture", list(FIXTURES))
def test_harness_with_hip_modules_matches_reference(golden, fixture, dtype):
    g, net, feats, masks, pos = _build(golden, fixture)
    if dtype == torch.bfloat16:
        _bf16_harness_vs_fixture(g, net, feats, masks, pos)
        return
    net = net.to("cuda:0")
    with torch.no_grad():
        outs = net([f.to("cuda:0") for f in feats], [m.to("cuda:0") for m in masks], [p.to("cuda:0") for p in pos])
    # 2 encoder + 3 decoder layers of fp32 GEMMs / LayerNorms between the kernels: 5e-4 on O(1) logits, and the
    # discrete top-k proposal choice must come out identical (a swapped proposal would show as an O(1) error)
    _check(g, outs, 5e-4)


def _bf16_harness_vs_fixture(g, net, feats, masks, pos):
    """The bf16 inference route (fused-producer MSDA on head-major value where it applies, generated-bias attention, fused box
    head, the glue kernels) on the reference's fixture network, 4 and 5 levels.
    (i)  Whole forward in bf16: finite, right shapes, and the two-stage proposals agree with the reference's as a SET (bf16
         reorders near-equal scores, so slots differ -- tests/test_gpu_fullsize.py explains why that is not comparable).
    (ii) The continuous bound: bf16 decoder vs fp32 decoder of the same network on IDENTICAL inputs -- the fp32 run's encoder
         memory and proposals, that fp32 run being the one the fp32 case of this test pins to the reference's fixture."""
    import copy
    dev = "cuda:0"
    net32 = net.to(dev)
    net16 = copy.deepcopy(net32).to(torch.bfloat16)
    f32, m32, p32 = [f.to(dev) for f in feats], [m.to(dev) for m in masks], [p.to(dev) for p in pos]
    seen = {}
    net32.encoder.register_forward_hook(lambda m, i, o: seen.__setitem__("memory", o.detach().clone()))
    with torch.no_grad():
        outs = net16([f.to(torch.bfloat16) for f in f32], m32, [p.to(torch.bfloat16) for p in p32])
        ref_outs = net32(f32, m32, p32)
    oc, ob, ec, eb = [o.float().cpu().numpy() for o in outs[:4]]
    assert np.isfinite(oc).all() and np.isfinite(ob).all() and np.isfinite(ec).all() and np.isfinite(eb).all()
    assert oc.shape == g["out_classes"].shape and ob.shape == g["out_coords"].shape
    d = np.abs(eb[:, :, None, :] - g["enc_coords"][:, None, :, :]).max(-1)            # [B, ours, reference]
    common = (d.min(1) < 2e-2).mean()
    print("bf16 two-stage proposals also picked by the reference:", common)
    assert common >= 0.7, common
    B = f32[0].shape[0]
    with torch.no_grad():
        geo, vr = net32.level_misc(m32)
        kw = dict(key_padding_mask=net32.flatten_levels(m32), reference_points=ref_outs[3].float().detach(),
                  spatial_shapes=geo["shapes"], level_start_index=geo["start"], valid_ratios=vr)
        c32, b32 = net32.decoder(query=net32.tgt_embed.weight.expand(B, -1, -1), value=seen["memory"], **kw)
        c16, b16 = net16.decoder(query=net16.tgt_embed.weight.expand(B, -1, -1), value=seen["memory"].to(torch.bfloat16), **kw)
    np.testing.assert_allclose(c32.cpu().numpy(), g["out_classes"], rtol=0, atol=5e-4)      # the fp32 side IS the fixture's decoder
    np.testing.assert_allclose(b32.cpu().numpy(), g["out_coords"], rtol=0, atol=5e-4)
    dbox, dcls = (b16.float() - b32).abs(), (c16.float() - c32).abs()
    scale = c32.abs().max().item()
    print(f"bf16 decoder vs fp32 on identical proposals ({geo['shapes'].shape[0]} levels): boxes max {dbox.max().item():.4f} mean "
          f"{dbox.mean().item():.5f}; logits max {dcls.max().item():.4f} mean {dcls.mean().item():.5f} (scale {scale:.2f})")
    assert dbox.max().item() <= 5e-3 and dbox.mean().item() <= 5e-4
    assert dcls.max().item() <= 2.0 ** -4 * scale and dcls.mean().item() <= 2.0 ** -7 * scale
