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
    """The bf16 inference route (fused-producer MSDA on head-major value, generated-bias attention, the glue kernels) on the
    reference's fixture.  The fixture's two-stage scores are separated by far more than bf16 resolution only for some
    proposals, so the discrete choice may differ: the ENCODER outputs are compared as sets (every reference proposal's box
    is found among ours), and the decoder is held to a bf16-sized bound on the queries whose proposals coincide."""
    dev = "cuda:0"
    net = net.to(dev).to(torch.bfloat16)
    with torch.no_grad():
        outs = net([f.to(dev).to(torch.bfloat16) for f in feats], [m.to(dev) for m in masks],
                   [p.to(dev).to(torch.bfloat16) for p in pos])
    oc, ob, ec, eb = [o.float().cpu().numpy() for o in outs[:4]]
    assert np.isfinite(oc).all() and np.isfinite(ob).all()
    assert oc.shape == g["out_classes"].shape and ob.shape == g["out_coords"].shape
    # encoder proposals: same boxes (sigmoid space) within bf16 noise wherever the same token was picked, in the same slot
    same = np.abs(eb - g["enc_coords"]).max(-1) < 2e-2                     # [B, N]
    assert same.mean() >= 0.8, same.mean()
    np.testing.assert_allclose(ec[same], g["enc_classes"][same], rtol=0, atol=6e-2)
    # decoder: images whose proposal sets coincide completely are comparable query by query (self-attention mixes queries)
    full = same.all(1)
    assert full.any(), "no image kept its full proposal set in bf16 -- fixture no longer discriminates"
    np.testing.assert_allclose(ob[:, full], g["out_coords"][:, full], rtol=0, atol=3e-2)
    np.testing.assert_allclose(oc[:, full], g["out_classes"][:, full], rtol=0, atol=0.15)
