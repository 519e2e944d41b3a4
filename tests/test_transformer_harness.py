"""Transformer-level harness (relation_detr_amd/transformer.py) against the reference's RelationTransformer
eval forward frozen in tests/golden/g7_transformer.npz (weights = helpers.synthetic_state_dict, identical on both
sides).  CPU: harness glue with the oracle's operators; GPU: the same harness with the HIP-backed modules."""
import numpy as np
import pytest
import torch

from helpers import synthetic_state_dict

T = torch.from_numpy


def _build(golden, **kw):
    from relation_detr_amd.transformer import build_relation_transformer
    g = golden("g7_transformer.npz")
    net = build_relation_transformer(num_classes=11, d_ffn=64, enc_layers=2, dec_layers=3, num_queries=24,
                                     hybrid_num_proposals=30, **kw).eval()
    names = [str(n) for n in g["param_names"]]
    shapes = [tuple(int(v) for v in s.split(";")) if s else () for s in g["param_shapes"]]
    sd = net.state_dict()
    assert list(sd.keys()) == names                         # the reference's parameter names, in its order
    assert [tuple(v.shape) for v in sd.values()] == shapes
    net.load_state_dict(synthetic_state_dict(sd))
    feats = [T(g[f"feat{i}"]) for i in range(4)]
    masks = [T(g[f"mask{i}"]) for i in range(4)]
    pos = [T(g[f"pos{i}"]) for i in range(4)]
    return g, net, feats, masks, pos


def _check(g, outs, atol):
    assert len(outs) == 8 and all(o is None for o in outs[4:])          # eval: no hybrid branch (relation_transformer.py:147-148)
    oc, ob, ec, eb = [o.float().cpu().numpy() for o in outs[:4]]
    np.testing.assert_allclose(ec, g["enc_classes"], rtol=0, atol=atol)
    np.testing.assert_allclose(eb, g["enc_coords"], rtol=0, atol=atol)
    np.testing.assert_allclose(oc, g["out_classes"], rtol=0, atol=atol)
    np.testing.assert_allclose(ob, g["out_coords"], rtol=0, atol=atol)


def test_harness_glue_matches_reference_on_cpu(golden):
    from oracle.cpu_modules import OracleMSDA, OracleRelation, OracleSelfAttention
    g, net, feats, masks, pos = _build(golden, msda_cls=OracleMSDA, self_attn_cls=OracleSelfAttention,
                                       relation_cls=OracleRelation)
    with torch.no_grad():
        outs = net(feats, masks, pos)
    assert outs[0].shape == (3, 3, 24, 11) and outs[1].shape == (3, 3, 24, 4)
    _check(g, outs, 2e-5)


@pytest.mark.gpu
def test_harness_with_hip_modules_matches_reference(golden):
    g, net, feats, masks, pos = _build(golden)
    net = net.to("cuda:0")
    with torch.no_grad():
        outs = net([f.to("cuda:0") for f in feats], [m.to("cuda:0") for m in masks], [p.to("cuda:0") for p in pos])
    # 2 encoder + 3 decoder layers of fp32 GEMMs / LayerNorms between the kernels: 5e-4 on O(1) logits, and the
    # discrete top-k proposal choice must come out identical (a swapped proposal would show as an O(1) error)
    _check(g, outs, 5e-4)
