"""Training-mode forward + backward of the transformer harness against the reference's RelationTransformer in
``.train()`` (tests/golden/g8_transformer_train.npz, made by oracle/gen_golden.py from the reference's own Python):
denoising queries concatenated in front of the matching queries with their visibility mask (relation_transformer.py:
120-123, the -inf fill of the relation bias :373-374), the hybrid one-to-many branch through the same decoder with
``skip_relation=True`` (:101-115, :136-146), all 8 outputs, and the autograd gradients of a fixed linear functional of
them (helpers.functional_weights).  CPU: harness glue with the oracle's operators; GPU: the HIP-backed modules, i.e.
msda forward/backward kernels, relation-bias kernel + its backward, bias-softmax kernel + its backward."""
import numpy as np
import pytest
import torch

from helpers import G8_FULL_GRADS, functional_weights, synthetic_state_dict

T = torch.from_numpy


def _run(golden, device, **kw):
    from relation_detr_amd.transformer import build_relation_transformer
    g = golden("g8_transformer_train.npz")
    net = build_relation_transformer(num_classes=11, d_ffn=64, enc_layers=2, dec_layers=3, num_queries=24,
                                     hybrid_num_proposals=30, **kw)
    net.load_state_dict(synthetic_state_dict(net.state_dict()))
    net = net.to(device).train()
    feats = [T(g[f"feat{i}"]).to(device).requires_grad_(True) for i in range(4)]
    masks = [T(g[f"mask{i}"]).to(device) for i in range(4)]
    pos = [T(g[f"pos{i}"]).to(device) for i in range(4)]
    dn_label = T(g["dn_label"]).to(device).requires_grad_(True)
    dn_box = T(g["dn_box"]).to(device).requires_grad_(True)
    outs = net(feats, masks, pos, dn_label, dn_box, T(g["attn_mask"]).to(device))
    assert len(outs) == 8 and all(o is not None for o in outs)
    loss = sum((o.float() * functional_weights(o.shape, i).to(device)).sum() for i, o in enumerate(outs))
    loss.backward()
    return g, net, outs, loss, feats, dn_label, dn_box


def _check(g, net, outs, loss, feats, dn_label, dn_box, atol, gtol):
    for i, o in enumerate(outs):
        assert tuple(o.shape) == g[f"out{i}"].shape
        np.testing.assert_allclose(o.detach().float().cpu().numpy(), g[f"out{i}"], rtol=0, atol=atol, err_msg=f"output {i}")
    assert abs(loss.item() - float(g["loss"])) <= 200 * atol
    params = dict(net.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    assert list(params) == names
    for n, want in zip(names, g["grad_norms"]):
        got = 0.0 if params[n].grad is None else params[n].grad.double().norm().item()
        assert abs(got - want) <= gtol * max(1.0, want), (n, got, want)
    def close(got, want, what):
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got.detach().float().cpu().numpy(), want, rtol=0, atol=gtol * scale, err_msg=what)
    for n in G8_FULL_GRADS:
        close(params[n].grad, g[f"grad.{n}"], n)
    close(feats[3].grad, g["grad_feat3"], "d loss / d level-3 features")
    assert abs(feats[0].grad.double().norm().item() - float(g["grad_feat0_norm"])) <= gtol * max(1.0, float(g["grad_feat0_norm"]))
    close(dn_label.grad, g["grad_dn_label"], "d loss / d denoising label queries")
    close(dn_box.grad, g["grad_dn_box"], "d loss / d denoising box queries")


def test_training_forward_backward_matches_reference_on_cpu(golden):
    from oracle.cpu_modules import OracleMSDA, OracleRelation, OracleSelfAttention
    res = _run(golden, "cpu", msda_cls=OracleMSDA, self_attn_cls=OracleSelfAttention, relation_cls=OracleRelation)
    _check(*res, atol=2e-5, gtol=2e-4)


def test_eval_mode_ignores_hybrid_branch_but_takes_denoising_queries(golden):
    """The reference concatenates the denoising queries whenever they are given (:120-123) and runs the hybrid branch
    only in training mode (:100-118, :136-148)."""
    from oracle.cpu_modules import OracleMSDA, OracleRelation, OracleSelfAttention
    from relation_detr_amd.transformer import build_relation_transformer
    g = golden("g8_transformer_train.npz")
    net = build_relation_transformer(num_classes=11, d_ffn=64, enc_layers=2, dec_layers=3, num_queries=24, hybrid_num_proposals=30,
                                     msda_cls=OracleMSDA, self_attn_cls=OracleSelfAttention, relation_cls=OracleRelation).eval()
    net.load_state_dict(synthetic_state_dict(net.state_dict()))
    args = ([T(g[f"feat{i}"]) for i in range(4)], [T(g[f"mask{i}"]) for i in range(4)], [T(g[f"pos{i}"]) for i in range(4)])
    with torch.no_grad():
        outs = net(*args, T(g["dn_label"]), T(g["dn_box"]), T(g["attn_mask"]))
    assert all(o is None for o in outs[4:])
    for i in range(4):                              # the one-to-one branch does not depend on the mode (dropout is 0)
        np.testing.assert_allclose(outs[i].numpy(), g[f"out{i}"], rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_training_forward_backward_with_hip_modules_matches_reference(golden):
    res = _run(golden, "cuda:0")
    # forward: as for g7 (fp32 GEMMs / LayerNorms between the kernels).  Gradients: the msda backward accumulates grad_value
    # with float atomics in arbitrary order, and every gradient here is a sum over 2 images x 5 decoder passes of such terms
    _check(*res, atol=5e-4, gtol=2e-3)
