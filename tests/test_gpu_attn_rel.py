"""rdetr_relation_attention_boxes_bf16 (csrc/attn_rel.hip): decoder self-attention whose position-relation bias is generated
inside the kernel from the boxes.  Oracle = the composition the reference runs: PositionRelationEmbedding.forward
(relation_transformer.py:520-532; expected bias tensors of tests/golden/g5_relation.npz, made by the reference itself) fed as
the float attn_mask of the attention arithmetic of :452-461 (fp32 softmax(QK^T/sqrt(d) + bias) V on the bf16-rounded inputs).
Bound: the one the materialised-bias kernel is held to (2^-7 relative + 4e-3 absolute: P and the output are rounded to bf16),
although here the 64 sine features are additionally rounded to bf16 before their projection."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def _attn_reference(q, k, v, H, bias, mask, scale):
    B, N, C = q.shape
    M, d = k.shape[1], C // H
    qh = q.float().view(B, N, H, d).transpose(1, 2)
    kh = k.float().view(B, M, H, d).transpose(1, 2)
    vh = v.float().view(B, M, H, d).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if bias is not None:
        s = s + bias.view(B, H, N, M)
    if mask is not None:
        s = s.masked_fill(mask, float("-inf"))
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, N, C)


def _qkv(B, N, M, seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda n: torch.randn(B, n, 256, generator=g).to(torch.bfloat16).to(DEV)
    return mk(N), mk(M), mk(M)


def _close(out, ref):
    err = (out - ref).abs()
    assert (err <= 2.0 ** -7 * ref.abs() + 4e-3).all(), (err.max().item(), ref.abs().max().item())


@pytest.mark.parametrize("which", ["", "tiny_"])
def test_generated_bias_matches_reference_composition_on_g5(golden, which):
    """Boxes, projection and EXPECTED BIAS from the reference (g5: typical boxes and 1e-4-sized boxes, N1 = 23 != N2 = 31)."""
    from relation_detr_amd import ops
    g = golden("g5_relation.npz")
    src, tgt = T(g[which + "src"]).to(DEV), T(g[which + "tgt"]).to(DEV)
    w, b = T(g["proj_weight"]).to(DEV), T(g["proj_bias"]).to(DEV)
    bias = T(g["bias_tiny" if which else "bias"]).to(DEV)
    q, k, v = _qkv(2, 23, 31, 5)
    out = ops.relation_attention_boxes(q, k, v, 8, src, tgt, w, b, None, 32 ** -0.5).float()
    _close(out, _attn_reference(q, k, v, 8, bias.flatten(0, 1), None, 32 ** -0.5))
    # projection without its bias term
    out = ops.relation_attention_boxes(q, k, v, 8, src, tgt, w, None, None, 32 ** -0.5).float()
    from oracle import torch_ref
    nb = torch_ref.relation_bias(src.cpu(), tgt.cpu(), w.cpu(), None).to(DEV)
    _close(out, _attn_reference(q, k, v, 8, nb.flatten(0, 1), None, 32 ** -0.5))


def _boxes(B, N, seed, tiny=0):
    g = torch.Generator().manual_seed(seed)
    b = torch.cat([torch.rand(B, N, 2, generator=g), torch.rand(B, N, 2, generator=g) * 0.49 + 0.01], -1)
    if tiny:
        b[:, :tiny, 2:] = 1e-4
    return b


@pytest.mark.parametrize("B,N,M,with_mask", [
    (4, 900, 900, False),           # BASELINE.json configs[1] / [3]: 900 queries, full size
    (2, 300, 300, False),           # 300 queries
    (1, 37, 37, False),             # partial query tile, partial key chunk
    (2, 130, 97, True),             # N != M, boolean visibility mask (denoising, relation_transformer.py:373-374)
    (1, 64, 1100, True),            # more keys than queries
    (3, 1, 5, False),
])
def test_generated_bias_matches_oracle_bias_and_materialised_kernel(B, N, M, with_mask):
    from oracle import torch_ref
    from relation_detr_amd import ops
    g = torch.Generator().manual_seed(N + 3 * M)
    src, tgt = _boxes(B, N, N, tiny=min(3, N)), _boxes(B, M, M + 1)
    w = (torch.rand(8, 64, 1, 1, generator=g) * 2 - 1) * 0.4            # a few times the default Conv2d init: biases up to ~5
    b = (torch.rand(8, generator=g) * 2 - 1) * 0.3
    q, k, v = _qkv(B, N, M, N * 7 + M)
    mask = None
    if with_mask:
        mask = torch.rand(N, M, generator=g) < 0.3
        mask[:, 0] = False
    bias = torch.cat([torch_ref.relation_bias(src[i:i + 1], tgt[i:i + 1], w, b) for i in range(B)]).flatten(0, 1).to(DEV)
    dmask = None if mask is None else mask.to(DEV)
    out = ops.relation_attention_boxes(q, k, v, 8, src.to(DEV), tgt.to(DEV), w.to(DEV), b.to(DEV), dmask, 32 ** -0.5).float()
    assert out.shape == (B, N, 256) and torch.isfinite(out).all()
    _close(out, _attn_reference(q, k, v, 8, bias, dmask, 32 ** -0.5))
    # and against the two-kernel route it replaces (fp32 bias kernel -> attention kernel reading it)
    two = ops.relation_attention(q, k, v, 8, ops.relation_bias(src.to(DEV), tgt.to(DEV), w.to(DEV), b.to(DEV)).flatten(0, 1),
                                 dmask, 32 ** -0.5).float()
    assert (out - two).abs().max().item() <= 2.0 ** -6 * max(1.0, two.abs().max().item())


def test_strided_views_and_fully_masked_row():
    from oracle import torch_ref
    from relation_detr_amd import ops
    B, N = 2, 70
    g = torch.Generator().manual_seed(3)
    qk = torch.randn(B, N, 512, generator=g).to(torch.bfloat16).to(DEV)          # packed in-projection output
    v = torch.randn(B, N, 256, generator=g).to(torch.bfloat16).to(DEV)
    q, k = qk[..., :256], qk[..., 256:]
    boxes = _boxes(B, N, 11)
    w, b = (torch.rand(8, 64, generator=g) - 0.5) * 0.5, torch.zeros(8)
    mask = torch.zeros(N, N, dtype=torch.bool)
    mask[:, 5] = True
    mask[11, :] = True                                    # a fully masked row -> NaN like torch.softmax
    out = ops.relation_attention_boxes(q, k, v, 8, boxes.to(DEV), boxes.to(DEV), w.to(DEV), b.to(DEV), mask.to(DEV)).float()
    bias = torch_ref.relation_bias(boxes, boxes, w, b).flatten(0, 1).to(DEV)
    ref = _attn_reference(q.contiguous(), k.contiguous(), v, 8, bias, mask.to(DEV), 32 ** -0.5)
    dead = torch.isnan(ref)
    assert dead.any() and torch.equal(torch.isnan(out), dead)
    _close(out[~dead], ref[~dead])


def test_unsupported_configurations_are_refused():
    from relation_detr_amd import _lib, ops
    q, k, v = _qkv(1, 16, 16, 0)
    boxes = _boxes(1, 16, 0).to(DEV)
    with pytest.raises(_lib.RdetrError):                                         # 4 heads of 64
        ops.relation_attention_boxes(q, k, v, 4, boxes, boxes, torch.zeros(4, 64, device=DEV), None)
    with pytest.raises(_lib.RdetrError):                                         # 32 sine features per coordinate
        ops.relation_attention_boxes(q, k, v, 8, boxes, boxes, torch.zeros(8, 128, device=DEV), None, num_pos_feats=32)


def test_decoder_takes_the_generated_bias_and_agrees_with_the_materialised_route(monkeypatch):
    """RelationTransformerDecoder in bf16 eval hands its layers a DeferredRelationBias; options.rel_fused = False keeps the reference's
    sequence (bias tensor -> masked_fill_ -> attn_mask).  Same boxes, same weights: the decoder outputs agree to bf16 noise."""
    from relation_detr_amd import ops
    from relation_detr_amd.transformer import build_relation_transformer
    torch.manual_seed(0)
    net = build_relation_transformer(num_classes=17, d_ffn=128, enc_layers=1, dec_layers=3, num_queries=50).to(DEV).to(torch.bfloat16).eval()
    shapes = [(20, 28), (10, 14), (5, 7), (3, 4)]
    g = torch.Generator().manual_seed(1)
    feats = [torch.randn(2, 256, h, w, generator=g).to(DEV).to(torch.bfloat16) for h, w in shapes]
    pos = [torch.randn(2, 256, h, w, generator=g).to(DEV).to(torch.bfloat16) for h, w in shapes]
    masks = [torch.zeros(2, h, w, dtype=torch.bool, device=DEV) for h, w in shapes]
    calls = []
    real = ops.relation_attention_boxes
    monkeypatch.setattr(ops, "relation_attention_boxes", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    with torch.no_grad():
        fused = net(feats, masks, pos)
        assert len(calls) == 2                                                   # layers 1 and 2 (layer 0 has no bias)
        from relation_detr_amd import options
        options.apply(net, rel_fused=False)
        plain = net(feats, masks, pos)
        assert len(calls) == 2
    for a, b_ in zip(fused[:2], plain[:2]):
        assert (a.float() - b_.float()).abs().max().item() <= 6e-2               # bf16 logits of O(1-5) through 3 layers
    assert (fused[1].float() - plain[1].float()).abs().mean().item() <= 2e-3     # boxes
