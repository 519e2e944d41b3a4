"""GPU tests of the kernels / utilities on the CALLER side of the hot path: the fused
residual-add + LayerNorm kernel against torch.nn.functional.layer_norm, and HIP-graph replay of the harness forward
against eager execution."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,C,dtype,with_res", [
    (1000, 256, torch.float32, True), (89292, 256, torch.bfloat16, True), (3600, 256, torch.float32, False),
    (7, 256, torch.bfloat16, False), (333, 192, torch.float32, True), (65, 1000, torch.bfloat16, True), (1, 64, torch.float32, True),
])
def test_add_layer_norm_vs_torch(rows, C, dtype, with_res):
    from relation_detr_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.3).to(dtype).to(DEV)
    r = (torch.randn(rows, C, generator=g)).to(dtype).to(DEV) if with_res else None
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(dtype).to(DEV)
    b = (0.1 * torch.randn(C, generator=g)).to(dtype).to(DEV)
    out = ops.add_layer_norm(x, r, w, b, 1e-5)
    s = x.float() if r is None else x.float() + r.float()
    ref = F.layer_norm(s, (C,), w.float(), b.float(), 1e-5)
    assert out.dtype == dtype and out.shape == x.shape
    if dtype == torch.float32:
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-5)
    else:       # one bf16 rounding of the output
        err = (out.float() - ref).abs()
        assert (err <= 2.0 ** -8 * ref.abs() + 1e-3).all(), err.max()


def test_add_layer_norm_3d_and_errors():
    from relation_detr_amd import _lib, ops
    x = torch.randn(2, 5, 256, device=DEV)
    w, b = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    out = ops.add_layer_norm(x, x, w, b)
    np.testing.assert_allclose(out.cpu().numpy(), F.layer_norm(2 * x, (256,)).cpu().numpy(), atol=2e-5)
    with pytest.raises(_lib.RdetrError):
        ops.add_layer_norm(x.cpu(), None, w, b)
    with pytest.raises(_lib.RdetrError):
        ops.add_layer_norm(x, x[:, :4], w, b)
    assert ops.add_layer_norm(x[:0], None, w, b).shape == (0, 5, 256)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graph_replay_matches_eager(dtype):
    """The eval forward captured into a hipGraph returns the bits of the eager run, also after the inputs change."""
    from relation_detr_amd.graph import GraphedCall
    from relation_detr_amd.transformer import build_relation_transformer, select_detections
    torch.manual_seed(0)
    shapes = [(40, 56), (20, 28), (10, 14), (5, 7)]
    net = build_relation_transformer(num_classes=17, d_ffn=128, enc_layers=2, dec_layers=2, num_queries=50,
                                     hybrid_num_proposals=60).eval().to(DEV).to(dtype)
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, "sampling_offsets"):
                m.sampling_offsets.weight.normal_(0, 0.02)
                m.attention_weights.weight.normal_(0, 0.05)
    B, L = 2, len(shapes)

    def make(seed):
        g = torch.Generator().manual_seed(seed)
        feats = [torch.randn(B, 256, h, w, generator=g).to(DEV, dtype) for h, w in shapes]
        pos = [torch.randn(B, 256, h, w, generator=g).to(DEV, dtype) for h, w in shapes]
        masks = []
        for h, w in shapes:
            m = torch.zeros(B, h, w, dtype=torch.bool)
            m[1, :, int(w * 0.8):] = True                      # right padding on image 1: valid_ratios < 1
            masks.append(m.to(DEV))
        return [*feats, *masks, *pos, torch.tensor([[400, 560], [400, 448]], device=DEV)]

    @torch.no_grad()
    def forward(*t):
        classes, coords, _, _ = net(list(t[:L]), list(t[L:2 * L]), list(t[2 * L:3 * L]))
        return select_detections(classes[-1].float(), coords[-1].float(), t[3 * L], k=20)

    a, b = make(1), make(2)
    run = GraphedCall(forward, a)
    for inputs in (a, b, a):
        eager = forward(*inputs).clone()
        replay = run(*inputs).clone()
        torch.cuda.synchronize()
        assert torch.equal(eager, replay)
