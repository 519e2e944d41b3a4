"""`select_detections` (relation_detr_amd/transformer.py) against a direct restatement of the reference's PostProcess
(models/bricks/post_process.py:21-44): sigmoid -> top-k over the flattened N x C scores -> `trunc` division for the box
index, modulo for the label -> cxcywh to xyxy -> scale by (w, h, w, h).

PARITY UNPINNED: the reference holds no fixture for PostProcess, and its module imports torchvision.ops.boxes, which is
not importable here; the restatement below follows the reference text line by line and the comparison is bit-exact on
hand-built inputs (ties included: torch.topk's tie order is the same kernel on both sides)."""
import pytest
import torch

from relation_detr_amd.transformer import select_detections


def _post_process_restated(logits, boxes, target_sizes, k):
    """post_process.py:21-44 with select_box_nums_for_evaluation = k, no NMS / confidence filter (the defaults of the
    relation_detr configs); box_cxcywh_to_xyxy restated (torchvision.ops.boxes._box_cxcywh_to_xyxy: cx -/+ 0.5 w)."""
    prob = logits.sigmoid()
    topk_values, topk_indexes = torch.topk(prob.view(logits.shape[0], -1), k, dim=1)
    scores = topk_values
    topk_boxes = torch.div(topk_indexes, logits.shape[2], rounding_mode="trunc")
    labels = topk_indexes % logits.shape[2]
    cx, cy, w, h = boxes.unbind(-1)
    xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)
    xyxy = torch.gather(xyxy, 1, topk_boxes.unsqueeze(-1).repeat(1, 1, 4))
    img_h, img_w = target_sizes.unbind(1)
    scale_fct = torch.stack([img_w, img_h, img_w, img_h], dim=1)
    return scores, labels, xyxy * scale_fct[:, None, :]


@pytest.mark.parametrize("B,N,C,k", [(2, 900, 91, 300), (1, 300, 91, 300), (3, 17, 5, 10), (1, 4, 3, 12)])
def test_select_detections_bit_exact(B, N, C, k):
    g = torch.Generator().manual_seed(N + C)
    logits = torch.randn(B, N, C, generator=g) * 3
    boxes = torch.cat([torch.rand(B, N, 2, generator=g), torch.rand(B, N, 2, generator=g) * 0.5 + 0.01], -1)
    sizes = torch.tensor([[800, 1333], [640, 480], [1200, 2000]][:B])
    s, l, x = _post_process_restated(logits, boxes, sizes.float(), k)
    det = select_detections(logits, boxes, sizes, k)
    assert det.shape == (B, k, 6)
    assert torch.equal(det[..., :4], x) and torch.equal(det[..., 4], s) and torch.equal(det[..., 5], l.float())


def test_select_detections_ties_and_extremes():
    # hand-built: exact ties across queries and classes, +-inf logits, the maximum label / box index
    B, N, C, k = 1, 6, 4, 8
    logits = torch.full((B, N, C), -2.0)
    logits[0, 5, 3] = 9.0            # unique best: box 5, label 3 (the last flat index)
    logits[0, 0, 0] = 1.5            # three-way tie ...
    logits[0, 2, 1] = 1.5
    logits[0, 4, 2] = 1.5
    logits[0, 1, 1] = float("inf")   # sigmoid = 1
    logits[0, 3, 0] = float("-inf")  # sigmoid = 0: never selected with k < N*C - 1
    boxes = torch.tensor([[[0.5, 0.5, 0.2, 0.4], [0.1, 0.2, 0.1, 0.1], [0.9, 0.9, 0.2, 0.2], [0.3, 0.3, 0.6, 0.6],
                           [0.25, 0.75, 0.5, 0.5], [0.6, 0.4, 0.8, 0.8]]])
    sizes = torch.tensor([[480, 640]])
    s, l, x = _post_process_restated(logits, boxes, sizes.float(), k)
    det = select_detections(logits, boxes, sizes, k)
    assert torch.equal(det[..., :4], x) and torch.equal(det[..., 4], s) and torch.equal(det[..., 5], l.float())
    assert det[0, 0, 5] == 1 and det[0, 0, 4] == 1.0                  # the +inf logit first, label 1 of box 1
    assert det[0, 1, 5] == 3 and torch.allclose(det[0, 1, :4], torch.tensor([0.2 * 640, 0.0, 1.0 * 640, 0.8 * 480]))
    assert set(det[0, 2:5, 5].tolist()) == {0.0, 1.0, 2.0}            # the three tied entries, each once
    assert (det[0, :, 4] >= det[0, :, 4].roll(-1))[:-1].all()          # scores sorted, non-increasing
