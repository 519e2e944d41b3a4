"""Image-parallel partitioning and the eval detection gather (one process per GPU, RCCL over xGMI).

The encoder/decoder stack has no cross-image operation, so the only parallelism the reference uses
is data parallel (accelerate DDP: main.py:106-115, test.py:71,113).  The one exchange on the eval path
is the end-of-eval gather of detections, done in the reference by pickling python dicts and two
``dist.all_gather`` calls of padded uint8 buffers (util/utils.py:79-119, called from
util/coco_eval.py:52,156-157).  Here it is a single fixed-shape all-gather of a
``[B_local, K, 6]`` fp32 tensor (x1, y1, x2, y2, score, label) plus the int64 image ids --
28.8 KB per rank at B_local = 4, K = 300: latency-bound, one ``ncclAllGather`` on the 8-GPU xGMI mesh.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def image_block(num_images: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of the images owned by ``rank``; sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(num_images, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def pad_block(dets: torch.Tensor, image_ids: torch.Tensor, b_local: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pad a rank's block to ``b_local`` images (image id -1, zero detections) so that every rank contributes the same
    shape to the fixed-shape all-gather; ``image_block`` sizes differ by at most one."""
    n = dets.shape[0]
    if n > b_local:
        raise ValueError(f"block of {n} images does not fit B_local = {b_local}")
    if n == b_local:
        return dets, image_ids
    pad_d = dets.new_zeros((b_local - n,) + tuple(dets.shape[1:]))
    pad_i = image_ids.new_full((b_local - n,), -1)
    return torch.cat([dets, pad_d], 0), torch.cat([image_ids, pad_i], 0)


def gather_detections(dets: torch.Tensor, image_ids: torch.Tensor, group=None, check_equal: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather ``dets [B_local, K, 6]`` and ``image_ids [B_local]`` from every rank (equal
    B_local on all ranks) -> ``([world*B_local, K, 6], [world*B_local])`` ordered by rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return dets, image_ids
    world = dist.get_world_size(group)
    if dets.shape[0] != image_ids.shape[0]:
        raise ValueError("gather_detections: dets and image_ids disagree on B_local")
    # a fixed-shape all-gather hangs or corrupts data on unequal inputs (image_block sizes may differ by one): agree on
    # B_local first -- one tiny all-reduce, only when the check is asked for (the bench's equal blocks skip it)
    if check_equal:
        probe = torch.tensor([dets.shape[0], -dets.shape[0]], device=dets.device, dtype=torch.int64)
        dist.all_reduce(probe, op=dist.ReduceOp.MAX, group=group)
        if int(probe[0]) != -int(probe[1]):
            raise ValueError(f"gather_detections: ranks hold between {-int(probe[1])} and {int(probe[0])} images; pad every "
                             f"rank's block to the same B_local (dist.pad_block) before the gather")
    dets = dets.contiguous()
    image_ids = image_ids.contiguous()
    out_d = dets.new_empty((world * dets.shape[0],) + tuple(dets.shape[1:]))
    out_i = image_ids.new_empty(world * image_ids.shape[0])
    if dist.get_backend(group) == "gloo":          # CPU rehearsal path used by the multi-process tests
        dist.all_gather(list(out_d.chunk(world)), dets, group=group)
        dist.all_gather(list(out_i.chunk(world)), image_ids, group=group)
    else:                                          # RCCL: one ncclAllGather each, no staging copies
        dist.all_gather_into_tensor(out_d, dets, group=group)
        dist.all_gather_into_tensor(out_i, image_ids, group=group)
    return out_d, out_i
