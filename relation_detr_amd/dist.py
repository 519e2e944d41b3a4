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


def gather_detections(dets: torch.Tensor, image_ids: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather ``dets [B_local, K, 6]`` and ``image_ids [B_local]`` from every rank (equal
    B_local on all ranks) -> ``([world*B_local, K, 6], [world*B_local])`` ordered by rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return dets, image_ids
    world = dist.get_world_size(group)
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
