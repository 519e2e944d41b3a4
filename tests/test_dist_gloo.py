"""world_size-2 gloo rehearsal of the image-parallel path (CPU): block partition + detection gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from relation_detr_amd.dist import gather_detections, image_block


def test_image_block_partition():
    for n in (0, 1, 7, 8, 32, 33):
        for world in (1, 2, 8):
            blocks = [image_block(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        image_block(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, e = image_block(8, rank, world)
        ids = torch.arange(b, e)
        dets = torch.stack([torch.full((300, 6), float(i)) for i in ids.tolist()])
        all_d, all_i = gather_detections(dets, ids)
        ok = all_i.tolist() == list(range(8)) and all((all_d[i] == float(i)).all().item() for i in range(8))
        q.put((rank, ok, tuple(all_d.shape)))
    finally:
        dist.destroy_process_group()


def test_gather_detections_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2] == (8, 300, 6) for r in res)


def test_gather_is_identity_without_process_group():
    d, i = torch.zeros(2, 300, 6), torch.arange(2)
    out_d, out_i = gather_detections(d, i)
    assert out_d is d and out_i is i
