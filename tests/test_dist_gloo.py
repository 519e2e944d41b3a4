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


@pytest.mark.gpu
def test_gather_detections_rccl_single_rank():
    """The RCCL code path (all_gather_into_tensor on device tensors) with a one-rank process group: the 8-GPU run
    belongs to the driver, this at least executes the same calls on the GPU box."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        d = torch.rand(4, 300, 6, device="cuda:0")
        i = torch.arange(4, device="cuda:0")
        out_d, out_i = gather_detections(d, i)
        torch.cuda.synchronize()
        assert out_d.shape == (4, 300, 6) and torch.equal(out_d, d) and torch.equal(out_i, i)
        dist.barrier()
        # bench.py's multi-GPU step: HIP-graph capture + replay while the RCCL process group (and its watchdog
        # thread) is alive, then the gather of the replayed detections
        from relation_detr_amd.graph import GraphedCall
        lin = torch.nn.Linear(6, 6).to("cuda:0")
        run = GraphedCall(lambda t: lin(t).relu(), [d])
        eager = lin(d).relu()
        for _ in range(3):
            got, _ = gather_detections(run(d), i)
        torch.cuda.synchronize()
        assert torch.allclose(got, eager)
    finally:
        dist.destroy_process_group()
