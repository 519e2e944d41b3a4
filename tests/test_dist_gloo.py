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


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_launcher_two_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher around it starts two fresh rank processes itself, and rank 0 prints
    ONE line with n_gpus = 2 (here over gloo with the stand-in step: the launcher, the barrier-bracketed loop, the
    per-step gather and the max-over-ranks reduction are the code the 8-GPU run uses; the hot path needs a GPU)."""
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "4"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    res = lines[0]
    assert res["n_gpus"] == 2 and res["world_size_seen"] == 2 and res["dry_run"] is True and res["gather_ok"] is True
    assert len(res["per_rank_step_ms"]) == 2 and res["value"] is None
    assert res["metric"].startswith("images/sec @ 800") and "300 queries" in res["metric"]


def test_bench_launcher_two_ranks_dry_run_focalnet_config():
    """BASELINE.json configs[4] through the same launcher path the driver will use on an 8-GPU node: `--config focalnet`
    takes its per-rank batch (B = 2) and its metric string from the config table, the per-step gather carries
    [2 x world, 300, 6] detections (util/utils.py:79-119, main.py:106-115 equivalents)."""
    rc, lines, err = _run_bench(["--config", "focalnet", "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    res = lines[0]
    assert res["n_gpus"] == 2 and res["world_size_seen"] == 2 and res["dry_run"] is True and res["gather_ok"] is True
    assert res["metric"].startswith("images/sec @ 1200") and "FocalNet-L 5-level" in res["metric"]
    cfg = res["config"]
    assert cfg["batch_per_gpu"] == 2 and cfg["global_batch"] == 4 and cfg["levels"] == 5
    assert cfg["name"] == "relation_detr_focalnet_large_lrf_fl4_1200_2000"


def test_bench_launcher_ends_all_ranks_when_one_dies():
    """A rank that exits non-zero before its first collective must not leave the launcher waiting on the other rank's
    collective timeout (poll all children, end the rest on the first failure, return non-zero) -- and the
    rendezvous is a file store in a private directory, so no TCP port is chosen and raced for."""
    import time
    t0 = time.monotonic()
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--steps", "50", "--warmup", "1", "--batch", "4"],
                                {"RDETR_BENCH_DRY_FAIL_RANK": "1"}, timeout=120)
    assert rc == 1 and not lines, (rc, lines, err[-1000:])
    assert "ranks failed" in err and "(1, 3)" in err
    assert time.monotonic() - t0 < 90                   # not the ~10-minute collective timeout


def test_bench_under_torch_distributed_run_two_ranks():
    """The driver's own launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- with N = 2 on the CPU (dry run, gloo): the env:// rendezvous torchrun
    exports is used as it is, rank 0 prints ONE line with n_gpus = 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert lines[0]["n_gpus"] == 2 and lines[0]["world_size_seen"] == 2 and lines[0]["gather_ok"] is True


def test_bench_refuses_world_size_mismatch():
    """--gpus N must agree with the ranks actually launched: a silent 1-GPU run labelled otherwise is refused."""
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                                {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert rc != 0 and not lines and "WORLD_SIZE" in err
    rc, lines, err = _run_bench(["--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "0"])
    assert rc == 0 and lines[0]["n_gpus"] == 1 and lines[0]["gather_ok"] is True


def _unequal_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from relation_detr_amd.dist import pad_block
        b, e = image_block(7, rank, world)                        # 4 + 3 images: unequal blocks
        ids = torch.arange(b, e)
        dets = torch.stack([torch.full((5, 6), float(i)) for i in ids.tolist()])
        try:
            gather_detections(dets, ids, check_equal=True)
            refused = False
        except ValueError:
            refused = True
        d2, i2 = pad_block(dets, ids, 4)
        all_d, all_i = gather_detections(d2, i2, check_equal=True)
        keep = all_i >= 0
        ok = all_i[keep].tolist() == list(range(7)) and all((all_d[keep][i] == float(i)).all().item() for i in range(7))
        q.put((rank, refused, ok))
    finally:
        dist.destroy_process_group()


def test_gather_refuses_unequal_blocks_and_pads():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_unequal_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] and r[2] for r in res), res


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
