#!/usr/bin/env python
"""Benchmark of the Relation-DETR hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--queries 900] [--batch 4] [--config r50|focalnet]

--config focalnet = BASELINE.json configs[4] per rank: relation_detr_focalnet_large_lrf_fl4_1200_2000, padded image
1216x2016, 5-level pyramid (304,504)(152,252)(76,126)(38,63)(19,32), S = 204,098 tokens, 900 queries, 2 images per GPU.

Default workload (BASELINE.json configs[1]): relation_detr_resnet50_800_1333 -- padded image 800x1344, 4-level
pyramid (100,168)(50,84)(25,42)(13,21), S = 22,323 tokens, 256 channels, 8 heads x 32, 4 points, d_ffn 2048,
6 encoder + 6 decoder layers, 91 classes, N_q two-stage queries (900 = what the reference config runs;
BASELINE.json's "300 queries" = detections kept per image, also selectable with --queries 300), batch 4 per GPU.

One step = the whole transformer stack on one batch of synthetic, HBM-resident feature pyramids
(relation_detr_amd/transformer.py: encoder with MSDA self-attention -> two-stage top-k -> decoder with
relation-biased self-attention + MSDA cross-attention + iterative box refinement -> top-300 detections).
The backbone and neck are NOT in the step (stock convolutions, out of the hot path -- SURVEY.md section 8d:
"backbone excluded, stated explicitly"); weights are random-init, inputs synthetic ("data": "synthetic").
--dtype bf16 converts the network to bfloat16 once (value / output of the MSDA core bf16, fp32 accumulate;
relation bias and softmax stay fp32); fp32 runs everything in fp32.

Multi-GPU: images are independent, so ranks take disjoint image blocks (weak scaling, `--batch` images per GPU)
with no data-path collective; the only exchange is the eval all-gather of the [B,300,6] detections over RCCL,
done once per step.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel = encoder-shape
MSDA gather; algorithmic bytes from BASELINE.md section 4 / SURVEY.md section 8d) and `cpu_baseline` (the same
stack with the oracle's PyTorch-CPU operators on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import tempfile
import time

# The dense layers of the stack are library GEMMs (hipBLASLt / rocBLAS).  PyTorch's TunableOp can pick the fastest library
# kernel per GEMM shape during the warm-up steps (about 20 shapes, ~5 s; +1.5-3.6 % images/s) -- OPT-IN
# (RDETR_BENCH_TUNABLEOP=1, set before torch is imported): `value` is measured with the library's own heuristics, the tuned
# rate is an optional side value from a child process (`value_gemm_tuned`, RDETR_BENCH_TUNED_SIDE=1; measured 923 tuned vs 975
# untuned images/s on the same box in round 3 -- tuning no longer pays).  Round 2's headline depended on tuning, and tuning is what
# faulted the GPU twice in round 2 (TunableOp undersizes its scratch copy of a strided-batched operand whose leading
# dimension exceeds its row length, DESIGN.md 5 -- the operands this package hands the library are dense now, but the
# headline should not depend on a tuner).
if os.environ.get("RDETR_BENCH_TUNABLEOP", "0") == "1":
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "30")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "5")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(tempfile.gettempdir(), f"rdetr_tunableop_{os.getuid()}_%d.csv"))

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R50_SHAPES = [(100, 168), (50, 84), (25, 42), (13, 21)]
HBM_PEAK = 8.0e12           # B/s, MI355X spec (MI355X_MICROARCH.md)
SETUP_REPLAYS = 12              # untimed replays before the contract's W warm-up steps (clock ramp after the idle capture phase)
KERNEL_REPS = 50                # timed launches of the roofline kernel; 2 x KERNEL_REPS untimed launches right before them
METRIC = "images/sec @ 800\u00d71333, 300 queries, R50 4-level; achieved HBM GB/s"      # BASELINE.json, verbatim
ROOFLINE_KERNEL = ("%s on the head-major value [B,H,S,D] the module path's value projection writes "
                   "(encoder shape, B=%d; operator form with materialised locations / weights = SURVEY 8d's bytes)")
# padded pyramid per reference config; `batch` = images per GPU
CONFIGS = {
    "r50": dict(name="relation_detr_resnet50_800_1333", shapes=R50_SHAPES, image=(800, 1333), batch=4, metric=METRIC),
    "focalnet": dict(name="relation_detr_focalnet_large_lrf_fl4_1200_2000",
                     shapes=[(304, 504), (152, 252), (76, 126), (38, 63), (19, 32)], image=(1200, 2000), batch=2,
                     metric="images/sec @ 1200\u00d72000, 900 queries, FocalNet-L 5-level (BASELINE.json configs[4], per rank); "
                            "achieved HBM GB/s"),
}


def msda_algorithmic_bytes(B, S, Nq, L, P, H, D, value_bytes):
    """value read once (or only touched rows) + loc + weights + output."""
    touched = min(S * H * D, Nq * H * L * P * 4 * D)
    return B * (touched * value_bytes + Nq * H * L * P * 2 * 4 + Nq * H * L * P * 4 + Nq * H * D * value_bytes)


def build_pyramid(B, dev, seed, dtype=torch.float32, shapes=R50_SHAPES):
    """Synthetic multi-level features / masks / position embeddings: N(0,1) features,
    all-valid masks, N(0,1) position embeddings."""
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(B, 256, h, w, generator=g).to(dev, dtype) for h, w in shapes]
    pos = [torch.randn(B, 256, h, w, generator=g).to(dev, dtype) for h, w in shapes]
    masks = [torch.zeros(B, h, w, dtype=torch.bool, device=dev) for h, w in shapes]
    return feats, masks, pos


def build_network(Nq, seed=0, num_levels=4, class_scale=3.0, exchangeable_queries=True, box_head_std=0.01, **classes):
    """Random-init RelationTransformer of the R50 / FocalNet-L config (they differ in the level count only,
    configs/relation_detr/relation_detr_focalnet_large_lrf_fl4_1200_2000.py:20-29); fresh MSDA modules have zero offset /
    attention weights (ms_deform_attn.py:268,279-280), so those get a trained-like spread to make the gather data dependent.
    Three more departures from the reference's init, none of which changes a shape, a launch or a FLOP -- they make the
    DETECTIONS of two arithmetic routes (bf16 vs fp32, replay vs eager) comparable, measured by tools/exp_separation.py
    (profiles/r03/detection_separation.txt):
    ``class_scale``: factor on the weights of every class head (encoder, decoder layers, hybrid).  At 1.0 the final scores
    span 0.04-0.08; at 3.0 they span 0.4-0.7 (logit spread ~ +-1.5 around the prior, no sigmoid saturation: at 8 the top
    scores are all 1.0 in fp32).
    ``box_head_std``: the last layer of every box head is ZERO in the reference's init (relation_transformer.py:304-305), so
    the refined boxes would equal the proposals through all six layers and the box path would never be exercised; N(0, std)
    weights give refinements of ~0.1 in logit space per layer.
    ``exchangeable_queries``: every row of ``tgt_embed`` (the content query of slot r, relation_transformer.py:117) gets the
    values of row 0.  The two-stage selection hands slot r the r-th best proposal; with N(0,1) rows every slot is a different
    random function of its proposal, so two routes that agree on the proposal SET but order near-equal scores differently
    (bf16 noise is ~20 rank spacings) produce unrelated detections.  With equal rows the decoder is permutation-equivariant
    in its proposals and a detection match measures arithmetic, not slot assignment."""
    from relation_detr_amd.transformer import build_relation_transformer
    torch.manual_seed(seed)
    net = build_relation_transformer(num_classes=91, d_ffn=2048, enc_layers=6, dec_layers=6, num_queries=Nq,
                                     hybrid_num_proposals=1500, num_levels=num_levels, **classes)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, mod in net.named_modules():
            if hasattr(mod, "sampling_offsets"):
                mod.sampling_offsets.weight.copy_(torch.randn(mod.sampling_offsets.weight.shape, generator=g) * 0.02)
                mod.attention_weights.weight.copy_(torch.randn(mod.attention_weights.weight.shape, generator=g) * 0.05)
        if exchangeable_queries:
            net.tgt_embed.weight.copy_(net.tgt_embed.weight[:1].expand_as(net.tgt_embed.weight).clone())
        if box_head_std:
            for head in (net.encoder_bbox_head, net.hybrid_bbox_head, *net.decoder.bbox_head):
                head.layers[-1].weight.copy_(torch.randn(head.layers[-1].weight.shape, generator=g) * box_head_std)
        if class_scale != 1.0:
            for head in (net.encoder_class_head, net.hybrid_class_head, *net.decoder.class_head):
                head.weight.mul_(class_scale)
    return net.eval()


def encoder_kernel_inputs(B, dev, dtype, level_shapes=R50_SHAPES):
    """Inputs of the dominant kernel at the encoder shape: pixel-centre reference points + N(0, (k/W_l)^2)
    offsets for point k = 1..4, softmaxed N(0,1) weights (SURVEY.md section 8d / BASELINE.md section 3)."""
    g = torch.Generator().manual_seed(123)
    shapes = torch.tensor(level_shapes, dtype=torch.int64)
    areas = shapes[:, 0] * shapes[:, 1]
    start = torch.cat([areas.new_zeros(1), areas.cumsum(0)[:-1]])
    S, L = int(areas.sum()), len(level_shapes)
    refs = []
    for h, w in level_shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)
    value = torch.randn(B, S, 8, 32, generator=g).to(dev).to(dtype)
    wh = shapes.flip(-1).float()
    k = torch.arange(1, 5, dtype=torch.float32).view(1, 1, 1, 1, 4, 1)
    off = torch.randn(B, S, 8, L, 4, 2, generator=g) * k / wh.view(1, 1, 1, L, 1, 2)
    off = off * float(os.environ.get("RDETR_BENCH_SPREAD", "1.0"))      # diagnostic knob, default = SURVEY 8d
    loc = (ref[None, :, None, None, None, :] + off).contiguous().to(dev)
    attn = torch.softmax(torch.randn(B, S, 8, L * 4, generator=g), -1).view(B, S, 8, L, 4).contiguous().to(dev)
    return value, shapes.to(dev), start.to(dev), loc, attn, S, L


def time_encoder_kernel(B, dev, dtype, reps=50, layout=None, level_shapes=R50_SHAPES, busy=None, algo=None):
    """Average duration of the dominant kernel from device events recorded on the stream it is launched on.
    layout None = the one the stack runs the kernel in: head-major [B,H,S,D] for bf16 (written by the value projection's
    epilogue, relation_detr_amd/ms_deform_attn.py), the reference operator's [B,S,H,D] for fp32.
    ``busy``: a callable that enqueues ~50 ms of the stack's own work.  It runs right before 2 x reps untimed and reps timed
    launches, with no host synchronisation in between, so that the kernel is timed at sustained clocks: after an idle gap (the inputs
    are built on the host) the first ~200 launches run at ramping clocks -- 135 -> 107 us over 25 ms, 119 us after a 50-ms
    pause (tools/exp_kernel_timing.py, profiles/r03/kernel_timing_vs_clock_ramp.txt); round 2's 3 warm-up + 20 timed launches
    measured the ramp."""
    import relation_detr_amd as rd
    value, shapes, start, loc, attn, S, L = encoder_kernel_inputs(B, dev, dtype, level_shapes)
    if layout is None:
        layout = "bhsd" if dtype == torch.bfloat16 else "bshd"
    if layout == "bhsd":
        value = value.permute(0, 2, 1, 3).contiguous()
    kw = {"value_layout": layout} if dtype == torch.bfloat16 else {}
    if algo is not None:                                    # name the kernel (default: what the operator picks itself)
        kw["algo"] = algo
    for _ in range(3):
        rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64, **kw)
    torch.cuda.synchronize()
    if busy is not None:
        busy()
        for _ in range(2 * reps):                           # ... and the kernel itself up to its own steady state
            rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3, S, L


def gather_kernel_name(B, Nq, L, fused, level_shapes=None):
    """Which kernel the bf16 operator on a head-major value launches at this shape (relation_detr_amd/ops.py::_resident_pays)."""
    import relation_detr_amd as rd
    if rd.ops._resident_pays(B, Nq, L, level_shapes):
        return ("msda_fwd_res_kernel<L=%d, FUSED=%s> (csrc/msda_res.hip: persistent workgroups, the coarse levels of an (image, head) "
                "plane resident in LDS, fine levels through the buffer descriptor)" % (L, "true" if fused else "false"))
    return "msda_fwd_qrun_kernel<bf16, L=%d, FUSED=%s> (csrc/msda_fwd.hip: query-run kernel, every corner row through the buffer descriptor)" % (
        L, "true" if fused else "false")


def _timed_launches(fn, reps, busy=None):
    """Average duration of `fn` from device events on the current stream: 3 launches, a synchronize, optionally `busy()` and
    2 x reps untimed launches (sustained clocks), then reps timed ones."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if busy is not None:
        busy()
        for _ in range(2 * reps):
            fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def time_in_stack_kernel(B, dev, level_shapes=R50_SHAPES, reps=50, busy=None):
    """The kernel instantiation the timed region really launches for the encoder's MSDA: the FUSED-producer
    form msda_fwd_qrun_kernel<bf16, L, FUSED = true> on the head-major value -- raw bf16 offsets / logits and fp32 reference
    points in, softmax and location arithmetic inside the kernel (relation_detr_amd/ms_deform_attn.py), at the encoder shape.
    Returns (seconds, bytes it moves, SURVEY 8d's operator-form bytes)."""
    import relation_detr_amd as rd
    value, shapes, start, loc, attn, S, L = encoder_kernel_inputs(B, dev, torch.bfloat16, level_shapes)
    vh = value.permute(0, 2, 1, 3).contiguous()
    g = torch.Generator().manual_seed(321)
    wh = shapes.flip(-1).float().cpu()
    k = torch.arange(1, 5, dtype=torch.float32).view(1, 1, 1, 1, 4, 1)
    off = (torch.randn(B, S, 8, L, 4, 2, generator=g) * k).to(torch.bfloat16).to(dev)       # pixels: the module divides by (W_l, H_l)
    logits = torch.randn(B, S, 8, L * 4, generator=g).to(torch.bfloat16).to(dev)
    refs = []
    for h, w in level_shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(B, S, L, 2).contiguous().to(dev)
    del wh, loc, attn
    fn = lambda: rd.ops.ms_deform_attn_forward_fused(vh, shapes, start, off, logits, ref, value_layout="bhsd")
    t = _timed_launches(fn, reps, busy)
    moved = B * (S * 256 * 2 + S * 8 * L * 4 * 2 * 2 + S * 8 * L * 4 * 2 + S * L * 2 * 4 + S * 256 * 2)
    return t, moved, msda_algorithmic_bytes(B, S, S, L, 4, 8, 32, 2)


def time_relation_kernels(B, dev, N=900, reps=30, busy=None):
    """SURVEY 8d's other two named kernels at B images x N queries (relation_transformer.py:493-532, 452-461): the materialised
    relation bias (rdetr_relation_bias_ws_f32: HBM-write bound, B * 8 * N * N * 4 bytes) and the fused decoder self-attention
    that generates the bias inside the kernel (rdetr_relation_attention_boxes_bf16: VALU / transcendental bound, time only)."""
    import relation_detr_amd as rd
    g = torch.Generator().manual_seed(11)
    def boxes():
        return torch.cat([torch.rand(B, N, 2, generator=g), torch.rand(B, N, 2, generator=g) * 0.49 + 0.01], -1).to(dev)
    src, tgt = boxes(), boxes()
    w = ((torch.rand(8, 64, generator=g) * 2 - 1) * (6.0 / 72) ** 0.5).to(dev)
    b = ((torch.rand(8, generator=g) * 2 - 1) * 0.125).to(dev)
    t_bias = _timed_launches(lambda: rd.ops.relation_bias(src, tgt, w, b), reps, busy)
    q = (torch.randn(B, N, 256, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    kk = (torch.randn(B, N, 256, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    v = torch.randn(B, N, 256, generator=g).to(torch.bfloat16).to(dev)
    t_attn = _timed_launches(lambda: rd.ops.relation_attention_boxes(q, kk, v, 8, src, tgt, w, b), reps, busy)
    return t_bias, B * 8 * N * N * 4, t_attn


def time_sweep_kernel(B, dev, level_shapes=R50_SHAPES, reps=30, busy=None):
    """The opt-in LDS-sourced kernel (csrc/msda_sweep.hip) on the same inputs as the roofline kernel, for the record."""
    import relation_detr_amd as rd
    value, shapes, start, loc, attn, S, L = encoder_kernel_inputs(B, dev, torch.bfloat16, level_shapes)
    vh = value.permute(0, 2, 1, 3).contiguous()
    return _timed_launches(lambda: rd.ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo="sweep"), reps, busy)


def pmc_traffic(dtype_name, B, config="r50"):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read inside a timed run (separate
    `rocprofv3 --pmc` passes, MI355X_MICROARCH.md), so the live line carries the figure of the last committed PMC pass
    of this kernel (tools/profile_msda.py -> profiles/r02/pmc_msda_fwd_B4_encoder.json) TOGETHER with the commit it was
    taken at; `traffic` is null when no pass of the current default kernel is on file.
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE doubled as the guide prescribes for gfx950."""
    name, batch = ("pmc_msda_fwd_B4_encoder.json", 4) if config == "r50" else ("pmc_msda_fwd_B2_focalnet.json", 2)
    path = next((q for q in (os.path.join(ROOT, "profiles", r, name) for r in ("r04", "r03", "r02")) if os.path.exists(q)), None)
    if B != batch or path is None:
        return {"traffic": None}
    rec = json.load(open(path))
    c = rec.get("per_launch_mean", {}).get(dtype_name)
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return {"traffic": None}
    return {"traffic": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024, "traffic_profiled_at": rec.get("commit", "unknown"),
            "traffic_kernel": rec.get("kernel", "unknown")}


def cpu_baseline(Nq, budget_s=25.0, cfg=None):
    """The same transformer stack with the oracle's PyTorch-CPU operators (per-level grid_sample + stack + weighted
    sum, materialised relation embedding, softmax attention) on the host cores: ONE image per pass, repeated until
    ~budget_s of CPU work; returns images/s."""
    from oracle.cpu_modules import OracleMSDA, OracleRelation, OracleSelfAttention
    from relation_detr_amd.transformer import select_detections
    # the GPU box exposes 256 logical CPUs but a 1-GPU job owns a 16-core share; more threads than that
    # oversubscribe (measured: 62 s/image with 256 threads)
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("RDETR_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    cfg = cfg or CONFIGS["r50"]
    net = build_network(Nq, 0, num_levels=len(cfg["shapes"]), msda_cls=OracleMSDA, self_attn_cls=OracleSelfAttention,
                        relation_cls=OracleRelation)
    feats, masks, pos = build_pyramid(1, "cpu", 7, shapes=cfg["shapes"])
    sizes = torch.tensor([list(cfg["image"])])

    def one_image():
        with torch.no_grad():
            classes, coords = net(feats, masks, pos)[:2]
            return select_detections(classes[-1], coords[-1], sizes)

    if len(cfg["shapes"]) == 4:
        one_image()                                         # warm (the 5-level image alone is ~30 s of CPU work: timed cold)
    n, t0 = 0, time.perf_counter()
    while True:
        one_image()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 16:
            break
    return {"value": n / el, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} x 1 image (B=1) of the same enc+dec stack, torch {torch.__version__} CPU kernels, fp32, "
                      f"{cores} threads, hot ops = oracle/torch_ref.py (reference fallback path, op for op)"}


def detection_drift(det_a, det_b, iou_thr=0.9):
    """How far two detection sets [B,K,6] = (x1,y1,x2,y2,score,label) of the same images are apart: fraction of the
    detections of `det_b` (the reference side) that `det_a` reproduces with the same label and IoU >= iou_thr, the
    largest box-coordinate distance (pixels) and score distance over the matched pairs.  Used for the bf16-vs-fp32 figure
    in the bench line and by tests/test_gpu_fullsize.py."""
    import torch
    B, K, _ = det_a.shape
    a, b = det_a.float(), det_b.float()
    x1 = torch.maximum(a[:, :, None, 0], b[:, None, :, 0]); y1 = torch.maximum(a[:, :, None, 1], b[:, None, :, 1])
    x2 = torch.minimum(a[:, :, None, 2], b[:, None, :, 2]); y2 = torch.minimum(a[:, :, None, 3], b[:, None, :, 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    area_a = ((a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]))[:, :, None]
    area_b = ((b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]))[:, None, :]
    iou = inter / (area_a + area_b - inter).clamp(min=1e-9)
    iou = torch.where(a[:, :, None, 5] == b[:, None, :, 5], iou, torch.zeros_like(iou))     # same label only
    best, idx = iou.max(1)                                       # per reference detection: best candidate of det_a
    ok = best >= iou_thr
    cand = torch.gather(a, 1, idx[..., None].expand(-1, -1, 6))
    dbox = (cand[..., :4] - b[..., :4]).abs().max(-1)[0]
    dscore = (cand[..., 4] - b[..., 4]).abs()
    n_ok = int(ok.sum())
    return {"matched_frac": n_ok / float(B * K), "iou_thr": iou_thr,
            "max_box_dist_px": float(dbox[ok].max()) if n_ok else None,
            "max_score_dist": float(dscore[ok].max()) if n_ok else None}


def init_process_group(backend, rank, world, **kw):
    """Rendezvous of the ranks.  Launched by torch.distributed.run (or any launcher that exports MASTER_ADDR / MASTER_PORT):
    env://, as the driver's contract says.  Launched by this script's own launcher (`launch_ranks`) or as a single forced rank:
    a FILE store in a fresh private directory -- no TCP port is picked, so there is no window in which another process
    can take a port between choosing it and binding it."""
    path = os.environ.get("RDETR_BENCH_INIT_FILE")
    if path is None and "MASTER_PORT" not in os.environ:
        if world != 1:
            raise SystemExit("[bench] no rendezvous: set MASTER_ADDR / MASTER_PORT (torch.distributed.run does) or start the "
                             "ranks with `python bench.py --gpus N`")
        path = os.path.join(tempfile.mkdtemp(prefix="rdetr_bench_"), "store")
    if path is not None:
        dist.init_process_group(backend, init_method="file://" + path, rank=rank, world_size=world, **kw)
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)


def launch_ranks(n, argv, deadline_s=None):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes of this script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE and the rendezvous file in their environment), BEFORE anything in this process touches
    the GPU.  All children are polled together: the first rank that exits non-zero (or the optional deadline) ends the
    others at once -- they would otherwise sit in a collective until the RCCL timeout -- and the launcher returns 1.
    The parent never initialises HIP and is never replaced by another program (the reference's launcher for the same job is
    `accelerate launch`, test.py:71-113)."""
    import shutil
    import subprocess
    store_dir = tempfile.mkdtemp(prefix="rdetr_bench_")
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_PORT",)}
    env.update(WORLD_SIZE=str(n), RDETR_BENCH_INIT_FILE=os.path.join(store_dir, "store"),
               HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), RDETR_BENCH_CHILD="1")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=e))
    t0 = time.monotonic()
    failed = None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"ranks failed (rank, exit code): {bad}"
                break
            if all(c == 0 for c in codes):
                break
            if deadline_s is not None and time.monotonic() - t0 > deadline_s:
                failed = f"deadline of {deadline_s:.0f} s passed with ranks {[r for r, c in enumerate(codes) if c is None]} still running"
                break
            time.sleep(0.05)
    finally:
        for p in procs:                                     # the exact processes this function started
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        shutil.rmtree(store_dir, ignore_errors=True)
    if failed:
        print(f"[bench] {failed}; remaining ranks ended", file=sys.stderr)
        return 1
    return 0


def dry_run(args, world, rank):
    """Launcher + collectives rehearsal WITHOUT a GPU (`--dry-run`, gloo): every rank runs the bench's distributed
    scaffolding -- process group, barrier-bracketed timed loop, per-step detection gather, max-over-ranks reduction,
    rank-0 JSON line -- around a stand-in step that only fabricates a [B,300,6] tensor.  It measures nothing about the
    hot path (the product has no CPU path) and says so in the line; tests/test_dist_gloo.py drives it with 2 ranks."""
    from relation_detr_amd.dist import gather_detections
    if world > 1 or os.environ.get("RDETR_BENCH_FORCE_DIST") == "1":
        init_process_group("gloo", rank, world)
    if os.environ.get("RDETR_BENCH_DRY_FAIL_RANK") == str(rank):       # test hook: this rank dies before its first collective
        raise SystemExit(3)
    B = args.batch
    ids = torch.arange(B) + rank * B

    def step():
        dets = torch.full((B, 300, 6), float(rank))
        return gather_detections(dets, ids, check_equal=True)

    for _ in range(args.warmup):
        step()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        all_d, all_i = step()
    if dist.is_initialized():
        dist.barrier()
    el_local = time.perf_counter() - t0
    el, per_rank = el_local, [el_local]
    if dist.is_initialized():
        t = torch.tensor([el_local], dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [g.item() for g in gathered]
        el = max(per_rank)
    ok = all_i.tolist() == list(range(world * B)) and all(bool((all_d[i] == float(i // B)).all()) for i in range(world * B))
    if rank == 0:
        print(json.dumps({
            "metric": CONFIGS[args.config]["metric"], "value": None, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "none (dry run)", "dry_run": True, "gather_ok": ok,
            "world_size_seen": dist.get_world_size() if dist.is_initialized() else 1,
            "per_rank_step_ms": [t / max(args.steps, 1) * 1e3 for t in per_rank],
            "config": {"workload": "DRY RUN: launcher + gloo collectives only, no hot-path work, value is null",
                       "parallelism": f"image-parallel x{world}", "batch_per_gpu": B, "global_batch": B * world,
                       "name": CONFIGS[args.config]["name"], "levels": len(CONFIGS[args.config]["shapes"])}}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0 if ok else 1


_T0 = time.perf_counter()


def note(msg):
    """progress on stderr (the one JSON line is the only thing on stdout)"""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def timed_loop(run, inputs, steps, warmup):
    for _ in range(warmup):
        run(*inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run(*inputs)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--queries", type=int, default=900)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: 4 for r50, 2 for focalnet)")
    ap.add_argument("--config", default="r50", choices=sorted(CONFIGS), help="r50 = BASELINE.json configs[1] (default); "
                    "focalnet = configs[4] per rank (5 levels, 1200x2000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every kernel from Python instead of replaying a HIP graph")
    ap.add_argument("--no-extras", action="store_true", help="skip the 300-query / fp32 / drift side measurements")
    ap.add_argument("--dump-dets", default=None, help="save the last step's detections [B,300,6] to this file (torch.save)")
    ap.add_argument("--dry-run", action="store_true", help="launcher + gloo collectives only, no GPU work (CPU rehearsal)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    cfg = CONFIGS[args.config]
    if args.batch is None:
        args.batch = cfg["batch"]
    tuned = os.environ.get("PYTORCH_TUNABLEOP_ENABLED", "0") == "1"

    # ---- one process per GPU -------------------------------------------------------------------------------------------
    # Launched by torch.distributed.run (RANK / WORLD_SIZE in the environment): this process is one rank.  Launched bare
    # with --gpus N > 1: this process is the launcher, it starts the N ranks and exits with their status.
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N, or torch.distributed.run --nproc-per-node N bench.py --gpus N)")
    if args.dry_run:
        raise SystemExit(dry_run(args, world, rank))

    # RDETR_BENCH_FORCE_DIST=1 runs the multi-rank code path (RCCL group, barrier, gather, max-reduce) with ONE rank, so that
    # it can be exercised on a one-GPU box
    use_dist = world > 1 or os.environ.get("RDETR_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        init_process_group("nccl", rank, world, device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from relation_detr_amd import _lib
    from relation_detr_amd.dist import gather_detections
    from relation_detr_amd.graph import GraphedCall, ImageGroups
    from relation_detr_amd.transformer import select_detections
    _lib.load()                                             # fail loudly if the HIP library is missing

    B, Nq = args.batch, args.queries
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    feats, masks, pos = build_pyramid(B, dev, seed=1000 + rank, dtype=dtype, shapes=cfg["shapes"])   # each rank owns its image block
    sizes = torch.tensor([list(cfg["image"])] * B, device=dev)
    img_ids = torch.arange(B, device=dev) + rank * B
    L = len(feats)

    # The images of a batch are independent, so the batch runs as `nstreams` image groups on parallel HIP streams, forked and
    # joined inside the captured graph (relation_detr_amd/graph.py::ImageGroups): the gather-bound kernels of one group
    # overlap the MFMA-bound GEMMs of the other.  Same kernels, same results per image.  RDETR_BENCH_STREAMS=1: one stream.
    # fp32: ONE group.  Two fp32 image groups side by side stop making progress on this image -- eagerly and under capture alike,
    # every piece of the stack and the whole one-group stack capture and replay fine (tools/exp_fp32_capture.py, DESIGN.md 5:
    # what co-runs are the library's fp32 GEMM kernels of the two groups).  No configuration of this script may hang a GPU, so
    # an explicit RDETR_BENCH_STREAMS > 1 is refused for fp32.
    nstreams = int(os.environ.get("RDETR_BENCH_STREAMS", "2" if B % 2 == 0 and args.dtype == "bf16" else "1"))
    if nstreams < 1 or B % nstreams:
        raise SystemExit("RDETR_BENCH_STREAMS must divide --batch")
    if args.dtype == "fp32" and nstreams > 1:                # relation_detr_amd.graph.ImageGroups refuses it too
        raise SystemExit("[bench] fp32 runs as one image group (two fp32 groups side by side hang on this image, DESIGN.md 5)")

    def make_runner(queries, net_dtype, inputs, groups=None):
        """(callable, launch mode) of the whole stack + top-300 detections for one network configuration."""
        net = build_network(queries, 0, num_levels=L).to(dev).to(net_dtype)           # same weights on every rank

        @torch.no_grad()
        def forward_images(*t):                             # device tensors in and out
            classes, coords = net(list(t[:L]), list(t[L:2 * L]), list(t[2 * L:3 * L]))[:2]
            return select_detections(classes[-1].float(), coords[-1].float(), t[3 * L])

        fwd = ImageGroups(forward_images, groups or nstreams, device=dev)
        if not args.no_graph:                               # same kernels, one hipGraph launch per batch (graph.py)
            try:
                return GraphedCall(fwd, inputs), "hipGraph replay"
            except RuntimeError as e:                       # capture refused by the runtime: enqueue from Python instead
                print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {str(e)[:200]}); running eagerly", file=sys.stderr)
                torch.cuda.synchronize()
        return fwd, "python"

    flat_inputs = [*feats, *masks, *pos, sizes]
    note("building the stack + capture")
    run, launch = make_runner(Nq, dtype, flat_inputs)
    note(f"launch = {launch}; warm-up")

    def step():
        dets = run(*flat_inputs)
        if use_dist:
            gather_detections(dets, img_ids)                # eval path: one RCCL all-gather per step
        return dets

    # Set-up, not part of the contract's W + K steps: ~50 ms of replays so that the W warm-up steps and the K timed ones run at
    # sustained clocks (after the idle capture phase the first ~25 ms of any load run at ramping clocks, DESIGN.md 4.1 "Round 3").
    for _ in range(SETUP_REPLAYS):
        run(*flat_inputs)
    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dets_main = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    el_local = time.perf_counter() - t0
    el, per_rank = el_local, [el_local]
    if use_dist:
        t = torch.tensor([el_local], device=dev, dtype=torch.float64)
        allt = torch.zeros(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, t)
        per_rank = allt.tolist()
        el = max(per_rank)                                  # MAX over ranks

    note(f"timed loop done: {world * B * args.steps / el:.1f} images/s; dominant kernel")
    def busy():                                             # ~50 ms of the stack itself, enqueued without a host sync
        for _ in range(max(4, int(0.05 * args.steps / max(el, 1e-6)))):
            run(*flat_inputs)

    t_kernel_cold = time_encoder_kernel(B, dev, dtype, reps=20, level_shapes=cfg["shapes"], busy=None)[0]     # round-2 protocol: 3 warm-up + 20
    t_kernel, S, L = time_encoder_kernel(B, dev, dtype, reps=KERNEL_REPS, level_shapes=cfg["shapes"], busy=busy)
    t_kernel_bshd = time_encoder_kernel(B, dev, dtype, layout="bshd", level_shapes=cfg["shapes"], busy=busy)[0] if args.dtype == "bf16" else None
    t_kernel_qrun = time_encoder_kernel(B, dev, dtype, level_shapes=cfg["shapes"], busy=busy, algo="direct")[0] if args.dtype == "bf16" else None
    t_kernel_fp32 = time_encoder_kernel(B, dev, torch.float32, level_shapes=cfg["shapes"], busy=busy)[0] if args.dtype == "bf16" else None
    in_stack = relation = sweep_ms = None
    if args.dtype == "bf16" and rank == 0:
        note("in-stack gather instantiation, relation kernels")
        in_stack = {"kernel": "fused-producer form on the head-major value (what the timed region launches for the encoder's MSDA: raw bf16 "
                              "offsets / logits + fp32 reference points in, softmax and location arithmetic inside the kernel), encoder "
                              "shape, isolated; per batch size the kernel the operator picks: see `kernel` of each entry"}
        for bb in sorted({B, max(1, B // max(1, nstreams))}, reverse=True):      # the whole batch, and one image group of it
            t_f, moved, survey = time_in_stack_kernel(bb, dev, level_shapes=cfg["shapes"], busy=busy)
            in_stack["B%d" % bb] = {"kernel": gather_kernel_name(bb, S, L, True, cfg["shapes"]), "kernel_ms": t_f * 1e3, "bytes_moved": moved, "frac_of_bytes_moved": moved / t_f / HBM_PEAK,
                                    "survey_8d_bytes": survey, "frac_of_survey_bytes": survey / t_f / HBM_PEAK}
        t_bias, bias_bytes, t_attn = time_relation_kernels(B, dev, N=Nq, busy=busy)
        relation = {"bias_materialised": {"kernel": "relation_bias_kernel via rdetr_relation_bias_ws_f32, B=%d, N=%d "
                                                    "(relation_transformer.py:493-532)" % (B, Nq), "bound": "hbm (write)",
                                          "kernel_ms": t_bias * 1e3, "algorithmic_bytes": bias_bytes,
                                          "achieved": bias_bytes / t_bias / 1e9, "frac": bias_bytes / t_bias / HBM_PEAK},
                    "attention_generating_the_bias": {"kernel": "relation_attention_boxes_kernel via rdetr_relation_attention_boxes_bf16, "
                                                                "B=%d, N=%d (relation_transformer.py:452-461 with :493-532 inside)" % (B, Nq),
                                                      "bound": "valu / transcendental (time only)", "kernel_ms": t_attn * 1e3}}
        if args.config == "r50":
            try:
                sweep_ms = time_sweep_kernel(B, dev, level_shapes=cfg["shapes"], busy=busy) * 1e3
            except RuntimeError as e:
                print(f"[bench] sweep kernel skipped ({type(e).__name__}: {str(e)[:120]})", file=sys.stderr)
    note("side measurements")
    alg = msda_algorithmic_bytes(B, S, S, L, 4, 8, 32, 2 if args.dtype == "bf16" else 4)

    # Side measurements on rank 0 at N = 1 (their own short timed loops; never part of `value`):
    #  * BASELINE.json quotes the metric "@ 300 queries" while the reference config of that name runs 900 two-stage queries
    #    and keeps 300 detections (what `value` measures): the same stack with 300 two-stage queries -> value_300_queries;
    #  * the reference's gather is fp32 even under bf16 autocast (ms_deform_attn.py:360): the fp32 stack's rate, and how far
    #    the bf16 detections are from the fp32 ones on the same images.
    value_300 = fp32_ips = drift = value_tuned = None
    extras = world == 1 and rank == 0 and not args.no_extras and os.environ.get("RDETR_BENCH_ALT300", "1") != "0"
    if extras and Nq != 300:
        try:
            note("300-query variant")
            run300, _ = make_runner(300, dtype, flat_inputs)
            t300, _ = timed_loop(run300, flat_inputs, 10, 3)
            value_300 = B * 10 / t300
            del run300
        except RuntimeError as e:
            print(f"[bench] 300-query variant skipped ({type(e).__name__}: {str(e)[:160]})", file=sys.stderr)
    # * serving headroom: TWO batches of B images in flight (two graphs on two streams, consecutive batches alternate between
    #   them), so that one batch's decoder -- a latency chain that leaves the chip idle -- overlaps the next batch's encoder.
    #   Whole-job throughput only: a batch's latency doubles.  Never `value`: that stays one batch per step.
    value_two_in_flight = None
    if extras and args.dtype == "bf16" and launch == "hipGraph replay":
        try:
            note("two batches in flight")
            f2, m2, p2 = build_pyramid(B, dev, seed=2000 + rank, dtype=dtype, shapes=cfg["shapes"])
            flat2 = [*f2, *m2, *p2, sizes]
            # each batch as ONE image group: with two batches in flight the streams already are the groups (two graphs of two
            # groups each: 953 images/s, below `value`; two graphs of one group: 1,112 -- tools/exp_chain_overlap.py)
            run1, _ = make_runner(Nq, dtype, flat_inputs, groups=1)
            run2, _ = make_runner(Nq, dtype, flat2, groups=1)
            lanes = ((run1, flat_inputs, torch.cuda.Stream(device=dev)), (run2, flat2, torch.cuda.Stream(device=dev)))
            for _, _, s_ in lanes:
                s_.wait_stream(torch.cuda.current_stream())

            def batches(n):
                for i in range(n):
                    r_, f_, s_ = lanes[i % 2]
                    with torch.cuda.stream(s_):
                        r_(*f_)
            nb = 2 * max(5, args.steps // 2)
            batches(nb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batches(nb)
            torch.cuda.synchronize()
            value_two_in_flight = B * nb / (time.perf_counter() - t0)
            del run1, run2, lanes
        except RuntimeError as e:
            print(f"[bench] two-batches-in-flight variant skipped ({type(e).__name__}: {str(e)[:160]})", file=sys.stderr)
            torch.cuda.synchronize()
    if extras and args.dtype == "bf16":
        # In a CHILD process: the same script with --dtype fp32 on the same synthetic images (same seeds), its detections dumped
        # for the comparison (enqueued from Python: see the fp32 note at the top of main).  A child that does not finish is
        # ended by its timeout and the two fields stay null.
        import subprocess
        dump = os.path.join(tempfile.gettempdir(), f"rdetr_bench_fp32_dets_{os.getpid()}.pt")
        note("fp32 side run (child process)")
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RDETR_BENCH_CHILD")}
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--dtype", "fp32", "--steps", "5", "--warmup", "2",
                                 "--queries", str(Nq), "--batch", str(B), "--config", args.config, "--no-extras",
                                 "--no-cpu-baseline", "--dump-dets", dump],
                                env=env, capture_output=True, text=True, timeout=150)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode == 0 and line:
                fp32_ips = json.loads(line[-1])["value"]
                dets32 = torch.load(dump, weights_only=True).to(dev)
                drift = detection_drift(dets_main.float(), dets32, iou_thr=0.9)
            else:
                print(f"[bench] fp32 side run failed (exit {cp.returncode}): {cp.stderr[-300:]}", file=sys.stderr)
        except (subprocess.TimeoutExpired, RuntimeError, OSError) as e:
            print(f"[bench] fp32 side run skipped ({type(e).__name__})", file=sys.stderr)
        finally:
            if os.path.exists(dump):
                os.remove(dump)

    if extras and args.dtype == "bf16" and not tuned and os.environ.get("RDETR_BENCH_TUNED_SIDE", "0") == "1":
        # The same bf16 run with PyTorch's TunableOp choosing the library GEMM kernels, in a child process (a tuner fault
        # cannot take the headline down): side value only.
        import subprocess
        note("GEMM-tuned side run (child process)")
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RDETR_BENCH_CHILD")}
            env["RDETR_BENCH_TUNABLEOP"] = "1"
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--dtype", "bf16", "--steps", str(args.steps),
                                 "--warmup", str(max(args.warmup, 5)), "--queries", str(Nq), "--batch", str(B), "--config",
                                 args.config, "--no-extras", "--no-cpu-baseline"],
                                env=env, capture_output=True, text=True, timeout=200)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode == 0 and line:
                value_tuned = json.loads(line[-1])["value"]
            else:
                print(f"[bench] tuned side run failed (exit {cp.returncode}): {cp.stderr[-300:]}", file=sys.stderr)
        except (subprocess.TimeoutExpired, OSError) as e:
            print(f"[bench] tuned side run skipped ({type(e).__name__})", file=sys.stderr)

    if rank == 0 and args.dump_dets:
        torch.save(dets_main.detach().float().cpu(), args.dump_dets)
    if rank == 0:
        res = {
            "metric": cfg["metric"],
            "value": world * B * args.steps / el, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "setup_replays": SETUP_REPLAYS, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "value_300_queries": value_300, "value_gemm_tuned": value_tuned, "value_two_batches_in_flight": value_two_in_flight,
            "world_size_seen": dist.get_world_size() if use_dist else 1,
            "per_rank_images_per_s": [B * args.steps / t for t in per_rank],
            "config": {"workload": f"{cfg['name']} transformer stack from feature pyramids: 6 encoder "
                                   f"layers (MSDA self-attn, S={S}, {L} levels) + two-stage top-k + 6 decoder layers (relation-biased "
                                   "self-attn + MSDA cross-attn + box refinement) + top-300 detections; backbone/neck excluded",
                       "batch_per_gpu": B, "global_batch": B * world, "queries": Nq,
                       "queries_note": "value: 900 two-stage queries (what the reference config runs), 300 detections kept; "
                                       "value_300_queries: 300 two-stage queries (BASELINE.json's wording); "
                                       "value_two_batches_in_flight: the same batches with two of them in flight on two streams "
                                       "(serving throughput, a batch's latency doubles) -- informational, never `value`",
                       "weights": "random init (seeded); MSDA offset / attention projections N(0, .02) / N(0, .05), class heads x3, "
                                  "last box-head layers N(0, .01), tgt_embed rows equal -- see build_network",
                       "levels": L, "fp32_images_per_s": fp32_ips, "bf16_vs_fp32_detections": drift,
                       "launch": launch, "streams": nstreams,
                       "gemm_tuning": tuned,
                       "parallelism": f"image-parallel x{world}"},
            "roofline": {"bound": "hbm", "kernel": (ROOFLINE_KERNEL % (gather_kernel_name(B, S, L, False, cfg["shapes"]), B)) if args.dtype == "bf16" else "msda_fwd_qrun_kernel<float, L=%d> (encoder shape, B=%d)" % (L, B),
                         "achieved": alg / t_kernel / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": alg / t_kernel / HBM_PEAK, **pmc_traffic(args.dtype, B, args.config),
                         "algorithmic_bytes": alg, "kernel_ms": t_kernel * 1e3,
                         "kernel_ms_cold": t_kernel_cold * 1e3, "frac_cold": alg / t_kernel_cold / HBM_PEAK,
                         "kernel_timing": "kernel_ms: %d timed launches right after ~50 ms of the stack's own replays and %d untimed "
                                          "launches, no host synchronisation in between (sustained clocks); kernel_ms_cold: 3 warm-up + "
                                          "20 timed launches after a host synchronisation (round-2 protocol: measures the clock ramp)"
                                          % (KERNEL_REPS, 2 * KERNEL_REPS),
                         "kernel_preheat_launches": 2 * KERNEL_REPS,
                         "query_run_kernel_ms": None if t_kernel_qrun is None else t_kernel_qrun * 1e3,
                         "query_run_kernel_note": "the same operator on the same inputs through msda_fwd_qrun_kernel (algo='direct': every corner row "
                                                  "through the texture path, the kernel of rounds 1-3), same timing protocol",
                         "in_stack": in_stack, "relation": relation,
                         "lds_sourced_alternative": None if sweep_ms is None else {
                             "kernel": "msda_fwd_sweep_kernel (csrc/msda_sweep.hip, opt-in algo='sweep'), same inputs", "kernel_ms": sweep_ms,
                             "frac": alg / (sweep_ms * 1e-3) / HBM_PEAK,
                             "data_path_ceiling": "tools/microbench/lds_gather_mfma_rate: 56 us for the ds_read_b64_tr_b16 + MFMA loop alone "
                                                  "(0.51 of the roofline with nothing else), profiles/r04/"},
                         "fp32": None if t_kernel_fp32 is None else {
                             "kernel": "msda_fwd_qrun_kernel<float> on value [B,S,H,D] (the reference operator's own arithmetic, ms_deform_attn.py:360)",
                             "kernel_ms": t_kernel_fp32 * 1e3, "algorithmic_bytes": msda_algorithmic_bytes(B, S, S, L, 4, 8, 32, 4),
                             "frac": msda_algorithmic_bytes(B, S, S, L, 4, 8, 32, 4) / t_kernel_fp32 / HBM_PEAK},
                         "reference_operator_layout_ms": t_kernel_bshd * 1e3 if t_kernel_bshd else None,
                         "reference_operator_layout_note": "same operator on value [B,S,H,D] (the _C contract): LDS-window MFMA "
                                                           "kernel msda_fwd_win_kernel, algo = auto"},
        }
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline")
            res["cpu_baseline"] = cpu_baseline(Nq, cfg=cfg)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
