"""HIP-graph replay of the eval forward for fixed input shapes (serving path).

The transformer stack launches a few hundred small kernels per image batch; enqueueing them from Python costs about as
much wall time as the GPU needs to run them.  Every operator of this package only enqueues work on the current stream
(include/relation_detr_amd.h: no allocation, no synchronisation, no global state), and the harness keeps all
shape-dependent tables on the device (RelationTransformer.level_geometry), so the whole forward can be captured once
into a hipGraph and replayed with one launch per batch.

    run = GraphedCall(lambda f0, ..., m0, ..., p0, ...: net(...), example_tensors)
    out = run(*tensors)        # tensors with the captured shapes / dtypes; copied into the static inputs if they are
                               # not the captured objects themselves; the returned tensors are overwritten by the next call
"""
from __future__ import annotations

import threading
from typing import Callable, Sequence

import torch

from . import _lib


class ImageGroups:
    """Run ``fn`` on ``groups`` equal slices of the batch dimension, one HIP stream per slice, forked from and joined
    back into the current stream (so the whole thing can sit inside a captured graph).

    Images are independent units of the path, and its kernels are bound by different resources: the
    deformable gather by L2 requests and the VALU, the projections / FFN by the matrix cores, add+LayerNorm by HBM.  Two
    image groups in flight let the hardware overlap them (measured on the R50 stack, B = 4: +8 % images/s with 2 groups;
    4 groups of one image lose GEMM efficiency again).  ``fn(*slices) -> Tensor | tuple[Tensor]`` must only enqueue on
    the current stream; results are concatenated along dim 0 on the calling stream."""

    def __init__(self, fn: Callable, groups: int, device=None):
        if groups < 1:
            raise _lib.RdetrError("ImageGroups: groups must be >= 1")
        self._fn, self.groups = fn, groups
        self._streams = [torch.cuda.Stream(device=device) for _ in range(groups)] if groups > 1 else []

    def __call__(self, *tensors: torch.Tensor):
        if self.groups == 1:
            return self._fn(*tensors)
        # fp32 activations: refused before anything is launched.  The dense projections of an fp32 stack run on the GEMM library,
        # whose fp32 kernels on this image are all stream-K variants (`Cijk_..._SK3_SKXCCM8_...` in the one-stream trace,
        # profiles/r01/bench_fullstack_fp32_kernel_stats.csv): persistent grids whose workgroups wait on partial tiles of their
        # peers.  Two of them side by side on two streams stop making progress -- the device hangs (profiles/r03/
        # fp32_two_group_hang_bisect.txt: `gemm_2streams`).  bf16 stacks use this package's own projection kernels.
        if any(t.dtype == torch.float32 for t in tensors):
            raise _lib.RdetrError("ImageGroups: float32 inputs run as ONE image group (two fp32 groups on parallel streams hang "
                                  "the device: concurrent stream-K library GEMMs, DESIGN.md section 4.8); use groups=1 or bfloat16")
        B = tensors[0].shape[0]
        if B % self.groups or any(t.shape[0] != B for t in tensors):
            raise _lib.RdetrError(f"ImageGroups: every tensor needs the same batch size, divisible by {self.groups}")
        per, cur, outs = B // self.groups, torch.cuda.current_stream(), []
        for i, s in enumerate(self._streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(self._fn(*[t[i * per:(i + 1) * per] for t in tensors]))
        for s in self._streams:
            cur.wait_stream(s)
        for o in outs:                                  # the caching allocator must not recycle them under the side streams
            for t in (o if isinstance(o, (tuple, list)) else (o,)):
                t.record_stream(cur)
        if isinstance(outs[0], (tuple, list)):
            return tuple(torch.cat(parts, 0) for parts in zip(*outs))
        return torch.cat(outs, 0)


class GraphedCall:
    """Capture ``fn(*example_inputs)`` once, replay it with one launch per call.

    What construction assumes, and enforces (a second GraphedCall warmed up while the first one was
    replaying on another stream ended in a GPU memory fault):
      * the DEVICE IS IDLE for this process while a graph is warmed up and captured.  The warm-up runs eagerly on a side
        stream and allocates from the caching allocator's ordinary pool; host-side caches (packed weights, level tables) are
        filled by whichever call comes first and are only fenced against the streams that exist at that moment.  A replay of
        another graph in flight at that time is outside what those fences cover.  So the constructor starts with a device-
        wide synchronize, and construction and replay exclude each other through one process-wide lock: a replay attempted
        (from another thread) while some GraphedCall is being built raises instead of racing it;
      * each instance owns a private memory pool (torch's default for a new CUDAGraph); the tensors it returns live there and
        are overwritten by its next replay -- two instances never share buffers;
      * replays of DIFFERENT instances may overlap on different streams once both are built."""

    _build_lock = threading.Lock()

    def __init__(self, fn: Callable, example_inputs: Sequence[torch.Tensor], warmup: int = 3):
        if not example_inputs or not all(t.is_cuda for t in example_inputs):
            raise _lib.RdetrError("GraphedCall needs device tensors (there is no CPU path)")
        _lib.load()
        self._fn = fn
        self._inputs = list(example_inputs)             # the captured objects: replay reads these buffers
        if torch.cuda.is_current_stream_capturing():
            raise _lib.RdetrError("GraphedCall cannot be built inside another capture")
        if not GraphedCall._build_lock.acquire(blocking=False):
            raise _lib.RdetrError("another GraphedCall is being built on another thread: build graphs one at a time, with the "
                                  "device idle")
        try:
            torch.cuda.synchronize()                    # nothing of this process in flight: earlier graphs' replays have drained
            side = torch.cuda.Stream(device=self._inputs[0].device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():  # warm-up outside capture: host-side caches, lazy handles, autotuning
                for _ in range(max(1, warmup)):
                    fn(*self._inputs)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (the RCCL watchdog of an initialised process group polls events)
            # must not invalidate the capture
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"), torch.no_grad():
                self._outputs = fn(*self._inputs)
            torch.cuda.synchronize()
        finally:
            GraphedCall._build_lock.release()

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self._inputs):
            raise _lib.RdetrError(f"expected {len(self._inputs)} tensors, got {len(inputs)}")
        if GraphedCall._build_lock.locked():
            raise _lib.RdetrError("a GraphedCall is being built on another thread: replays must wait for it (device idle during capture)")
        for dst, src in zip(self._inputs, inputs):
            if src is not dst:
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise _lib.RdetrError("GraphedCall: input shape / dtype differs from the captured one")
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        return self._outputs
