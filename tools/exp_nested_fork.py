#!/usr/bin/env python3
"""Does HIP graph capture survive a NESTED stream fork (origin -> group stream -> side stream -> join)?  One variant per run:
    python3 tools/exp_nested_fork.py <variant> [capture_error_mode]      variants: flat | nested_events | nested_waitstream | nested_gemm | nested_multi | sibling"""
import sys

import torch


def main():
    variant = sys.argv[1]
    dev = torch.device("cuda", 0)
    x = torch.randn(4096, 256, device=dev, dtype=torch.bfloat16)
    w = torch.randn(256, 256, device=dev, dtype=torch.bfloat16)
    s0, side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def body():
        cur = torch.cuda.current_stream()
        s0.wait_stream(cur)
        if variant == "sibling":
            side.wait_stream(cur)                 # forked from the ORIGIN, like s0; the s0 -> side -> s0 edges are events
        with torch.cuda.stream(s0):
            y = x * 2
            if variant == "flat":
                z = y @ w
            elif variant == "nested_waitstream":
                side.wait_stream(s0)
                with torch.cuda.stream(side):
                    z = y + 1
                s0.wait_stream(side)
            elif variant == "nested_events":
                side.wait_stream(s0)
                with torch.cuda.stream(side):
                    z = y + 1
                    ev = torch.cuda.Event()
                    ev.record(side)
                s0.wait_event(ev)
            elif variant == "nested_gemm":
                side.wait_stream(s0)
                with torch.cuda.stream(side):
                    z = y @ w
                s0.wait_stream(side)
            elif variant == "nested_multi":
                side.wait_stream(s0)
                evs, zs = [], []
                with torch.cuda.stream(side):
                    for _ in range(3):
                        zs.append(y @ w)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        evs.append(ev)
                z = y
                for ev, t in zip(evs, zs):
                    s0.wait_event(ev)
                    z = z + t
            elif variant == "sibling":
                ev1 = torch.cuda.Event()
                ev1.record(s0)
                side.wait_event(ev1)
                evs, zs = [], []
                with torch.cuda.stream(side):
                    for _ in range(3):
                        zs.append(y @ w)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        evs.append(ev)
                z = y
                for ev, t in zip(evs, zs):
                    s0.wait_event(ev)
                    z = z + t
            out = z.float().sum()
        cur.wait_stream(s0)
        if variant == "sibling":
            cur.wait_stream(side)
        return out

    warm = torch.cuda.Stream(device=dev)
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    mode = sys.argv[2] if len(sys.argv) > 2 else "thread_local"
    with torch.cuda.graph(g, capture_error_mode=mode):
        out = body()
    g.replay()
    torch.cuda.synchronize()
    print(variant, "ok", float(out))


if __name__ == "__main__":
    main()
