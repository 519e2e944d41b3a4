#!/usr/bin/env python3
"""Max |kernel - reference| of the relation bias against the fp32 and the fp64 evaluation of the reference formula
(oracle/torch_ref.py), typical boxes and tiny boxes (w, h ~ 1e-4): the numbers quoted in DESIGN.md 4.4."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import relation_detr_amd as rd
from oracle import torch_ref
g = torch.Generator().manual_seed(0)
for name, lo, hi in (("typical", 0.01, 0.5), ("tiny", 1e-4, 2e-4), ("mixed", 1e-4, 0.9)):
    src = torch.cat([torch.rand(2, 300, 2, generator=g), torch.rand(2, 300, 2, generator=g) * (hi - lo) + lo], -1)
    tgt = torch.cat([torch.rand(2, 300, 2, generator=g), torch.rand(2, 300, 2, generator=g) * (hi - lo) + lo], -1)
    w = (torch.rand(8, 64, 1, 1, generator=g) - 0.5) * 0.58
    b = (torch.rand(8, generator=g) - 0.5) * 0.25
    out = rd.relation_bias(src.cuda(), tgt.cuda(), w.cuda(), b.cuda()).cpu()
    r32 = torch_ref.relation_bias(src, tgt, w, b)
    r64 = torch_ref.relation_bias(src.double(), tgt.double(), w.double(), b.double()).float()
    print(f"{name:8s} |kernel-ref32| {(out-r32).abs().max():.2e}  |kernel-ref64| {(out-r64).abs().max():.2e}  |ref32-ref64| {(r32-r64).abs().max():.2e}")

# the same through the entry point WITHOUT the tables (every coordinate evaluated per pair in fp32)
from relation_detr_amd import _lib
lib = _lib.load()
g = torch.Generator().manual_seed(0)
for name, lo, hi in (("typical", 0.01, 0.5), ("tiny", 1e-4, 2e-4), ("mixed", 1e-4, 0.9)):
    src = torch.cat([torch.rand(2, 300, 2, generator=g), torch.rand(2, 300, 2, generator=g) * (hi - lo) + lo], -1)
    tgt = torch.cat([torch.rand(2, 300, 2, generator=g), torch.rand(2, 300, 2, generator=g) * (hi - lo) + lo], -1)
    w = (torch.rand(8, 64, 1, 1, generator=g) - 0.5) * 0.58
    b = (torch.rand(8, generator=g) - 0.5) * 0.25
    s_, t_, w_, b_ = src.cuda(), tgt.cuda(), w.reshape(8, 64).contiguous().cuda(), b.cuda()
    out = torch.empty(2, 8, 300, 300, device="cuda")
    assert lib.rdetr_relation_bias_f32(s_.data_ptr(), t_.data_ptr(), w_.data_ptr(), b_.data_ptr(), 2, 300, 300, 8, 16, 100.0, 10000.0,
                                       1e-5, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    out = out.cpu()
    r32 = torch_ref.relation_bias(src, tgt, w, b)
    r64 = torch_ref.relation_bias(src.double(), tgt.double(), w.double(), b.double()).float()
    print(f"no-table {name:8s} |kernel-ref32| {(out-r32).abs().max():.2e}  |kernel-ref64| {(out-r64).abs().max():.2e}")
