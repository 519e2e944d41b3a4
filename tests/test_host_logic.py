"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the nn.Modules keep the reference's constructor / attribute / state_dict contract, the
pre-kernel torch logic (sampling locations) matches the oracle, and the product path refuses to run
without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import relation_detr_amd as rd
from relation_detr_amd import _lib, ops
from oracle import torch_ref

T = torch.from_numpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "relation_detr_amd.h")).read()
    declared = set(re.findall(r"\b(rdetr_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().rdetr_abi_version() == 3
    assert _lib.load().rdetr_status_string(-2).decode().startswith("shape not supported")
    assert _lib.load().rdetr_msda_fast_path(8, 32, 4, 4) == 1
    assert _lib.load().rdetr_msda_fast_path(8, 32, 5, 4) == 1
    assert _lib.load().rdetr_msda_fast_path(8, 64, 4, 4) == 0


def test_c_abi_argument_validation_without_gpu():
    """Argument checks run before any HIP call, so they are testable on a CPU-only host."""
    lib = _lib.load()
    assert lib.rdetr_msda_forward_f32(None, None, None, None, None, 1, 10, 8, 32, 4, 5, 4, None, None) == -1
    assert lib.rdetr_msda_forward_f32(None, None, None, None, None, 0, 10, 8, 32, 4, 5, 4, None, None) == 0   # empty batch
    assert lib.rdetr_msda_forward_f32(None, None, None, None, None, 1, 10, 8, 32, 4, -1, 4, None, None) == -1
    assert lib.rdetr_relation_bias_f32(None, None, None, None, 1, 4, 4, 8, 15, 100.0, 10000.0, 1e-5, None, None) == -2  # odd F
    assert lib.rdetr_relation_bias_f32(None, None, None, None, 1, 4, 4, 8, 16, 100.0, 10000.0, 1e-5, None, None) == -1
    assert lib.rdetr_bias_softmax_f32(None, None, None, 0, 4, 4, None) == 0
    assert lib.rdetr_bias_softmax_f32(None, None, None, 2, 4, 4, None) == -1
    assert lib.rdetr_msda_backward_f32(*([None] * 6), 1, 10, 8, 32, 4, 5, 4, None, None, None, None) == -1
    # the sweep kernel's entry point takes the level table as HOST pointers and refuses before any HIP call
    shapes = (ctypes.c_int64 * 8)(64, 96, 32, 48, 16, 24, 8, 12)
    starts = (ctypes.c_int64 * 4)(0, 6144, 7680, 8064)
    one = ctypes.c_void_p(16)                                       # an aligned non-null dummy: never dereferenced on these paths
    assert lib.rdetr_msda_forward_sweep_bf16(None, 0, shapes, starts, None, None, 1, 8160, 8, 32, 4, 8160, 4, None, None) == -1
    assert lib.rdetr_msda_forward_sweep_bf16(None, 0, shapes, starts, None, None, 0, 8160, 8, 32, 4, 8160, 4, None, None) == 0
    assert lib.rdetr_msda_forward_sweep_bf16(one, 7, shapes, starts, one, one, 1, 8160, 8, 32, 4, 8160, 4, one, None) == -1    # layout
    assert lib.rdetr_msda_forward_sweep_bf16(one, 0, shapes, starts, one, one, 1, 8160, 8, 64, 4, 8160, 4, one, None) == -2    # head dim
    assert lib.rdetr_msda_forward_sweep_bf16(one, 0, shapes, starts, one, one, 1, 8160, 8, 32, 4, 900, 4, one, None) == -2     # Nq != S
    assert lib.rdetr_msda_forward_sweep_bf16(one, 0, shapes, starts, one, one, 1, 9000, 8, 32, 4, 9000, 4, one, None) == -2    # levels do not tile S
    # ... and so do the resident-levels kernel's (csrc/msda_res.hip): shape refusals are RDETR_ERR_UNSUPPORTED (callers fall back)
    res, resf = lib.rdetr_msda_forward_resident_bf16, lib.rdetr_msda_forward_fused_resident_bf16
    assert res(None, shapes, starts, None, None, 1, 8160, 8, 32, 4, 8160, 4, None, None) == -1
    assert res(None, shapes, starts, None, None, 0, 8160, 8, 32, 4, 8160, 4, None, None) == 0
    assert res(one, shapes, starts, one, one, 1, 8160, 8, 64, 4, 8160, 4, one, None) == -2                  # head dim
    assert res(one, shapes, starts, one, one, 1, 9000, 8, 32, 4, 9000, 4, one, None) == -2                  # levels do not tile S
    assert res(one, shapes, starts, one, one, 1, 8064, 8, 32, 3, 8064, 4, one, None) == -2                  # three levels
    assert res(one, shapes, starts, ctypes.c_void_p(8), one, 1, 8160, 8, 32, 4, 8160, 4, one, None) == -2   # locations not 16-byte aligned
    big = (ctypes.c_int64 * 8)(64, 64, 56, 56, 52, 52, 48, 48)
    bigs = (ctypes.c_int64 * 4)(0, 4096, 7232, 9936)
    assert res(one, big, bigs, one, one, 1, 12240, 8, 32, 4, 12240, 4, one, None) == -2                     # coarsest level (147 KB) does not fit the LDS
    assert resf(one, shapes, starts, one, 0, one, 0, None, 2, 1, 8160, 8, 32, 4, 8160, 4, one, None) == -1  # no reference points
    assert resf(one, shapes, starts, one, 0, one, 0, one, 3, 1, 8160, 8, 32, 4, 8160, 4, one, None) == -1   # ref_dim
    assert resf(one, shapes, starts, one, 0, one, 0, one, 4, 1, 8160, 8, 32, 4, 8160, 4, one, None) == -2   # 4-d reference points: the query-run kernel's
    assert resf(one, shapes, starts, one, 100, one, 0, one, 2, 1, 8160, 8, 32, 4, 8160, 4, one, None) == -1  # row stride below a row


def test_msda_module_contract(golden):
    g = golden("g4_msda_module.npz")
    mod = rd.MultiScaleDeformableAttention(256, 4, 8, 4)              # positional, as relation_transformer.py:223,401
    assert (mod.im2col_step, mod.embed_dim, mod.num_heads, mod.num_levels, mod.num_points) == (64, 256, 8, 4, 4)
    ref_sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    assert {k: tuple(v.shape) for k, v in mod.state_dict().items()} == {k: v.shape for k, v in ref_sd.items()}
    # fresh-module initialisation equals the reference's: the offset bias ring pattern was NOT randomised
    # in the fixture, zero offset weights / uniform attention are the reference's documented init
    np.testing.assert_allclose(mod.sampling_offsets.bias.detach().numpy(), ref_sd["sampling_offsets.bias"], atol=1e-6)
    assert mod.sampling_offsets.weight.abs().max() == 0 and mod.attention_weights.weight.abs().max() == 0
    assert mod.attention_weights.bias.abs().max() == 0 and mod.value_proj.bias.abs().max() == 0
    mod.load_state_dict({k: T(v) for k, v in ref_sd.items()})          # released checkpoints load
    with pytest.raises(ValueError):
        rd.MultiScaleDeformableAttention(250, 4, 8, 4)


def test_msda_module_pre_kernel_logic_matches_oracle(golden):
    g = golden("g4_msda_module.npz")
    mod = rd.MultiScaleDeformableAttention(256, 4, 8, 4)
    mod.load_state_dict({k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")})
    shapes = T(g["shapes"])
    with torch.no_grad():
        for q, ref, mask in ((T(g["q_enc"]), T(g["ref2"]), T(g["mask"])), (T(g["q_dec"]), T(g["ref4"]), None)):
            v, loc, w = mod.project_inputs(q, ref, T(g["feat"]), shapes, mask)
            off = mod.sampling_offsets(q).view(*q.shape[:2], 8, 4, 4, 2)
            np.testing.assert_allclose(loc.numpy(), torch_ref.sampling_locations_from_reference(ref, off, shapes, 4).numpy(), atol=1e-6)
            assert v.shape == (2, 321, 8, 32) and w.shape == (*q.shape[:2], 8, 4, 4)
            np.testing.assert_allclose(w.sum((-1, -2)).numpy(), 1.0, atol=1e-5)
            if mask is not None:
                assert v[mask].abs().max() == 0
            # the core of the product path is HIP-only: on CPU tensors it refuses instead of falling back
            with pytest.raises(_lib.RdetrError):
                mod(query=q, reference_points=ref, value=T(g["feat"]), spatial_shapes=shapes,
                    level_start_index=T(g["level_start"]), key_padding_mask=mask)
    with pytest.raises(ValueError):
        rd.ms_deform_attn.sampling_locations(torch.zeros(1, 2, 4, 3), torch.zeros(1, 2, 8, 4, 4, 2), shapes, 4)


def test_relation_module_contract(golden):
    g = golden("g5_relation.npz")
    rel = rd.PositionRelationEmbedding(16, 8)
    assert rd.PositionRelationEncoder is rd.PositionRelationEmbedding
    assert {k: tuple(v.shape) for k, v in rel.state_dict().items()} == {"pos_proj.0.weight": (8, 64, 1, 1), "pos_proj.0.bias": (8,)}
    rel.load_state_dict({"pos_proj.0.weight": T(g["proj_weight"]), "pos_proj.0.bias": T(g["proj_bias"])})
    np.testing.assert_allclose(rd.box_rel_encoding(T(g["src"]), T(g["tgt"])).numpy(), g["enc"], atol=1e-6)
    with pytest.raises(_lib.RdetrError):
        rel(T(g["src"]), T(g["tgt"]))                                  # CPU tensors: no fallback
    with pytest.raises(Exception):
        rel(torch.zeros(1, 3, 5))
    with pytest.raises(NotImplementedError):
        rd.PositionRelationEmbedding(16, 8, activation_layer=torch.nn.GELU)


def test_self_attention_contract(golden):
    g = golden("g6_self_attn.npz")
    att = rd.RelationSelfAttention(256, 8, dropout=0.0, batch_first=True)
    ref = torch.nn.MultiheadAttention(256, 8, dropout=0.0, batch_first=True)
    assert {k: tuple(v.shape) for k, v in att.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    att.load_state_dict(ref.state_dict())
    with pytest.raises(_lib.RdetrError):
        att(query=T(g["qp"]), key=T(g["qp"]), value=T(g["vv"]), attn_mask=None, need_weights=False)


def test_ops_reject_cpu_tensors_and_bad_levels():
    v = torch.zeros(1, 16, 8, 32)
    shp = torch.tensor([[4, 4]])
    st = torch.tensor([0])
    loc, aw = torch.zeros(1, 2, 8, 1, 4, 2), torch.zeros(1, 2, 8, 1, 4)
    with pytest.raises(_lib.RdetrError):
        ops.ms_deform_attn_forward(v, shp, st, loc, aw, 64)
    with pytest.raises(_lib.RdetrError):
        ops.bias_softmax_(torch.zeros(2, 3, 4))
    with pytest.raises(_lib.RdetrError):
        ops.relation_bias(torch.zeros(1, 2, 4), torch.zeros(1, 2, 4), torch.zeros(8, 64), None)
    with pytest.raises(_lib.RdetrError):
        ops.check_levels(torch.tensor([[5, 4]]), st, 16)               # 20 positions do not fit S = 16
    with pytest.raises(_lib.RdetrError):
        ops.check_levels(shp.int(), st, 16)                            # wrong dtype
    ops.check_levels(shp, st, 16)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librelation_detr_amd.so")
    with pytest.raises(_lib.RdetrError, match="no CPU or PyTorch fallback"):
        _lib.load()


def test_packed_weight_cache_is_tied_to_tensor_objects():
    """ops._PackedWeightCache (ffn_k256 / linear_ln_k256 packed weights): a hit needs the SAME tensor objects at the same
    version; a new tensor that happens to reuse a dead tensor's address, id and version must miss."""
    import gc

    import torch

    from relation_detr_amd.ops import _PackedWeightCache
    cache, builds = _PackedWeightCache(prune_every=4), []

    def build_for(t):
        def build():
            builds.append(float(t.sum()))
            return t.clone() * 2
        return build

    a = torch.ones(4, 4)
    p1 = cache.get((a,), build_for(a))
    assert cache.get((a,), build_for(a)) is p1 and len(builds) == 1            # hit
    a.add_(1)                                                                  # in-place update: version bump
    p2 = cache.get((a,), build_for(a))
    assert p2 is not p1 and len(builds) == 2 and torch.equal(p2, a * 2)
    # a dead tensor's id can be recycled by the allocator: simulate the worst case by planting a stale entry under the new id
    b = torch.full((4, 4), 5.0)
    stale = cache._entries[(id(a),)]
    cache._entries[(id(b),)] = stale                                           # same "version", wrong object behind the weakref
    p3 = cache.get((b,), build_for(b))
    assert torch.equal(p3, b * 2) and len(builds) == 3
    # lifetime: the packed copy of a LIVE tensor is never evicted, however many other weights pass through --
    # a captured HIP graph may read it on replay and the cache holds the only reference
    keep = [torch.full((2, 2), float(i)) for i in range(40)]
    packed_b = cache.get((b,), build_for(b))
    for t in keep:
        assert torch.equal(cache.get((t,), build_for(t)), t * 2)
    assert cache.get((b,), build_for(b)) is packed_b and len(cache) >= 41
    for t in keep:
        assert cache.get((t,), build_for(t)) is cache.get((t,), build_for(t))
    # ... and entries of dead tensors are pruned
    n_live = len(cache)
    del keep, t
    gc.collect()
    for i in range(8):
        u = torch.full((3, 3), float(i))
        cache.get((u,), build_for(u))
        del u
    gc.collect()
    assert len(cache) < n_live


def test_levels_window_ok_host_helper():
    """rdetr_msda_levels_window_ok: the precondition of RDETR_MSDA_WINDOW / _AUTO_PACKED, pure host arithmetic."""
    import ctypes
    from relation_detr_amd import _lib
    lib = _lib.load()

    def ok(shapes, starts, S):
        hs = (ctypes.c_int64 * (2 * len(shapes)))(*[v for hw in shapes for v in hw])
        st = (ctypes.c_int64 * len(starts))(*starts)
        return bool(lib.rdetr_msda_levels_window_ok(hs, st, len(shapes), S))

    r50 = [(100, 168), (50, 84), (25, 42), (13, 21)]
    starts = [0, 16800, 21000, 22050]
    assert ok(r50, starts, 22323)
    assert not ok(r50, starts, 22324)                               # value longer than the levels
    assert not ok(r50, [0, 16900, 21100, 22150], 22423)             # gap after level 0
    assert not ok(r50[::-1], [0, 273, 1323, 5523], 22323)           # cumulative but growing: level 1 outgrows level 0
    assert not ok(r50, [22050 - 16800, 0, 1, 2], 22323)             # not cumulative
    assert not ok([(0, 5)], [0], 0)


def test_options_object():
    """One immutable options object: read from the environment once, copied by modules at construction, replaceable per
    module tree -- nothing on a forward path reads the environment (VERDICT r02 hygiene item 13)."""
    import dataclasses
    import inspect
    from relation_detr_amd import ms_deform_attn, options, transformer
    o = options.Options.from_env({"RDETR_FFN_FUSED": "0", "RDETR_LINEAR_K256": "1", "RDETR_MASK_IN_KERNEL": "never", "OTHER": "x"})
    assert (o.ffn_fused, o.linear_k256, o.mask_in_kernel, o.rel_fused) == (False, True, "never", True)
    with pytest.raises(ValueError):
        options.Options.from_env({"RDETR_TOPK": "yes"})
    with pytest.raises(dataclasses.FrozenInstanceError):
        o.topk = False
    with options.override(rel_fused=False):
        net = transformer.build_relation_transformer(num_classes=3, d_ffn=16, enc_layers=1, dec_layers=1, num_queries=4)
    assert net.decoder.options.rel_fused is False and options.get().rel_fused is True
    options.apply(net, rel_fused=True, box_head=False)
    assert all(m.options.rel_fused and not m.options.box_head for m in net.modules() if hasattr(m, "options"))
    for mod in (transformer, ms_deform_attn):
        assert "os.environ" not in inspect.getsource(mod)


def test_contiguous_copies_of_small_operands_stay_referenced():
    """ops._cptr: a contiguous tensor is passed through; a copy made for a strided one is kept referenced past the expression
    that asked for its pointer (the launch reading it is enqueued after ALL arguments are evaluated)."""
    import gc
    from relation_detr_amd import ops
    t = torch.arange(8, dtype=torch.float32)
    assert ops._cptr(t) == t.data_ptr()
    before = len(ops._KEEP)
    p = ops._cptr(t[::2])
    gc.collect()
    assert p != t.data_ptr() and len(ops._KEEP) == min(before + 1, ops._KEEP.maxlen)
    kept = ops._KEEP[-1]
    assert kept.data_ptr() == p and kept.is_contiguous() and torch.equal(kept, t[::2])


def test_resident_kernel_policy():
    """'auto' takes the resident-levels kernel only where it was measured faster: 4 levels, levels 2 + 3 resident, many queries."""
    r50 = ((100, 168), (50, 84), (25, 42), (13, 21))
    assert ops._resident_pays(4, 22323, 4, r50) and ops._resident_pays(1, 22323, 4, r50)
    assert not ops._resident_pays(4, 900, 4, r50)                                      # decoder cross-attention: few queries
    assert not ops._resident_pays(2, 204098, 5, ((300, 500), (150, 250), (75, 125), (38, 63), (19, 32)))
    assert not ops._resident_pays(2, 46000, 4, ((150, 250), (75, 125), (38, 63), (19, 32)))   # levels 2 + 3 = 192 KB: only one would fit
