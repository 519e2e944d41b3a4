"""The BENCHED workload, validated end to end ("the benched workload is never validated"):
relation_detr_resnet50_800_1333 -- R50 pyramid (100,168)(50,84)(25,42)(13,21), 6 + 6 layers, d_ffn 2048, 900 / 300
two-stage queries -- through the same objects bench.py times.

  (i)   fp32 HIP harness vs the CPU oracle harness (oracle/cpu_modules.py: the reference's operators op for op) on one
        full-size image: identical two-stage proposal choice, logits / boxes <= 5e-4.
  (ii)  bf16, B = 4, two image groups, HIP-graph replay (bench.py's default launch: fused FFN, linear+LayerNorm, key-split
        attention, window / direct MSDA all in composition) vs the fp32 HIP harness on the same images: encoder memory
        within 2^-5 relative (L2), detections reproduced to the stated bound.
  (iii) relation_bias at B = 4, N1 = N2 = 900 (BASELINE.json configs[3]: "stresses the N_q^2 relation kernel") vs
        oracle.torch_ref.relation_bias evaluated in fp32 and in fp64: <= 1e-4.
  (iv)  fused bf16 decoder self-attention (kernel and module) on the PINNED fixture g6 (the reference's own
        nn.MultiheadAttention output): 2^-7 |ref| + 4e-3.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


@pytest.fixture(scope="module")
def bench():
    # GEMM tuning (PyTorch's TunableOp) is opt-in in bench.py and off here: these tests compare numerics, and the library kernel a
    # tuner picks may differ from run to run.  test_fp32_self_attention_under_gemm_tuning below switches it on on purpose.
    import bench as b
    return b


def _forward(net, feats, masks, pos):
    with torch.no_grad():
        return net(feats, masks, pos)


@pytest.mark.parametrize("config,queries", [("r50", 900), ("r50", 300), ("focalnet", 900)])
def test_fp32_stack_matches_cpu_oracle_full_size(bench, config, queries):
    """BASELINE.json configs[1] (R50, 4 levels, S = 22,323) and configs[4] (FocalNet-L, 5 levels (304,504) ... (19,32),
    S = 204,098: pyramid_points, nchw_to_tokens, the memory-fusion buffer, the L = 5 gather and the 900-query decoder with
    5-level cross-attention) on one full-size padded image.
    Encoder + two-stage scores end to end, then the decoder on IDENTICAL inputs.  The two-stage top-k over 22,323 nearly
    equal scores (random-init class head: prior bias + O(1e-2)) is decided at the 1e-6 level, where fp32 summation order
    already differs between ATen's CPU kernels and the GPU's: a single swapped proposal reorders the queries.  So the
    continuous quantities are compared where they are continuous -- encoder memory and the full [S, C] score map -- the
    discrete choice is required to agree on >= 99 % of the proposals, and the decoder (6 layers, 900 / 300 queries, relation
    bias, box refinement) is run on both sides from the GPU's own proposals."""
    from oracle.cpu_modules import OracleMSDA, OracleRelation, OracleSelfAttention
    torch.set_num_threads(16)
    cfg = bench.CONFIGS[config]
    nlev = len(cfg["shapes"])
    cpu_net = bench.build_network(queries, 0, num_levels=nlev, msda_cls=OracleMSDA, self_attn_cls=OracleSelfAttention,
                                  relation_cls=OracleRelation)
    gpu_net = bench.build_network(queries, 0, num_levels=nlev).to(DEV)
    assert all(torch.equal(a, b.cpu()) for a, b in zip(cpu_net.state_dict().values(), gpu_net.state_dict().values()))
    feats, masks, pos = bench.build_pyramid(1, "cpu", 7, shapes=cfg["shapes"])
    for l in range(nlev):                                    # right padding: valid_ratios < 1, padded value rows
        w = masks[l].shape[2]
        masks[l][0, :, max(1, w * 150 // 168):] = True
    seen = {}
    for tag, net in (("cpu", cpu_net), ("gpu", gpu_net)):
        net.encoder.register_forward_hook(lambda m, i, o, tag=tag: seen.__setitem__(tag + "_memory", o.detach().float().cpu()))
        net.encoder_class_head.register_forward_hook(lambda m, i, o, tag=tag: seen.__setitem__(tag + "_scores", o.detach().float().cpu()))
    want = _forward(cpu_net, feats, masks, pos)
    dfeats, dmasks, dpos = [f.to(DEV) for f in feats], [m.to(DEV) for m in masks], [p.to(DEV) for p in pos]
    got = _forward(gpu_net, dfeats, dmasks, dpos)
    # (a) the continuous part of the encoder side, full size: 6 MSDA layers, FFNs, memory fusion, encoder heads
    np.testing.assert_allclose(seen["gpu_memory"].numpy(), seen["cpu_memory"].numpy(), rtol=0, atol=5e-4)
    np.testing.assert_allclose(seen["gpu_scores"].numpy(), seen["cpu_scores"].numpy(), rtol=0, atol=5e-4)
    # (b) the discrete two-stage choice: the same proposals up to ties at fp32 resolution
    gc, cc = got[3].float().cpu()[0], want[3][0]
    same = (torch.cdist(gc.double(), cc.double(), p=float("inf")) < 1e-5).any(1).float().mean().item()
    assert same >= 0.99, f"only {same:.3f} of the two-stage proposals agree"
    # (c) the decoder on identical inputs: the GPU run's memory and proposals, handed to both
    geo, vr = gpu_net.level_misc(dmasks)
    mask = gpu_net.flatten_levels(dmasks)
    memory, ref = seen["gpu_memory"], got[3].float().detach()
    with torch.no_grad():
        g_cls, g_box = gpu_net.decoder(query=gpu_net.tgt_embed.weight.expand(1, -1, -1), value=memory.to(DEV), key_padding_mask=mask,
                                       reference_points=ref, spatial_shapes=geo["shapes"], level_start_index=geo["start"],
                                       valid_ratios=vr)
        c_cls, c_box = cpu_net.decoder(query=cpu_net.tgt_embed.weight.expand(1, -1, -1), value=memory, key_padding_mask=mask.cpu(),
                                       reference_points=ref.cpu(), spatial_shapes=geo["shapes"].cpu(),
                                       level_start_index=geo["start"].cpu(), valid_ratios=vr.cpu())
    assert g_cls.shape == (6, 1, queries, 91) and geo["shapes"].shape[0] == nlev
    np.testing.assert_allclose(g_cls.float().cpu().numpy(), c_cls.numpy(), rtol=0, atol=5e-4)
    np.testing.assert_allclose(g_box.float().cpu().numpy(), c_box.numpy(), rtol=0, atol=5e-4)
    # and the harness's own decoder outputs are the ones just checked (same proposals in, same kernels)
    # (the full forward takes its valid ratios / reference points from rdetr_pyramid_points, the re-run from the torch
    # statements: equal up to the last bit, which six layers turn into <= 1e-3 on logits of +-7 at 204k tokens)
    # -- the loose bound is the FocalNet case's only; r50: 1e-4 (measured 6.5e-5 at 900 queries: the same
    # last-bit difference of the reference points through six layers at 22k tokens; 1e-5 does not hold there either)
    if config == "focalnet":
        np.testing.assert_allclose(got[0].float().cpu().numpy(), g_cls.float().cpu().numpy(), rtol=1e-4, atol=1e-3)
    else:
        np.testing.assert_allclose(got[0].float().cpu().numpy(), g_cls.float().cpu().numpy(), rtol=0, atol=1e-4)


def test_bf16_two_group_replay_vs_fp32_full_size(bench):
    """bench.py's default launch against the fp32 harness, on bench.py's network (build_network: class heads x3 so the scores
    are spread far above bf16 resolution, box heads that move the boxes, exchangeable content queries).
      * encoder memory bf16 vs fp32: relative L2 < 2^-5;
      * replay vs eager: every detection reproduced (>= 0.99 at IoU 0.9; measured bit-identical: both selections are
        rdetr_topk's total order, nothing on the path is order-dependent any more);
      * detections bf16 vs fp32 at IoU 0.9, same label: >= 0.95 (measured 0.97, tools/exp_separation.py).
    Why `exchangeable_queries`: the two-stage cut ranks 22,323 continuous scores and hands query slot r the r-th best proposal.
    bf16 noise of ~1 % of the score spread is ~20 rank spacings, so the two routes agree on 99 % of the proposal SET but on 5 %
    of the SLOTS (measured), and with N(0,1) tgt_embed rows every slot is a different random function of its proposal: 0.44-0.48
    matched whatever the class scale -- a property of a random-init decoder, not of a kernel.  With equal rows the decoder is
    permutation-equivariant and the match measures arithmetic.  The decoder is bounded on IDENTICAL proposals by
    test_bf16_decoder_on_identical_proposals_full_size."""
    from relation_detr_amd.graph import GraphedCall, ImageGroups
    from relation_detr_amd.transformer import select_detections
    B, L = 4, 4
    feats, masks, pos = bench.build_pyramid(B, DEV, seed=1000, dtype=torch.float32)
    sizes = torch.tensor([[800, 1333]] * B, device=DEV)
    net32 = bench.build_network(900, 0).to(DEV)
    net16 = bench.build_network(900, 0).to(DEV).to(torch.bfloat16)
    mem = {"fp32": [], "bf16": []}                       # per image group, in launch order
    net32.encoder.register_forward_hook(lambda m, i, o: mem["fp32"].append(o.detach().float().clone()))
    net16.encoder.register_forward_hook(lambda m, i, o: mem["bf16"].append(o.detach().float().clone()))

    def make(net):
        @torch.no_grad()
        def fwd(*t):
            classes, coords = net(list(t[:L]), list(t[L:2 * L]), list(t[2 * L:3 * L]))[:2]
            return select_detections(classes[-1].float(), coords[-1].float(), t[3 * L])
        return fwd

    in32 = [*feats, *masks, *pos, sizes]
    in16 = [t.to(torch.bfloat16) if t.is_floating_point() else t for t in in32]
    det32 = make(net32)(*in32).clone()
    eager16 = ImageGroups(make(net16), 2, device=DEV)(*in16).clone()           # hooks fire here (not under capture)
    torch.cuda.synchronize()
    m32, m16 = torch.cat(mem["fp32"], 0), torch.cat(mem["bf16"], 0)
    assert m32.shape == m16.shape and m32.shape[0] == B
    rel = ((m16 - m32).norm() / m32.norm()).item()
    assert rel < 2.0 ** -5, f"encoder memory bf16 vs fp32: relative L2 {rel:.4f}"
    mem["bf16"].clear()                                  # the hook now fires under capture: its clones live in the graph's pool
    run = GraphedCall(ImageGroups(make(net16), 2, device=DEV), in16)
    det16 = run(*in16).clone()
    torch.cuda.synchronize()
    # replay vs eager: the same kernels on the same inputs.  The library GEMMs the heuristics pick under capture are not always
    # the ones picked eagerly (a few bf16 ulps in the encoder memory after six layers); the two selections are rdetr_topk's
    # total order (value, then index), so equal scores no longer reorder from call to call.
    m16_replay = torch.cat(mem["bf16"][-2:], 0)
    rel_replay = ((m16_replay - m16).norm() / m16.norm()).item()
    assert rel_replay < 2.0 ** -7, f"encoder memory replay vs eager: relative L2 {rel_replay:.5f}"
    same = bench.detection_drift(det16, eager16, iou_thr=0.9)
    print("replay vs eager:", same)
    assert same["matched_frac"] >= 0.99, same
    d9 = bench.detection_drift(det16, det32, iou_thr=0.9)
    d5 = bench.detection_drift(det16, det32, iou_thr=0.5)
    print("bf16 vs fp32 detections:", d9, d5, "encoder memory rel L2:", rel)
    assert torch.isfinite(det16).all() and det16.shape == (B, 300, 6)
    assert d9["matched_frac"] >= BF16_MATCH_IOU90, d9
    assert d5["matched_frac"] >= d9["matched_frac"]


BF16_MATCH_IOU90 = 0.95


def test_bf16_decoder_on_identical_proposals_full_size(bench):
    """The continuous bound the detection statistic cannot give: the bf16 decoder -- generated-bias
    attention (csrc/attn_rel.hip), fused box head (csrc/mlp.hip), bf16 fused-producer MSDA cross-attention, decoder_reference,
    scaled_pos, six layers in composition at 900 queries -- against the fp32 decoder on IDENTICAL inputs: the fp32 run's encoder
    memory and its top-900 proposals, B = 2 with one padded image.  Mirrors the fp32-vs-CPU-oracle check above."""
    B, L = 2, 4
    feats, masks, pos = bench.build_pyramid(B, DEV, seed=77, dtype=torch.float32)
    for l in range(L):
        w = masks[l].shape[2]
        masks[l][1, :, w * 150 // 168:] = True
    # DISTINCT content queries (the reference's N(0, 1) tgt_embed rows): with identical proposals handed to both decoders the
    # slot-permutation argument for bench.py's exchangeable rows does not apply, and per-slot query handling (query_pos_k256,
    # the stride-0 expand of layer 0) is only exercised by rows that differ
    net32 = bench.build_network(900, 0, exchangeable_queries=False).to(DEV)
    net16 = bench.build_network(900, 0, exchangeable_queries=False).to(DEV).to(torch.bfloat16)
    assert (net32.tgt_embed.weight[0] - net32.tgt_embed.weight[1]).abs().max().item() > 0.1
    seen = {}
    net32.encoder.register_forward_hook(lambda m, i, o: seen.__setitem__("memory", o.detach().clone()))
    with torch.no_grad():
        got = net32(feats, masks, pos)
        geo, vr = net32.level_misc(masks)
        mask = net32.flatten_levels(masks)
        memory, ref = seen["memory"], got[3].float().detach()
        kw = dict(key_padding_mask=mask, reference_points=ref, spatial_shapes=geo["shapes"], level_start_index=geo["start"],
                  valid_ratios=vr)
        c32, b32 = net32.decoder(query=net32.tgt_embed.weight.expand(B, -1, -1), value=memory, **kw)
        c16, b16 = net16.decoder(query=net16.tgt_embed.weight.expand(B, -1, -1), value=memory.to(torch.bfloat16), **kw)
    np.testing.assert_allclose(got[0].float().cpu().numpy(), c32.cpu().numpy(), rtol=1e-4, atol=5e-4)   # same proposals in, same kernels
    c32, b32, c16, b16 = c32.float(), b32.float(), c16.float(), b16.float()
    assert torch.isfinite(c16).all() and torch.isfinite(b16).all()
    spread = (c32 - c32.mean()).abs().max().item()               # logits: O(1-5) around the prior (class heads x3)
    dbox, dcls = (b16 - b32).abs(), (c16 - c32).abs()
    print(f"bf16 decoder vs fp32 on identical proposals: boxes max {dbox.max().item():.4f} mean {dbox.mean().item():.5f}; "
          f"logits max {dcls.max().item():.4f} mean {dcls.mean().item():.5f} (spread {spread:.2f}); per layer box max "
          f"{[round(v, 4) for v in dbox.amax((1, 2, 3)).tolist()]} logit max {[round(v, 4) for v in dcls.amax((1, 2, 3)).tolist()]}")
    # measured: boxes max 1.7e-3 / mean 8e-5, logits max 0.146 / mean 0.011 at a spread of 6.9 (profiles/r03/gpu_tests_tail.txt)
    assert dbox.max().item() <= 5e-3 and dbox.mean().item() <= 3e-4
    assert dcls.max().item() <= 2.0 ** -5 * spread and dcls.mean().item() <= 2.0 ** -8 * spread


def test_bf16_focalnet_5_level_stack_full_size(bench):
    """BASELINE.json configs[4] per rank in bf16 (the route bench.py --config focalnet times): 5 levels, S = 204,098, 900
    queries, B = 1 -- finite outputs of the right shape, and the encoder memory within 2^-5 relative L2 of the fp32 HIP run
    on the same image (whose own parity against the CPU oracle is the focalnet case of the fp32 test above)."""
    cfg = bench.CONFIGS["focalnet"]
    L = len(cfg["shapes"])
    feats, masks, pos = bench.build_pyramid(1, DEV, seed=5, dtype=torch.float32, shapes=cfg["shapes"])
    for l in range(L):
        h = masks[l].shape[1]
        masks[l][0, max(1, h * 9 // 10):, :] = True            # bottom padding
    net32 = bench.build_network(900, 0, num_levels=L).to(DEV)
    net16 = bench.build_network(900, 0, num_levels=L).to(DEV).to(torch.bfloat16)
    mem = {}
    net32.encoder.register_forward_hook(lambda m, i, o: mem.__setitem__("fp32", o.detach().float().clone()))
    net16.encoder.register_forward_hook(lambda m, i, o: mem.__setitem__("bf16", o.detach().float().clone()))
    with torch.no_grad():
        o32 = net32(feats, masks, pos)
        o16 = net16([f.to(torch.bfloat16) for f in feats], masks, [p.to(torch.bfloat16) for p in pos])
    torch.cuda.synchronize()
    assert mem["fp32"].shape == (1, 204098, 256)
    rel = ((mem["bf16"] - mem["fp32"]).norm() / mem["fp32"].norm()).item()
    print("focalnet encoder memory bf16 vs fp32: relative L2", rel)
    assert rel < 2.0 ** -5
    assert o16[0].shape == (6, 1, 900, 91) and o16[1].shape == (6, 1, 900, 4) and o16[3].shape == (1, 900, 4)
    assert all(torch.isfinite(t.float()).all() for t in o16[:4]) and all(torch.isfinite(t).all() for t in o32[:4])
    assert ((o16[1].float() >= 0) & (o16[1].float() <= 1)).all()


def test_fp32_self_attention_under_gemm_tuning():
    """Round 2's GPU memory fault (gpurun_out/r02/fullsize_1.log) came from PyTorch's TunableOp tuning the fp32 decoder
    self-attention of ONE image: q / k were column-slice views of the packed projection, which for B = 1 fold into a
    strided-batched GEMM with batch stride 32 and leading dimension 512 -- overlapping matrices -- and TunableOp sizes its
    scratch copy of such an operand without the leading dimension (GemmStridedBatchedParams::GetSizeA), so candidate kernels
    read past its end.  RelationSelfAttention now hands the library dense per-head operands; this test runs exactly that
    call (B = 1, N = 900, float bias) with tuning ON, once, and checks the result against the untuned run."""
    from relation_detr_amd.self_attn import RelationSelfAttention
    torch.manual_seed(3)
    att = RelationSelfAttention(256, 8).to(DEV).eval()
    x, v = torch.randn(1, 900, 256, device=DEV), torch.randn(1, 900, 256, device=DEV)
    bias = torch.randn(8, 900, 900, device=DEV)
    with torch.no_grad():
        want = att(query=x, key=x, value=v, attn_mask=bias)[0].clone()
        was_on, was_tuning = torch.cuda.tunable.is_enabled(), torch.cuda.tunable.tuning_is_enabled()
        try:
            torch.cuda.tunable.set_max_tuning_duration(30)
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.tuning_enable(True)
            got = att(query=x, key=x, value=v, attn_mask=bias)[0].clone()
            torch.cuda.synchronize()
        finally:
            torch.cuda.tunable.tuning_enable(was_tuning)
            torch.cuda.tunable.enable(was_on)
    assert (got - want).abs().max().item() <= 1e-4


def test_relation_bias_config4_900_queries():
    from oracle import torch_ref
    from relation_detr_amd import ops
    g = torch.Generator().manual_seed(900)
    B, N = 4, 900
    boxes = torch.cat([torch.rand(B, N, 2, generator=g), torch.rand(B, N, 2, generator=g) * 0.49 + 0.01], -1)
    w = (torch.rand(8, 64, 1, 1, generator=g) - 0.5) * 0.25
    b = (torch.rand(8, generator=g) - 0.5) * 0.25
    out = ops.relation_bias(boxes.to(DEV), boxes.to(DEV), w.to(DEV), b.to(DEV)).cpu()
    assert out.shape == (B, 8, N, N)
    for i in range(B):                                       # one image at a time bounds the host's [N, N, 64] feature tensor
        ref32 = torch_ref.relation_bias(boxes[i:i + 1], boxes[i:i + 1], w, b)
        ref64 = torch_ref.relation_bias(boxes[i:i + 1].double(), boxes[i:i + 1].double(), w.double(), b.double()).float()
        assert (out[i:i + 1] - ref32).abs().max().item() <= 1e-4
        assert (out[i:i + 1] - ref64).abs().max().item() <= 1e-4


def test_fused_bf16_self_attention_on_pinned_fixture(golden):
    """g6 = the reference's nn.MultiheadAttention call (relation_transformer.py:452-461) with the float relation bias, a bool
    mask, both, or none.  The bf16 module path (in-projection GEMMs + rdetr_relation_attention_bf16 + out-projection) and the
    bare kernel on the fixture's own q / k / v."""
    from relation_detr_amd import ops
    from relation_detr_amd.self_attn import RelationSelfAttention
    g = golden("g6_self_attn.npz")
    att = RelationSelfAttention(256, 8)
    att.load_state_dict({"in_proj_weight": T(g["in_proj_weight"]), "in_proj_bias": T(g["in_proj_bias"]),
                         "out_proj.weight": T(g["out_proj_weight"]), "out_proj.bias": T(g["out_proj_bias"])})
    att16 = att.to(DEV).to(torch.bfloat16).eval()
    qp, vv = T(g["qp"]).to(DEV), T(g["vv"]).to(DEV)
    rb, bm = T(g["rel_bias"]).to(DEV), T(g["bool_mask"]).to(DEV)
    cases = {"out_bias": rb, "out_none": None, "out_bool": bm, "out_bias_inf": rb.masked_fill(bm, float("-inf"))}
    with torch.no_grad():
        for key, mask in cases.items():
            out = att16(query=qp.to(torch.bfloat16), key=qp.to(torch.bfloat16), value=vv.to(torch.bfloat16), attn_mask=mask,
                        need_weights=False)[0].float().cpu().numpy()
            ref = g[key]
            # three bf16 GEMMs around the kernel: 2^-6 of the output scale on top of the kernel's own 2^-7 |ref| + 4e-3
            bound = 2.0 ** -7 * np.abs(ref) + 4e-3 + 2.0 ** -6 * np.abs(ref).max()
            assert (np.abs(out - ref) <= bound).all(), (key, np.abs(out - ref).max())
    # the bare kernel against the oracle's attention on the same bf16-rounded projections
    from oracle import torch_ref
    B, N, C = qp.shape
    w, bvec = T(g["in_proj_weight"]), T(g["in_proj_bias"])
    q = torch.nn.functional.linear(T(g["qp"]), w[:C], bvec[:C]).to(torch.bfloat16)
    k = torch.nn.functional.linear(T(g["qp"]), w[C:2 * C], bvec[C:2 * C]).to(torch.bfloat16)
    v = torch.nn.functional.linear(T(g["vv"]), w[2 * C:], bvec[2 * C:]).to(torch.bfloat16)
    got = ops.relation_attention(q.to(DEV), k.to(DEV), v.to(DEV), 8, rb, None, 32 ** -0.5).float().cpu()
    qh, kh, vh = (t.float().view(B, N, 8, 32).transpose(1, 2) for t in (q, k, v))
    probs = torch_ref.bias_softmax((qh @ kh.transpose(-1, -2) * 32 ** -0.5).reshape(B * 8, N, N), T(g["rel_bias"]))
    want = (probs.view(B, 8, N, N) @ vh).transpose(1, 2).reshape(B, N, C)
    assert ((got - want).abs() <= 2.0 ** -7 * want.abs() + 4e-3).all()
