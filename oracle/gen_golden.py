"""Generate tests/golden/*.npz by running the REFERENCE's own Python (build container only).

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

The reference cannot travel to the GPU box, so its outputs are frozen here as small data
fixtures (inputs + expected outputs only -- no reference source text).  The import recipe is
SURVEY.md Appendix A: a stub ``torchvision`` exposing ``_is_tracing`` and a stub ``util.misc``
exposing the reference's ``inverse_sigmoid``; everything else is the reference's unmodified code.

Fixtures (see tests/golden/README.md):
  g1_msda_core.npz      core fwd, 4 levels, edge-case locations        (ms_deform_attn.py:159-212)
  g2_msda_core_bwd.npz  autograd grads through the same fallback       (ms_deform_attn.py:159-212)
  g3_msda_core_l5.npz   5-level variant (FocalNet-style pyramid)
  g4_msda_module.npz    nn.Module fwd, 2-d and 4-d reference points, padding mask, state_dict
  g5_relation.npz       box_rel_encoding, sine embed, PositionRelationEmbedding(16, 8)
  g6_self_attn.npz      nn.MultiheadAttention with float relation bias / bool mask / None
  g7_transformer.npz    RelationTransformer eval forward (2 enc + 3 dec layers, d_ffn 64, 24 queries) on a padded
                        3-image batch; weights are tests/helpers.py::synthetic_state_dict (not stored)
  g8_transformer_train.npz  the same network in TRAINING mode: denoising queries in front of the matching queries with
                        their visibility mask, the hybrid (one-to-many) branch, all 8 outputs, and autograd gradients
                        of a fixed linear functional of the outputs   (``python oracle/gen_golden.py g8`` writes only it)
  g9_transformer_l5.npz the eval forward with FIVE feature levels (the FocalNet-L configuration's level count,
                        configs/relation_detr/relation_detr_focalnet_large_lrf_fl4_1200_2000.py:24,35-70) on a padded
                        3-image batch                                 (``python oracle/gen_golden.py g9`` writes only it)
"""
import ast
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("RELATION_DETR_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    tv = types.ModuleType("torchvision")
    tv._is_tracing = lambda: False
    sys.modules["torchvision"] = tv
    sys.path.insert(0, REF)
    tree = ast.parse(open(os.path.join(REF, "util/misc.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "inverse_sigmoid"][0]
    um = types.ModuleType("util.misc")
    um.torch = torch
    exec(compile(ast.Module([fn], []), "util/misc.py", "exec"), um.__dict__)
    up = types.ModuleType("util")
    up.__path__ = []
    sys.modules["util"], sys.modules["util.misc"] = up, um
    from models.bricks import ms_deform_attn as M, position_encoding as P, relation_transformer as R
    return M, P, R


def pyramid(shapes):
    shapes = torch.tensor(shapes, dtype=torch.int64)
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    return shapes, start, int((shapes[:, 0] * shapes[:, 1]).sum())


def edge_locations(g, B, Nq, H, shapes, P):
    """U(-0.1, 1.1) locations with hand-placed edge cases: exact 0 / 1, half-pixel borders,
    one pixel outside on every side, far outside."""
    L = shapes.shape[0]
    loc = torch.rand(B, Nq, H, L, P, 2, generator=g) * 1.2 - 0.1
    for l in range(L):
        h, w = [float(v) for v in shapes[l]]
        specials = [
            (0.0, 0.0), (1.0, 1.0), (0.0, 1.0), (1.0, 0.0),
            (0.5 / w, 0.5 / h), (1 - 0.5 / w, 1 - 0.5 / h),       # exact first / last pixel centres
            (-0.5 / w, 0.3), (0.3, -0.5 / h),                      # x = -1 / y = -1 pixel: excluded boundary
            (-0.49 / w, 0.3), (0.3, -0.49 / h),                    # just inside the (-1, .) boundary
            (1 + 0.5 / w, 0.6), (0.6, 1 + 0.5 / h),                # x = W / y = H: excluded boundary
            (1 + 0.49 / w, 0.6), (0.6, 1 + 0.49 / h),
            (-3.0, 0.5), (0.5, 7.0), (1.5 / w, 2.5 / h), (2.0 / w, 3.0 / h),   # integer + .0 / .5 coords
        ]
        for i, (x, y) in enumerate(specials):
            q = i % Nq
            loc[:, q, (i + l) % H, l, i % P, 0] = x
            loc[:, q, (i + l) % H, l, i % P, 1] = y
    return loc


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    M, Pmod, R = import_reference()
    g = torch.Generator().manual_seed(20240601)
    H, D, P = 8, 32, 4

    # ---- G1 / G2: core forward + autograd backward, 4 levels --------------------------------
    shapes, start, S = pyramid([(12, 20), (6, 10), (3, 5), (2, 3)])
    B, Nq = 2, 37
    value = torch.randn(B, S, H, D, generator=g)
    loc = edge_locations(g, B, Nq, H, shapes, P)
    attn = torch.softmax(torch.randn(B, Nq, H, shapes.shape[0] * P, generator=g), -1).view(B, Nq, H, -1, P)
    out32 = M.multi_scale_deformable_attn_pytorch(value, shapes, loc, attn)
    out64 = M.multi_scale_deformable_attn_pytorch(value.double(), shapes, loc.double(), attn.double())
    np.savez_compressed(os.path.join(OUT, "g1_msda_core.npz"), value=value.numpy(), shapes=shapes.numpy(),
                        level_start=start.numpy(), loc=loc.numpy(), attn=attn.numpy(), out=out32.numpy(),
                        out_f64=out64.float().numpy())   # f64 result rounded once to f32

    v2, l2, a2 = value.clone().requires_grad_(), loc.clone().requires_grad_(), attn.clone().requires_grad_()
    grad_out = torch.randn(B, Nq, H * D, generator=g)
    M.multi_scale_deformable_attn_pytorch(v2, shapes, l2, a2).backward(grad_out)
    v3, l3, a3 = (t.detach().double().requires_grad_() for t in (value, loc, attn))
    M.multi_scale_deformable_attn_pytorch(v3, shapes, l3, a3).backward(grad_out.double())
    np.savez_compressed(os.path.join(OUT, "g2_msda_core_bwd.npz"), grad_out=grad_out.numpy(),
                        grad_value=v2.grad.numpy(), grad_loc=l2.grad.numpy(), grad_attn=a2.grad.numpy(),
                        grad_value_f64=v3.grad.float().numpy(), grad_loc_f64=l3.grad.float().numpy(),
                        grad_attn_f64=a3.grad.float().numpy())

    # ---- G3: 5-level variant ------------------------------------------------------------------
    shapes5, start5, S5 = pyramid([(10, 16), (5, 8), (3, 4), (2, 2), (1, 1)])
    B5, Nq5 = 1, 29
    value5 = torch.randn(B5, S5, H, D, generator=g)
    loc5 = edge_locations(g, B5, Nq5, H, shapes5, P)
    attn5 = torch.softmax(torch.randn(B5, Nq5, H, 5 * P, generator=g), -1).view(B5, Nq5, H, 5, P)
    out5 = M.multi_scale_deformable_attn_pytorch(value5, shapes5, loc5, attn5)
    np.savez_compressed(os.path.join(OUT, "g3_msda_core_l5.npz"), value=value5.numpy(), shapes=shapes5.numpy(),
                        level_start=start5.numpy(), loc=loc5.numpy(), attn=attn5.numpy(), out=out5.numpy())

    # ---- G4: module forward -------------------------------------------------------------------
    mod = M.MultiScaleDeformableAttention(256, 4, 8, 4).eval()
    with torch.no_grad():      # fresh modules have zero offset/attention weights: randomise to exercise the path
        mod.sampling_offsets.weight.copy_(torch.randn(mod.sampling_offsets.weight.shape, generator=g) * 0.05)
        mod.attention_weights.weight.copy_(torch.randn(mod.attention_weights.weight.shape, generator=g) * 0.1)
        mod.attention_weights.bias.copy_(torch.randn(mod.attention_weights.bias.shape, generator=g) * 0.1)
        mod.value_proj.bias.copy_(torch.randn(256, generator=g) * 0.1)
        mod.output_proj.bias.copy_(torch.randn(256, generator=g) * 0.1)
    Bm, Nm = 2, 19
    feat = torch.randn(Bm, S, 256, generator=g)
    q_enc = torch.randn(Bm, S, 256, generator=g)
    q_dec = torch.randn(Bm, Nm, 256, generator=g)
    mask = torch.zeros(Bm, S, dtype=torch.bool)
    mask[1, 200:240] = True
    mask[1, 290:300] = True
    ref2 = torch.rand(Bm, S, 4, 2, generator=g)
    ref4 = torch.cat([torch.rand(Bm, Nm, 4, 2, generator=g) * 0.8 + 0.1,
                      torch.rand(Bm, Nm, 4, 2, generator=g) * 0.5 + 0.02], -1)
    with torch.no_grad():
        out_enc = mod(query=q_enc, reference_points=ref2, value=feat, spatial_shapes=shapes,
                      level_start_index=start, key_padding_mask=mask)
        out_dec = mod(query=q_dec, reference_points=ref4, value=feat, spatial_shapes=shapes,
                      level_start_index=start, key_padding_mask=None)
    sd = {"sd." + k: v.numpy() for k, v in mod.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "g4_msda_module.npz"), shapes=shapes.numpy(), level_start=start.numpy(),
                        feat=feat.numpy(), q_enc=q_enc.numpy(), q_dec=q_dec.numpy(), mask=mask.numpy(),
                        ref2=ref2.numpy(), ref4=ref4.numpy(), out_enc=out_enc.numpy(), out_dec=out_dec.numpy(), **sd)

    # ---- G5: relation embedding ---------------------------------------------------------------
    rel = R.PositionRelationEmbedding(16, 8).eval()
    Br, N1, N2 = 2, 23, 31
    src = torch.cat([torch.rand(Br, N1, 2, generator=g), torch.rand(Br, N1, 2, generator=g) * 0.49 + 0.01], -1)
    tgt = torch.cat([torch.rand(Br, N2, 2, generator=g), torch.rand(Br, N2, 2, generator=g) * 0.49 + 0.01], -1)
    tiny_src, tiny_tgt = src.clone(), tgt.clone()
    tiny_src[..., 2:] = 1e-4 * (1 + torch.rand(Br, N1, 2, generator=g))
    tiny_tgt[0, :, 2:] = 1e-4 * (1 + torch.rand(N2, 2, generator=g))
    with torch.no_grad():
        enc = R.box_rel_encoding(src, tgt)
        sine = Pmod.get_sine_pos_embed(enc, num_pos_feats=16, temperature=10000.0, scale=100.0, exchange_xy=False)
        bias = rel(src, tgt)
        bias_self = rel(src)
        enc_tiny = R.box_rel_encoding(tiny_src, tiny_tgt)
        bias_tiny = rel(tiny_src, tiny_tgt)
        rel64 = R.PositionRelationEmbedding(16, 8).double().eval()
        rel64.load_state_dict({k: v.double() for k, v in rel.state_dict().items()})
        Pmod.get_dim_t.cache_clear()
        bias_f64 = rel64(src.double(), tgt.double())
        bias_tiny_f64 = rel64(tiny_src.double(), tiny_tgt.double())
        Pmod.get_dim_t.cache_clear()
    np.savez_compressed(os.path.join(OUT, "g5_relation.npz"), src=src.numpy(), tgt=tgt.numpy(),
                        tiny_src=tiny_src.numpy(), tiny_tgt=tiny_tgt.numpy(), enc=enc.numpy(), sine=sine.numpy(),
                        bias=bias.numpy(), bias_self=bias_self.numpy(), enc_tiny=enc_tiny.numpy(),
                        bias_tiny=bias_tiny.numpy(), bias_f64=bias_f64.float().numpy(), bias_tiny_f64=bias_tiny_f64.float().numpy(),
                        proj_weight=rel.pos_proj[0].weight.detach().numpy(),
                        proj_bias=rel.pos_proj[0].bias.detach().numpy())

    # ---- G6: decoder self-attention with the relation bias as float attn_mask -------------------
    Ba, Na = 2, 50
    mha = torch.nn.MultiheadAttention(256, 8, dropout=0.0, batch_first=True).eval()
    with torch.no_grad():
        mha.in_proj_bias.copy_(torch.randn(768, generator=g) * 0.1)
        mha.out_proj.bias.copy_(torch.randn(256, generator=g) * 0.1)
    qp = torch.randn(Ba, Na, 256, generator=g)
    vv = torch.randn(Ba, Na, 256, generator=g)
    boxes_a = torch.cat([torch.rand(Ba, Na, 2, generator=g), torch.rand(Ba, Na, 2, generator=g) * 0.4 + 0.02], -1)
    boxes_b = torch.cat([torch.rand(Ba, Na, 2, generator=g), torch.rand(Ba, Na, 2, generator=g) * 0.4 + 0.02], -1)
    bool_mask = torch.zeros(Na, Na, dtype=torch.bool)
    bool_mask[:20, 20:] = True
    bool_mask[20:, :20] = True
    with torch.no_grad():
        rb = rel(boxes_a, boxes_b).flatten(0, 1)                       # [B*8, N, N] as relation_transformer.py:372
        out_bias = mha(query=qp, key=qp, value=vv, attn_mask=rb, need_weights=False)[0]
        out_none = mha(query=qp, key=qp, value=vv, attn_mask=None, need_weights=False)[0]
        out_bool = mha(query=qp, key=qp, value=vv, attn_mask=bool_mask, need_weights=False)[0]
        rb_inf = rb.clone().masked_fill_(bool_mask, float("-inf"))    # training: relation_transformer.py:373-374
        out_bias_inf = mha(query=qp, key=qp, value=vv, attn_mask=rb_inf, need_weights=False)[0]
    np.savez_compressed(os.path.join(OUT, "g6_self_attn.npz"), qp=qp.numpy(), vv=vv.numpy(), rel_bias=rb.numpy(),
                        bool_mask=bool_mask.numpy(), out_bias=out_bias.numpy(), out_none=out_none.numpy(),
                        out_bool=out_bool.numpy(), out_bias_inf=out_bias_inf.numpy(),
                        in_proj_weight=mha.in_proj_weight.detach().numpy(),
                        in_proj_bias=mha.in_proj_bias.detach().numpy(),
                        out_proj_weight=mha.out_proj.weight.detach().numpy(),
                        out_proj_bias=mha.out_proj.bias.detach().numpy())

    # ---- G7: whole transformer, eval path ---------------------------------------------------------
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from helpers import synthetic_state_dict
    nlev, nq = 4, 24
    enc = R.RelationTransformerEncoder(R.RelationTransformerEncoderLayer(256, 64, 0.0, 8, torch.nn.ReLU(inplace=True), nlev, 4), 2)
    dec = R.RelationTransformerDecoder(R.RelationTransformerDecoderLayer(256, 64, 8, 0.0, torch.nn.ReLU(inplace=True), nlev, 4), 3, 11)
    tr = R.RelationTransformer(enc, dec, 11, nlev, nq, 30).eval()
    tr.load_state_dict(synthetic_state_dict(tr.state_dict()))
    Bt = 3
    feats = [torch.randn(Bt, 256, h, w, generator=g) for h, w in shapes.tolist()]
    pos = [torch.randn(Bt, 256, h, w, generator=g) * 0.5 for h, w in shapes.tolist()]
    masks = []
    for h, w in shapes.tolist():                                   # image 1 padded right, image 2 padded bottom+right
        mk = torch.zeros(Bt, h, w, dtype=torch.bool)
        mk[1, :, int(round(w * 0.75)):] = True
        mk[2, int(round(h * 0.6)):, :] = True
        mk[2, :, int(round(w * 0.9)):] = True
        masks.append(mk)
    with torch.no_grad():
        oc, ob, ec, eb = tr(feats, masks, pos)[:4]
    np.savez_compressed(os.path.join(OUT, "g7_transformer.npz"), shapes=shapes.numpy(),
                        **{f"feat{i}": f.numpy() for i, f in enumerate(feats)},
                        **{f"pos{i}": f.numpy() for i, f in enumerate(pos)},
                        **{f"mask{i}": f.numpy() for i, f in enumerate(masks)},
                        param_names=np.array(list(tr.state_dict().keys())),
                        param_shapes=np.array([";".join(map(str, v.shape)) for v in tr.state_dict().values()]),
                        out_classes=oc.numpy(), out_coords=ob.numpy(), enc_classes=ec.numpy(), enc_coords=eb.numpy())

    golden_g8(R)
    golden_g9(R)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


def dn_visibility_mask(group_size: int, groups: int, num_queries: int) -> torch.Tensor:
    """The mask the reference's denoising generator hands to the transformer (models/bricks/denoising.py:66-78; the
    generator module itself needs torchvision.ops, absent here): True = may not attend.  Matching queries never see
    denoising queries; a denoising group sees only itself (and the matching queries)."""
    ndn = group_size * groups
    m = torch.zeros(ndn + num_queries, ndn + num_queries, dtype=torch.bool)
    m[ndn:, :ndn] = True
    for i in range(groups):
        lo, hi = group_size * i, group_size * (i + 1)
        m[lo:hi, :lo] = True
        m[lo:hi, hi:ndn] = True
    return m


def golden_g8(R):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from helpers import synthetic_state_dict, functional_weights, G8_FULL_GRADS
    g = torch.Generator().manual_seed(20240608)
    shapes, _, _ = pyramid([(12, 20), (6, 10), (3, 5), (2, 3)])
    nlev, nq, nhyb = 4, 24, 30
    enc = R.RelationTransformerEncoder(R.RelationTransformerEncoderLayer(256, 64, 0.0, 8, torch.nn.ReLU(inplace=True), nlev, 4), 2)
    dec = R.RelationTransformerDecoder(R.RelationTransformerDecoderLayer(256, 64, 8, 0.0, torch.nn.ReLU(inplace=True), nlev, 4), 3, 11)
    tr = R.RelationTransformer(enc, dec, 11, nlev, nq, nhyb).train()
    tr.load_state_dict(synthetic_state_dict(tr.state_dict()))
    Bt, group, groups = 2, 3, 2
    feats = [torch.randn(Bt, 256, h, w, generator=g).requires_grad_(True) for h, w in shapes.tolist()]
    pos = [torch.randn(Bt, 256, h, w, generator=g) * 0.5 for h, w in shapes.tolist()]
    masks = []
    for h, w in shapes.tolist():                                   # image 1 padded right and bottom
        mk = torch.zeros(Bt, h, w, dtype=torch.bool)
        mk[1, :, int(round(w * 0.8)):] = True
        mk[1, int(round(h * 0.7)):, :] = True
        masks.append(mk)
    ndn = group * groups
    dn_label = torch.randn(Bt, ndn, 256, generator=g).requires_grad_(True)
    boxes = torch.cat([torch.rand(Bt, ndn, 2, generator=g) * 0.8 + 0.1, torch.rand(Bt, ndn, 2, generator=g) * 0.3 + 0.05], -1)
    dn_box = torch.log(boxes / (1 - boxes)).requires_grad_(True)    # logit space, as the generator returns them
    attn_mask = dn_visibility_mask(group, groups, nq)
    outs = tr(feats, masks, pos, dn_label, dn_box, attn_mask)
    assert len(outs) == 8 and all(o is not None for o in outs)
    loss = sum((o * functional_weights(o.shape, i)).sum() for i, o in enumerate(outs))
    loss.backward()
    names = [n for n, _ in tr.named_parameters()]
    grads = dict(tr.named_parameters())
    np.savez_compressed(
        os.path.join(OUT, "g8_transformer_train.npz"), shapes=shapes.numpy(),
        **{f"feat{i}": f.detach().numpy() for i, f in enumerate(feats)},
        **{f"pos{i}": f.numpy() for i, f in enumerate(pos)},
        **{f"mask{i}": f.numpy() for i, f in enumerate(masks)},
        dn_label=dn_label.detach().numpy(), dn_box=dn_box.detach().numpy(), attn_mask=attn_mask.numpy(),
        **{f"out{i}": o.detach().numpy() for i, o in enumerate(outs)},
        loss=np.float64(loss.item()),
        grad_names=np.array(names),
        grad_norms=np.array([0.0 if grads[n].grad is None else grads[n].grad.double().norm().item() for n in names]),
        **{f"grad.{n}": grads[n].grad.numpy() for n in G8_FULL_GRADS},
        grad_feat3=feats[3].grad.numpy(), grad_feat0_norm=np.float64(feats[0].grad.double().norm().item()),
        grad_dn_label=dn_label.grad.numpy(), grad_dn_box=dn_box.grad.numpy())


def golden_g9(R):
    """RelationTransformer with num_feature_levels = 5 in eval mode (relation_transformer.py:59-160; the FocalNet-L config's
    level count: the extra level is the neck's stride-2 conv of the last one, channel_mapper.py:43-59 -> (2,3) -> (1,2))."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from helpers import synthetic_state_dict
    g = torch.Generator().manual_seed(20240609)
    shapes, _, _ = pyramid([(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)])
    nlev, nq = 5, 24
    enc = R.RelationTransformerEncoder(R.RelationTransformerEncoderLayer(256, 64, 0.0, 8, torch.nn.ReLU(inplace=True), nlev, 4), 2)
    dec = R.RelationTransformerDecoder(R.RelationTransformerDecoderLayer(256, 64, 8, 0.0, torch.nn.ReLU(inplace=True), nlev, 4), 3, 11)
    tr = R.RelationTransformer(enc, dec, 11, nlev, nq, 30).eval()
    tr.load_state_dict(synthetic_state_dict(tr.state_dict()))
    Bt = 3
    feats = [torch.randn(Bt, 256, h, w, generator=g) for h, w in shapes.tolist()]
    pos = [torch.randn(Bt, 256, h, w, generator=g) * 0.5 for h, w in shapes.tolist()]
    masks = []
    for h, w in shapes.tolist():                                   # image 1 padded right, image 2 padded bottom + right
        mk = torch.zeros(Bt, h, w, dtype=torch.bool)
        if w > 2:
            mk[1, :, int(round(w * 0.75)):] = True
            mk[2, :, int(round(w * 0.9)):] = True
        if h > 1:
            mk[2, int(round(h * 0.6)):, :] = True
        masks.append(mk)
    with torch.no_grad():
        oc, ob, ec, eb = tr(feats, masks, pos)[:4]
    np.savez_compressed(os.path.join(OUT, "g9_transformer_l5.npz"), shapes=shapes.numpy(),
                        **{f"feat{i}": f.numpy() for i, f in enumerate(feats)},
                        **{f"pos{i}": f.numpy() for i, f in enumerate(pos)},
                        **{f"mask{i}": f.numpy() for i, f in enumerate(masks)},
                        param_names=np.array(list(tr.state_dict().keys())),
                        param_shapes=np.array([";".join(map(str, v.shape)) for v in tr.state_dict().values()]),
                        out_classes=oc.numpy(), out_coords=ob.numpy(), enc_classes=ec.numpy(), enc_coords=eb.numpy())


if __name__ == "__main__":
    if sys.argv[1:] == ["g9"]:
        torch.manual_seed(0)
        torch.set_num_threads(4)
        golden_g9(import_reference()[2])
    elif sys.argv[1:] == ["g8"]:
        torch.manual_seed(0)
        torch.set_num_threads(4)
        golden_g8(import_reference()[2])
    else:
        main()
