"""Shared helpers for the parity tests (synthetic inputs, kink masks)."""
import numpy as np
import torch


def pyramid(shapes):
    shapes = torch.tensor(shapes, dtype=torch.int64)
    areas = shapes[:, 0] * shapes[:, 1]
    start = torch.cat([areas.new_zeros(1), areas.cumsum(0)[:-1]])
    return shapes, start, int(areas.sum())


def kink_mask(loc: np.ndarray, shapes: np.ndarray, tol: float = 1e-3) -> np.ndarray:
    """True where d(out)/d(loc) is well defined: bilinear interpolation is only piecewise
    differentiable, with kinks where the pixel coordinate loc*size-0.5 is an integer.  There the
    one-sided derivative an implementation returns depends on how it rounds the coordinate (the
    reference's grid_sample path computes ((2*loc-1)+1)*size/2-0.5, the CUDA op loc*size-0.5), so
    those measure-zero points are excluded from grad_loc comparisons.  Shape = loc.shape."""
    L = shapes.shape[0]
    size = np.stack([shapes[:, 1], shapes[:, 0]], -1).astype(np.float64)          # (w, h) per level
    pix = loc.astype(np.float64) * size.reshape(1, 1, 1, L, 1, 2) - 0.5
    return np.abs(pix - np.round(pix)) > tol


def make_msda_inputs(B, Nq, shapes, H=8, D=32, P=4, seed=0, spread=0.15, dtype=torch.float32):
    """Seeded value / locations / weights for a given pyramid; locations ~ U(-spread, 1+spread)."""
    g = torch.Generator().manual_seed(seed)
    shapes_t, start, S = pyramid(shapes)
    L = shapes_t.shape[0]
    value = torch.randn(B, S, H, D, generator=g, dtype=dtype)
    loc = torch.rand(B, Nq, H, L, P, 2, generator=g) * (1 + 2 * spread) - spread
    attn = torch.softmax(torch.randn(B, Nq, H, L * P, generator=g), -1).view(B, Nq, H, L, P)
    return value, shapes_t, start, loc, attn


def synthetic_state_dict(reference_state_dict):
    """Deterministic, RNG-free parameter values for a transformer state_dict (name -> tensor), so that the
    reference model (in oracle/gen_golden.py) and the harness under test can be loaded with IDENTICAL weights
    without committing tens of MB of tensors: value = f(parameter name, flat index)."""
    import zlib
    out = {}
    for name, ref in reference_state_dict.items():
        n = ref.numel()
        idx = np.arange(n, dtype=np.float64)
        phase = (zlib.crc32(name.encode()) % 1000) / 1000.0 * 6.283185307179586
        if "norm" in name and name.endswith("weight"):
            v = 1.0 + 0.1 * np.sin(idx * 0.37 + phase)
        elif name.endswith("sampling_offsets.bias"):
            v = 2.0 * np.sin(idx * 0.77 + phase)                                  # offsets of a few pixels
        elif name.endswith("bias"):
            v = 0.05 * np.sin(idx * 0.91 + phase)
        elif ref.dim() >= 2 and "embed" not in name:
            fan_in = int(np.prod(ref.shape[1:]))
            v = np.sin(idx * 0.7071 + phase) * np.sqrt(3.0 / fan_in)
        else:                                                                      # embeddings, level_embeds
            v = 0.7 * np.sin(idx * 0.613 + phase)
        out[name] = torch.from_numpy(v.astype(np.float32)).reshape(ref.shape)
    return out


def functional_weights(shape, i: int) -> torch.Tensor:
    """RNG-free weights of the linear functional g8 differentiates: loss = sum_i <out_i, functional_weights(out_i.shape, i)>
    (shared by oracle/gen_golden.py and the tests, so the fixture need not store them)."""
    n = int(np.prod(shape))
    return torch.from_numpy(np.cos(np.arange(n, dtype=np.float64) * 0.37 + i).astype(np.float32)).reshape(tuple(shape))


# parameters whose full gradients g8 stores (all others: their L2 norms)
G8_FULL_GRADS = ("level_embeds", "tgt_embed.weight", "hybrid_tgt_embed.weight",
                 "decoder.position_relation_embedding.pos_proj.0.weight", "decoder.position_relation_embedding.pos_proj.0.bias",
                 "encoder.layers.0.self_attn.sampling_offsets.bias", "encoder.layers.1.self_attn.attention_weights.bias",
                 "decoder.layers.1.self_attn.in_proj_bias", "decoder.layers.2.cross_attn.sampling_offsets.bias",
                 "decoder.bbox_head.0.layers.2.bias", "hybrid_class_head.bias", "encoder_class_head.bias")
