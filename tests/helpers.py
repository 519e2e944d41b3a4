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
