"""Kernel-routing switches of the harness and the MSDA module, as ONE immutable object.

Every switch selects between a fused HIP kernel and the unfused sequence it replaces (same results within the stated
bounds); the defaults are what bench.py measures.  The object is read from the environment ONCE, when this module is
imported (``RDETR_<FIELD>=0|1`` -- how the same-box A/B scripts under tools/ flip one switch per process), and every
module of the package copies the current object when it is CONSTRUCTED (``self.options``).  Nothing on a forward path reads
the environment: a captured HIP graph and the eager run of the same module object always take the same route.

    from relation_detr_amd import options
    options.apply(net, rel_fused=False)          # A/B inside one process: replace the object on every sub-module
    with options.override(ffn_fused=False):      # ... or for the modules constructed inside the block
        net = build_relation_transformer(...)
"""
from __future__ import annotations

import contextlib
import dataclasses
import os
from typing import Mapping, Optional


@dataclasses.dataclass(frozen=True)
class Options:
    # --- transformer.py -------------------------------------------------------------------------------------------------
    linear_k256: bool = False        # encoder FFN linear1 through csrc/linear.hip (opt-in: -1 % in the two-group replay)
    ffn_fused: bool = True           # fused feed-forward block (csrc/ffn.hip) for tall bf16 inputs
    ffn_ln: bool = False             # ... with the closing add+LayerNorm in its epilogue (opt-in: 3-6 % slower in the stack)
    ln_pos: bool = True              # encoder: norm2 also emits the next layer's query + pos
    decoder_ln_pos: bool = True      # decoder: norm2 also emits the cross-attention's query + query_pos
    decoder_entry: bool = True       # decoder layer entry (reference scaling + sine embedding, scaled query_pos) as 2 kernels
    decoder_tail: bool = True        # reference-point head + query scale + product + query sum as one kernel (csrc/qpos.hip)
    decoder_value_batched: bool = False  # the six cross-attention value projections as ONE GEMM ahead of the decoder's chain
                                         # (opt-in: -1 % in the two-group replay, profiles/r03/ab_stack_decoder_value_batched_null_result.txt)
    box_head: bool = True            # box head + refinement as one kernel (csrc/mlp.hip)
    rel_fused: bool = True           # relation bias generated inside the attention kernel (csrc/attn_rel.hip)
    pyramid_points: bool = True      # valid ratios / reference points / proposal logits as two kernels
    topk: bool = True                # own total-order top-k (csrc/topk.hip) instead of torch.topk
    detections_kernel: bool = True   # PostProcess after its top-k as one kernel
    # --- ms_deform_attn.py ----------------------------------------------------------------------------------------------
    mask_in_kernel: Optional[str] = None   # None = by shape; "always" / "never" (A/B aid)
    value_head_major: bool = True    # bf16 encoder shape: gather on the head-major value [B,H,S,D]
    value_proj_hm: bool = True       # ... written by the value projection's own epilogue
    merged_proj: bool = True         # sampling_offsets + attention_weights as one GEMM
    proj_ln: bool = True             # output_proj + residual + LayerNorm in one kernel
    encoder_proj: bool = True        # value_proj (head-major) + the merged query projection of an encoder layer in one kernel

    @classmethod
    def from_env(cls, env: Mapping[str, str] = os.environ) -> "Options":
        kw = {}
        for f in dataclasses.fields(cls):
            raw = env.get("RDETR_" + f.name.upper())
            if raw is None:
                continue
            if f.name == "mask_in_kernel":
                if raw not in ("always", "never"):
                    raise ValueError("RDETR_MASK_IN_KERNEL must be 'always' or 'never'")
                kw[f.name] = raw
            elif raw in ("0", "1"):
                kw[f.name] = raw == "1"
            else:
                raise ValueError(f"RDETR_{f.name.upper()} must be 0 or 1, got {raw!r}")
        return cls(**kw)


_current = Options.from_env()


def get() -> Options:
    """The process-level object modules copy at construction."""
    return _current


@contextlib.contextmanager
def override(**changes):
    """Modules CONSTRUCTED inside the block get ``replace(current, **changes)``."""
    global _current
    saved = _current
    _current = dataclasses.replace(saved, **changes)
    try:
        yield _current
    finally:
        _current = saved


def apply(module, **changes):
    """Replace the options object of ``module`` and all its sub-modules (those that hold one); returns ``module``."""
    for m in module.modules():
        if isinstance(getattr(m, "options", None), Options):
            m.options = dataclasses.replace(m.options, **changes)
    return module
