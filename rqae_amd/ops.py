"""``torch.ops.rqae_hip.{encode, decode, quantize}`` -- the operator form of the hot path.

The same C ABI as the module methods (include/rqae_hip.h) behind dispatcher-visible operators: raw parameter tensors
in, fresh tensors out, launched on the current stream of the input's device, with fake (meta) implementations so that
``torch.compile`` / ``torch.export`` can trace through them.  ``rqae_amd.RQAE`` itself calls the library directly (it
keeps the packed parameter block cached across calls); these operators pack per call, with a one-entry cache
per operator that holds the caller's parameter tensors and their version counters (hit = same objects, same versions).

    q_out, idx = torch.ops.rqae_hip.encode(x, w_in, b_in, codebook, w_out, b_out, max_layers, want_recon)
    out        = torch.ops.rqae_hip.decode(idx, codebook0, w_out, b_out, level_mask)
    cv, idx    = torch.ops.rqae_hip.quantize(p, codebook_l)

Shapes: x [N, d] {f32, bf16, f16}; w_in [L, c, d]; b_in [L, c]; codebook [L, K, c]; w_out [L, d, c]; b_out [L, d];
idx [N, levels] int64; level_mask [L] bool (empty tensor = all levels).  q_out is an empty tensor when want_recon is False.
Values: exactly those of RQAE.forward / decode / quantize on the exact arm (rqae/model.py:187-252).
"""
from typing import Tuple

import torch

from . import _native

_PACK_CACHE = {}


def _cached_block(slot, keyed, build):
    """One-entry cache per operator of the packed parameter block.

    The entry holds the caller's tensors THEMSELVES (strong references) next to their version counters, and a hit needs the
    very same tensor objects at the same versions.  A key of (data_ptr, _version, shape) alone is not enough: a caller that
    stacks its per-level weights per call hands over temporaries, and after an optimizer step the next ``torch.stack`` can land
    at the same allocator address with version 0 -- same key, different weights.  While the entry keeps the
    old tensors alive their storage cannot be reused, and a fresh temporary is a different object: it misses and is packed.
    Inference tensors track no version counter: they are packed on every call."""
    try:
        versions = tuple(t._version for t in keyed)
    except RuntimeError:
        return build()
    hit = _PACK_CACHE.get(slot)
    if hit is not None and len(hit[0]) == len(keyed) and all(a is b for a, b in zip(hit[0], keyed)) and hit[1] == versions:
        return hit[2]
    blk = build()
    _PACK_CACHE[slot] = (tuple(keyed), versions, blk)
    return blk


def _packed(w_in, b_in, codebook, w_out, b_out):
    def build():
        args = [t.detach().float().contiguous() for t in (w_in, b_in, codebook, w_out, b_out)]
        shared = bool((args[2] == args[2][:1]).all())
        return _native.pack(args[0], args[1], args[2], args[3], args[4], shared)

    return _cached_block("encode", (w_in, b_in, codebook, w_out, b_out), build)


@torch.library.custom_op("rqae_hip::encode", mutates_args=())
def encode(x: torch.Tensor, w_in: torch.Tensor, b_in: torch.Tensor, codebook: torch.Tensor, w_out: torch.Tensor,
           b_out: torch.Tensor, max_layers: int, want_recon: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    L, c, d = w_in.shape
    K = codebook.shape[1]
    levels = min(L, max_layers)
    if levels < 1:
        raise RuntimeError("rqae_hip::encode: max_layers leaves no quantizer to run")
    q, idx = _native.encode(_packed(w_in, b_in, codebook, w_out, b_out), (d, c, K, L), x.contiguous(), levels,
                            want_recon=want_recon)
    return (q if q is not None else x.new_empty((0,), dtype=torch.float32)), idx


@encode.register_fake
def _(x, w_in, b_in, codebook, w_out, b_out, max_layers, want_recon):
    L = w_in.shape[0]
    levels = min(L, max_layers)
    q = x.new_empty(x.shape if want_recon else (0,), dtype=torch.float32)
    return q, x.new_empty((x.shape[0], levels), dtype=torch.int64)


@torch.library.custom_op("rqae_hip::decode", mutates_args=())
def decode(idx: torch.Tensor, codebook0: torch.Tensor, w_out: torch.Tensor, b_out: torch.Tensor,
           level_mask: torch.Tensor) -> torch.Tensor:
    L, d, c = w_out.shape
    K = codebook0.shape[0]
    # decode needs only the level-0 codebook and lin_out; the block's lin_in part is packed from zeros.  The cache is keyed on
    # the caller's three tensors, never on the temporaries made here.
    def build():
        cb = codebook0.detach().float().unsqueeze(0).expand(L, K, c).contiguous()
        zin = torch.zeros(L, c, d, device=w_out.device)
        return _native.pack(zin, torch.zeros(L, c, device=w_out.device), cb, w_out.detach().float().contiguous(),
                            b_out.detach().float().contiguous(), True)

    blk = _cached_block("decode", (codebook0, w_out, b_out), build)
    keep = None if level_mask.numel() == 0 else [bool(v) for v in level_mask.tolist()]
    return _native.decode(blk, (d, c, K, L), idx.contiguous(), keep)


@decode.register_fake
def _(idx, codebook0, w_out, b_out, level_mask):
    return idx.new_empty((idx.shape[0], w_out.shape[1]), dtype=torch.float32)


@torch.library.custom_op("rqae_hip::quantize", mutates_args=())
def quantize(p: torch.Tensor, codebook_l: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return _native.quantize(p.float().contiguous(), codebook_l.detach().float().contiguous())


@quantize.register_fake
def _(p, codebook_l):
    return p.new_empty(p.shape, dtype=torch.float32), p.new_empty((p.shape[0],), dtype=torch.int64)
