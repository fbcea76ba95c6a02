"""Deterministic synthetic RQAE weights and residual-stream activations.

Everything here is numpy ``Generator(PCG64)`` so the same seed gives the same
bits in the build container (where golden vectors are captured from the
reference) and on the GPU box (where the HIP path is checked against them).
torch RNG / ``nn.Linear`` default init are *not* bit-stable across builds, so
they are never used for parity inputs.

Shapes follow the reference: ``RQAE.__init__`` (rqae/model.py:18-73) and its
state-dict layout (``codebook (L,K,c)``, ``layers.{l}.0.weight (c,d)`` ...).
"""
from __future__ import annotations

from itertools import product

import numpy as np

__all__ = [
    "fsq_codebook",
    "weights",
    "activations",
    "state_dict_from_weights",
    "round_to_bf16",
    "sha256_of",
    "CONFIGS",
]

# BASELINE.json configs -> reference ctor args.
CONFIGS = {
    # name: (dim, codebook_dim, codebook_size, num_quantizers)
    "C1": dict(dim=256, codebook_dim=4, codebook_size=4, num_quantizers=4),
    "C2": dict(dim=2304, codebook_dim=5, codebook_size=4, num_quantizers=8),
    "C4": dict(dim=3584, codebook_dim=13, codebook_size=2, num_quantizers=16),
    "C5": dict(dim=2304, codebook_dim=4, codebook_size=8, num_quantizers=32),
    # pretrained shape (rqae/model.py:18-24 defaults), depth cut for tests
    "P": dict(dim=2304, codebook_dim=4, codebook_size=5, num_quantizers=16),
}


def fsq_codebook(codebook_dim: int, codebook_size: int, method: str = "round_fsq") -> np.ndarray:
    """One level's FSQ grid, float64 -> float32, as rqae/model.py:63-72 builds it.

    ``product(linspace(-1,1,cbs), repeat=c)``; ``round_fsq`` divides each row by
    its L2 norm, leaving an all-zero row at zero (its norm is replaced by 1).
    The reference broadcasts this one grid to every level.
    """
    grid = np.linspace(-1, 1, codebook_size)
    cb = np.array(list(product(grid, repeat=codebook_dim)), dtype=np.float64)
    if method == "round_fsq":
        norms = np.linalg.norm(cb, axis=-1, keepdims=True)
        norms = np.where(norms == 0, 1.0, norms)
        cb = cb / norms
    elif method != "fsq":
        raise ValueError(f"not an FSQ method: {method}")
    return cb.astype(np.float32)


def weights(seed: int, dim: int, codebook_dim: int, num_quantizers: int,
            codebook_size: int | None = None, method: str = "round_fsq",
            free_codebook_rows: int | None = None) -> dict:
    """Random RQAE parameters with ``nn.Linear``'s default law, U(-1/sqrt(fan_in), +).

    Returns float32 arrays ``w_in (L,c,d)``, ``b_in (L,c)``, ``w_out (L,d,c)``,
    ``b_out (L,d)``, ``codebook (L,K,c)``.  For FSQ methods the codebook is the
    reference grid repeated per level; otherwise (``free_codebook_rows`` given)
    per-level random unit rows as in rqae/model.py:46-54,126-131.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    L, c, d = num_quantizers, codebook_dim, dim
    k_in = 1.0 / np.sqrt(d)
    k_out = 1.0 / np.sqrt(c)
    w_in = rng.uniform(-k_in, k_in, size=(L, c, d)).astype(np.float32)
    b_in = rng.uniform(-k_in, k_in, size=(L, c)).astype(np.float32)
    w_out = rng.uniform(-k_out, k_out, size=(L, d, c)).astype(np.float32)
    b_out = rng.uniform(-k_out, k_out, size=(L, d)).astype(np.float32)
    if free_codebook_rows is None:
        assert codebook_size is not None
        cb1 = fsq_codebook(c, codebook_size, method)
        codebook = np.broadcast_to(cb1, (L,) + cb1.shape).copy()
    else:
        cb = rng.standard_normal(size=(L, free_codebook_rows, c))
        n2 = np.zeros(cb.shape[:-1])
        for k in range(c):  # explicit order: independent of numpy's SIMD reduction
            n2 = n2 + cb[..., k] * cb[..., k]
        codebook = (cb / np.sqrt(n2)[..., None]).astype(np.float32)
    return dict(w_in=w_in, b_in=b_in, w_out=w_out, b_out=b_out, codebook=codebook)


def activations(seed: int, n_tokens: int, dim: int) -> np.ndarray:
    """Gemma-like RMS-normalised residual stream, float32 ``(n_tokens, dim)``.

    N(0,1) x a heavy-tailed per-channel scale, four 'massive' channels x30, then
    the row is RMS-normalised and multiplied by (1+w), w ~ N(0, 0.1) -- the shape
    of ``norm(hs)`` that ``RQAE.hook`` feeds to ``forward`` (rqae/model.py:279-281,
    rqae/llm.py:65-66).  Only IEEE-exact float64 operations (+ * / sqrt) with an
    explicit summation order are used after the generator, so the float32 result
    does not depend on the host's libm or SIMD width; fixtures also carry a
    sha256 of the bytes as a guard.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    g = rng.standard_normal(dim)
    scale = 0.3 + g * g
    massive = rng.choice(dim, size=min(4, dim), replace=False)
    scale[massive] *= 30.0
    w = 0.1 * rng.standard_normal(dim)
    z = rng.standard_normal((n_tokens, dim)) * scale
    # sequential (outer-axis) reduction: one accumulator per row, channels in order
    ms = np.add.reduce(np.ascontiguousarray((z * z).T), axis=0) / dim
    z = z / np.sqrt(ms + 1e-6)[:, None]
    z = z * (1.0 + w)
    return z.astype(np.float32)


def sha256_of(a: np.ndarray) -> str:
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest-even bfloat16 and return it as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def state_dict_from_weights(w: dict):
    """torch state_dict with exactly the reference's keys."""
    import torch

    L = w["w_in"].shape[0]
    sd = {
        "codebook": torch.from_numpy(w["codebook"].copy()),
        "codebook_counts": torch.zeros(L, w["codebook"].shape[1]),
    }
    for l in range(L):
        sd[f"layers.{l}.0.weight"] = torch.from_numpy(w["w_in"][l].copy())
        sd[f"layers.{l}.0.bias"] = torch.from_numpy(w["b_in"][l].copy())
        sd[f"layers.{l}.1.weight"] = torch.from_numpy(w["w_out"][l].copy())
        sd[f"layers.{l}.1.bias"] = torch.from_numpy(w["b_out"][l].copy())
    return sd
