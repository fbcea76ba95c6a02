"""ctypes front end of librqae_oracle.so (see rqae_oracle.c for what it restates).

TEST INFRASTRUCTURE ONLY -- never imported by rqae_amd/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librqae_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile rqae_oracle.c with gcc (oracle/Makefile).  Returns the .so path."""
    src = os.path.join(_HERE, "rqae_oracle.c")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-s", "-C", _HERE, "-B", "librqae_oracle.so"], check=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int64)
        bp = ctypes.POINTER(ctypes.c_uint8)
        i, i64 = ctypes.c_int, ctypes.c_int64
        lib.rqae_oracle_forward.argtypes = [fp, i64, i, i, i, i, i, fp, fp, fp, fp, fp, fp, ip, fp, i]
        lib.rqae_oracle_forward.restype = None
        lib.rqae_oracle_quantize.argtypes = [fp, i64, i, i, fp, fp, ip, fp, i]
        lib.rqae_oracle_quantize.restype = None
        lib.rqae_oracle_decode.argtypes = [ip, i64, i, i, i, i, i, fp, fp, fp, bp, fp, i]
        lib.rqae_oracle_decode.restype = i
        lib.rqae_oracle_max_threads.restype = i
        lib.rqae_oracle_adjudicate.argtypes = [fp, i64, i, i, i, i, i, fp, fp, fp, fp, fp, ip, ctypes.POINTER(ctypes.c_double), ip, i]
        lib.rqae_oracle_adjudicate.restype = None
        dp = ctypes.POINTER(ctypes.c_double)
        lib.rqae_oracle_codebook_sims.argtypes = [fp, i, i, fp]
        lib.rqae_oracle_codebook_sims.restype = None
        lib.rqae_oracle_subfeatures.argtypes = [fp, fp, fp, i, i, i, i, fp]
        lib.rqae_oracle_subfeatures.restype = None
        lib.rqae_oracle_subfeature_sims_level.argtypes = [fp, i, i, dp]
        lib.rqae_oracle_subfeature_sims_level.restype = None
        lib.rqae_oracle_layer_norms.argtypes = [fp, i, i, i, dp]
        lib.rqae_oracle_layer_norms.restype = None
        _lib = lib
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def max_threads() -> int:
    return int(_load().rqae_oracle_max_threads())


def forward(x, w, max_layers=None, want_margins=False, want_recon=True, threads=0):
    """RQAE.forward (rqae/model.py:199-230) on x[..., d]; w = synth.weights()-style dict.

    Returns (q_out[..., d] float32 | None, indices[..., levels] int64[, margins]).
    """
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    lead, d = x.shape[:-1], x.shape[-1]
    L, c, dd = w["w_in"].shape
    K = w["codebook"].shape[1]
    assert dd == d
    levels = L if max_layers is None else int(min(L, max_layers))
    n = int(np.prod(lead)) if lead else 1
    xa, xp = _f(x.reshape(n, d))
    wi, wip = _f(w["w_in"]); bi, bip = _f(w["b_in"]); cb, cbp = _f(w["codebook"])
    wo, wop = _f(w["w_out"]); bo, bop = _f(w["b_out"])
    q = np.empty((n, d), np.float32) if want_recon else None
    idx = np.empty((n, levels), np.int64)
    mg = np.empty((n, levels), np.float32) if want_margins else None
    lib.rqae_oracle_forward(
        xp, n, d, c, K, L, levels, wip, bip, cbp, wop, bop,
        q.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if want_recon else None,
        idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
        mg.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if want_margins else None,
        int(threads))
    out = (q.reshape(lead + (d,)) if want_recon else None, idx.reshape(lead + (levels,)))
    if want_margins:
        out = out + (mg.reshape(lead + (levels,)),)
    return out


def adjudicate(x, w, codes, threads=0):
    """SURVEY 8(c) Tier-2 adjudicator: RQAE.forward's level loop in float64 ALONG the given code path.

    Returns (gap[..., levels] float64, argmax[..., levels] int64): gap = the double-precision cosine of the best row
    minus that of the given code at that level (0 = the code is the double-precision argmax or a copy of it)."""
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    codes = np.ascontiguousarray(codes, dtype=np.int64)
    lead, d = x.shape[:-1], x.shape[-1]
    levels = codes.shape[-1]
    L, c, dd = w["w_in"].shape
    K = w["codebook"].shape[1]
    assert dd == d and codes.shape[:-1] == lead and levels <= L
    n = int(np.prod(lead)) if lead else 1
    xa, xp = _f(x.reshape(n, d))
    wi, wip = _f(w["w_in"]); bi, bip = _f(w["b_in"]); cb, cbp = _f(w["codebook"])
    wo, wop = _f(w["w_out"]); bo, bop = _f(w["b_out"])
    gap = np.empty((n, levels), np.float64)
    am = np.empty((n, levels), np.int64)
    lib.rqae_oracle_adjudicate(xp, n, d, c, K, L, levels, wip, bip, cbp, wop, bop,
                               codes.reshape(n, levels).ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                               gap.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                               am.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), int(threads))
    return gap.reshape(lead + (levels,)), am.reshape(lead + (levels,))


def quantize(p, codebook_l, want_margins=False, threads=0):
    """RQAE.quantize (rqae/model.py:187-197): p[..., c] -> (cv[..., c], idx[...])."""
    lib = _load()
    p = np.ascontiguousarray(p, dtype=np.float32)
    lead, c = p.shape[:-1], p.shape[-1]
    n = int(np.prod(lead)) if lead else 1
    K = codebook_l.shape[0]
    pa, pp = _f(p.reshape(n, c)); cb, cbp = _f(codebook_l)
    cv = np.empty((n, c), np.float32); idx = np.empty((n,), np.int64)
    mg = np.empty((n,), np.float32) if want_margins else None
    lib.rqae_oracle_quantize(pp, n, c, K, cbp, cv.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                             idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                             mg.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if want_margins else None,
                             int(threads))
    out = (cv.reshape(lead + (c,)), idx.reshape(lead))
    if want_margins:
        out = out + (mg.reshape(lead),)
    return out


def decode(indices, w, layers=None, threads=0):
    """RQAE.decode (rqae/model.py:250-252): indices[..., NQ] -> q[..., d] or None."""
    lib = _load()
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    lead, nq = indices.shape[:-1], indices.shape[-1]
    L, c, d = w["w_in"].shape
    K = w["codebook"].shape[1]
    n = int(np.prod(lead)) if lead else 1
    mask = None
    if layers is not None:
        mask = np.zeros(L, np.uint8)
        for l in range(L):
            if l in layers:
                mask[l] = 1
    cb0, cbp = _f(w["codebook"][0]); wo, wop = _f(w["w_out"]); bo, bop = _f(w["b_out"])
    out = np.empty((n, d), np.float32)
    used = lib.rqae_oracle_decode(
        indices.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n, nq, d, c, K, L, cbp, wop, bop,
        mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)) if mask is not None else None,
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(threads))
    if used == 0:
        return None
    return out.reshape(lead + (d,))


# --- the step either side of forward in the real caller (RQAE.hook) -----------------

def gemma_norm(hs, weight, eps=1e-6):
    """Gemma2 RMSNorm as Gemma2.norm applies it (rqae/llm.py:65-66): fp32
    x * rsqrt(mean(x^2) + eps) * (1 + w).  numpy float32, op by op."""
    x = np.asarray(hs, dtype=np.float32)
    ms = np.mean(x * x, axis=-1, keepdims=True, dtype=np.float32)
    inv = (np.float32(1.0) / np.sqrt(ms + np.float32(eps))).astype(np.float32)
    return (x * inv * (np.float32(1.0) + np.asarray(weight, np.float32))).astype(np.float32)


def gemma_denorm(q, hs, weight, eps=1e-6):
    """Gemma2.denorm (rqae/llm.py:68-73): q / (1+w) / rsqrt(mean(hs^2) + eps)."""
    q = np.asarray(q, np.float32); x = np.asarray(hs, np.float32)
    q = q / (np.float32(1.0) + np.asarray(weight, np.float32))
    ms = np.mean(x * x, axis=-1, keepdims=True, dtype=np.float32)
    inv = (np.float32(1.0) / np.sqrt(ms + np.float32(eps))).astype(np.float32)
    return (q / inv).astype(np.float32)


def hook_reference(hs16, w, norm_weight, skip_bos=True):
    """What RQAE.hook's hook_fn stores (rqae/model.py:276-289) for fp16 hs[B,S,d]."""
    hs = np.asarray(hs16).astype(np.float32)
    normed = gemma_norm(hs, norm_weight)
    q, idx = forward(normed, w)
    new = gemma_denorm(q, hs, norm_weight)
    if skip_bos:
        new[:, 0] = hs[:, 0]
    return dict(original=hs, normed=normed, quantized=q, indices=idx, new=new)


def _lane_tree(v):
    """The 64-lane pairing of rqae_oracle.c:lane_tree on float32 arrays shaped [..., 64]."""
    a = (v[..., :32] + v[..., 32:]).astype(np.float32)
    b = (a[..., :16] + a[..., 16:]).astype(np.float32)
    c = (b[..., :8] + b[..., 8:]).astype(np.float32)
    u = (c[..., :4] + c[..., 7:3:-1]).astype(np.float32)
    return ((u[..., 0] + u[..., 1]).astype(np.float32) + (u[..., 2] + u[..., 3]).astype(np.float32)).astype(np.float32)


def _elem_of(P, s, i, lane):
    """oracle/rqae_oracle.c:elem_of: element of (slice s, step i, lane)."""
    base = s * 64 * P
    q4 = (P // 4) * 4
    if i < q4:
        return base + 256 * (i // 4) + 4 * lane + (i % 4)
    rem, j = P - q4, i - q4
    if rem >= 2 and j < 2:
        return base + 64 * q4 + 2 * lane + j
    return base + 64 * q4 + (128 if rem >= 2 else 0) + lane


def gemma_norm_canonical(hs, weight, eps=1e-6):
    """rqae_hip_gemma_norm (and the norm folded into rqae_hip_encode_gemma) restated.  The row sum of squares
    uses the canonical order of DESIGN.md section 3, the one lin_in uses: d zero-padded to 256*P, 4 slices,
    per (slice, lane) an fma chain over the element steps from +0, the 64-lane tree, slices as
    (s0+s1)+(s2+s3); then / d, rs = 1/sqrt(ms + eps), normed = (x * rs) * (1 + w).  Returns (normed, rs)."""
    x = np.asarray(hs).astype(np.float32)
    lead, d = x.shape[:-1], x.shape[-1]
    x2 = x.reshape(-1, d)
    n = x2.shape[0]
    P = (d + 255) // 256
    xp = np.zeros((n, 256 * P), np.float32)
    xp[:, :d] = x2
    lanes = np.arange(64)
    tot = []
    for s in range(4):
        ss = np.zeros((n, 64), np.float32)
        for i in range(P):
            v = xp[:, [_elem_of(P, s, i, int(l)) for l in lanes]]
            ss = (v.astype(np.float64) * v.astype(np.float64) + ss.astype(np.float64)).astype(np.float32)
        tot.append(_lane_tree(ss))
    total = ((tot[0] + tot[1]).astype(np.float32) + (tot[2] + tot[3]).astype(np.float32)).astype(np.float32)
    ms = (total / np.float32(d)).astype(np.float32)
    rs = (np.float32(1.0) / np.sqrt((ms + np.float32(eps)).astype(np.float32))).astype(np.float32)
    w1 = (np.float32(1.0) + np.asarray(weight, np.float32)).astype(np.float32)
    normed = ((x2 * rs[:, None]).astype(np.float32) * w1).astype(np.float32)
    return normed.reshape(lead + (d,)), rs.reshape(lead)


def gemma_denorm_canonical(q, rs, weight):
    w1 = (np.float32(1.0) + np.asarray(weight, np.float32)).astype(np.float32)
    t = (np.asarray(q, np.float32) / w1).astype(np.float32)
    return (t / np.asarray(rs, np.float32)[..., None]).astype(np.float32)


def intensity(codebook_sims, center, layer_weights, token_indices, layers):
    """RQAEFeature.intensity (rqae/feature.py:102-129) restated, torch-CPU fp16 semantics written out:
    gather sims[..., m] = codebook_sims[center[m], idx[..., m]] (fp16); `sims *= w` = fp16(fp32(s) * fp32(w));
    `cumsum` = sequential fp32 accumulator, every prefix rounded to fp16 (ATen cumsum_cpu_kernel: acc_type of
    Half on CPU is float); `/= w.cumsum()` = fp16(fp32(prefix) / fp32(fp16 prefix of w)); then `[..., layers]`."""
    layers = [int(l) for l in layers]
    M = max(layers) + 1
    T = np.asarray(codebook_sims, np.float16)
    c = np.asarray(center)[:M].astype(np.int64)
    idx = np.asarray(token_indices)[..., :M].astype(np.int64)
    w = np.asarray(layer_weights, np.float16)[:M]
    sims = T[c, idx]
    sims = (sims.astype(np.float32) * w.astype(np.float32)).astype(np.float16)
    acc = np.zeros(sims.shape[:-1], np.float32)
    pref = np.empty_like(sims)
    wacc = np.float32(0.0)
    wc = np.empty(M, np.float16)
    for m in range(M):
        acc = (acc + sims[..., m].astype(np.float32)).astype(np.float32)
        pref[..., m] = acc.astype(np.float16)
        wacc = np.float32(wacc + np.float32(w[m]))
        wc[m] = np.float16(wacc)
    out = (pref.astype(np.float32) / wc.astype(np.float32)).astype(np.float16)
    return out[..., layers]


# --- the model's cached tables ------------------------------------------------

def codebook_sims(codebook0):
    """RQAE.codebook_sims (rqae/model.py:133-143): (K, K) fp16, same bits as the reference on CPU."""
    cb, cbp = _f(codebook0)
    K, c = cb.shape
    out = np.empty((K, K), np.float32)
    _load().rqae_oracle_codebook_sims(cbp, K, c, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out.astype(np.float16)


def subfeatures(w):
    """RQAE.subfeatures (rqae/model.py:145-156): (L, K, d) fp32, same bits as the reference on CPU."""
    cb, cbp = _f(w["codebook"]); wo, wop = _f(w["w_out"]); bo, bop = _f(w["b_out"])
    L, K, c = cb.shape
    d = wo.shape[1]
    out = np.empty((L, K, d), np.float32)
    _load().rqae_oracle_subfeatures(cbp, wop, bop, L, K, c, d, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def subfeature_sims(w, levels=None):
    """RQAE.subfeature_sims (rqae/model.py:158-168) in double precision: (L, K, K) float64 (see rqae_oracle.c)."""
    sub = subfeatures(w)
    L, K, d = sub.shape
    levels = range(L) if levels is None else levels
    out = np.empty((len(levels), K, K), np.float64)
    for i, l in enumerate(levels):
        s, sp = _f(sub[l])
        _load().rqae_oracle_subfeature_sims_level(sp, K, d, out[i].ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out


def layer_norms(w):
    """RQAE.layer_norms (rqae/model.py:170-178) in double precision: (L,) float64."""
    wo, wop = _f(w["w_out"])
    L, d, c = wo.shape
    out = np.empty(L, np.float64)
    _load().rqae_oracle_layer_norms(wop, L, c, d, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out
