"""ctypes binding of the C-ABI library ``librqae_hip.so`` (include/rqae_hip.h).

This is the only route from Python to the HIP kernels.  There is no fallback:
if the shared library is missing or an entry point fails, a RuntimeError is
raised.  torch is used here for device memory and the current stream only.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RQAE_HIP_LIBRARY: another build of the same library (tools/: timeline and ablation builds); the product loads its own
SO_PATH = os.environ.get("RQAE_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "librqae_hip.so")

ABI_VERSION = 4
DT_F32, DT_BF16, DT_F16 = 0, 1, 2
IDX_I64, IDX_I32, IDX_U16 = 0, 1, 2

# every symbol include/rqae_hip.h declares
SYMBOLS = (
    "rqae_hip_abi_version",
    "rqae_hip_last_error",
    "rqae_hip_packed_bytes",
    "rqae_hip_pack",
    "rqae_hip_encode",
    "rqae_hip_quantize",
    "rqae_hip_decode",
    "rqae_hip_decode_values",
    "rqae_hip_encode_gemma_supported",
    "rqae_hip_encode_gemma",
    "rqae_hip_gemma_norm",
    "rqae_hip_gemma_denorm_patch",
    "rqae_hip_intensity_prepare",
    "rqae_hip_intensity",
    "rqae_hip_codebook_sims",
    "rqae_hip_subfeatures",
    "rqae_hip_tables_scratch_bytes",
    "rqae_hip_subfeature_sims",
    "rqae_hip_kernel_info",
    "rqae_hip_gram_supported",
    "rqae_hip_encode_gram_supported",
    "rqae_hip_gram_packed_bytes",
    "rqae_hip_gram_pack",
    "rqae_hip_gram_lattice",
    "rqae_hip_normalize_codebooks",
    "rqae_hip_gram_repack_codebooks",
    "rqae_hip_gram_workspace_bytes",
    "rqae_hip_encode_gram",
    "rqae_hip_encode_gram_gemma_supported",
    "rqae_hip_encode_gram_gemma",
    "rqae_hip_encode_gram_gemma_patch_supported",
    "rqae_hip_encode_gram_gemma_patch",
    "rqae_hip_encode_gram_phases",
    "rqae_hip_decode_gram_supported",
    "rqae_hip_decode_gram",
    "rqae_hip_search_supported",
    "rqae_hip_search_table_bytes",
    "rqae_hip_search_query_table",
    "rqae_hip_search_accumulate",
    "rqae_hip_search_rank_workspace_bytes",
    "rqae_hip_search_rank",
    "rqae_hip_search_gather",
)

_lib = None


class NativeLibraryError(RuntimeError):
    pass


class GramOptions(ctypes.Structure):
    """include/rqae_hip.h: rqae_hip_gram_options -- which kernels ONE rqae_hip_encode_gram call launches (-1 = default)."""
    _fields_ = [("size", ctypes.c_int32), ("fused", ctypes.c_int32), ("proj", ctypes.c_int32), ("recon", ctypes.c_int32),
                ("lean_levels", ctypes.c_int32), ("pipe_chunk", ctypes.c_int64)]


def gram_options(fused: int = -1, pipe_chunk: int = -1, proj: int = -1, recon: int = -1, lean_levels: int = -1) -> GramOptions:
    return GramOptions(ctypes.sizeof(GramOptions), fused, proj, recon, lean_levels, pipe_chunk)


def lib():
    """Load librqae_hip.so once; fail loudly when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise NativeLibraryError(
            f"{SO_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "rqae_amd has no CPU fallback.")
    L = ctypes.CDLL(SO_PATH)
    vp, i, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    L.rqae_hip_abi_version.restype = i
    L.rqae_hip_last_error.restype = ctypes.c_char_p
    L.rqae_hip_packed_bytes.argtypes = [i, i, i, i]
    L.rqae_hip_packed_bytes.restype = sz
    L.rqae_hip_pack.argtypes = [vp, i, i, i, i, vp, vp, vp, vp, vp, i, vp]
    L.rqae_hip_pack.restype = i
    L.rqae_hip_encode.argtypes = [vp, i, i, i, i, vp, i, i64, i, vp, vp, vp]
    L.rqae_hip_encode.restype = i
    L.rqae_hip_quantize.argtypes = [vp, i64, i, i, vp, vp, vp, vp]
    L.rqae_hip_quantize.restype = i
    L.rqae_hip_decode.argtypes = [vp, i, i, i, i, vp, i, i64, i, vp, vp, vp, vp]
    L.rqae_hip_decode.restype = i
    L.rqae_hip_decode_values.argtypes = [vp, i, i, i, i, vp, i64, i, vp, vp, vp]
    L.rqae_hip_decode_values.restype = i
    L.rqae_hip_encode_gemma_supported.argtypes = [i, i, i, i]
    L.rqae_hip_encode_gemma_supported.restype = i
    L.rqae_hip_encode_gemma.argtypes = [vp, i, i, i, i, vp, i, vp, ctypes.c_float, i64, i, vp, vp, vp, vp]
    L.rqae_hip_encode_gemma.restype = i
    L.rqae_hip_gemma_norm.argtypes = [vp, i, vp, ctypes.c_float, i64, i, vp, vp, vp]
    L.rqae_hip_gemma_norm.restype = i
    L.rqae_hip_gemma_denorm_patch.argtypes = [vp, vp, vp, i64, i, i, i, vp, i, vp, vp]
    L.rqae_hip_gemma_denorm_patch.restype = i
    L.rqae_hip_intensity_prepare.argtypes = [vp, vp, vp, i, i, vp, vp, vp]
    L.rqae_hip_intensity_prepare.restype = i
    L.rqae_hip_intensity.argtypes = [vp, vp, vp, i, i, vp, i, i64, i, vp, i, vp, vp, vp]
    L.rqae_hip_intensity.restype = i
    L.rqae_hip_codebook_sims.argtypes = [vp, i, i, vp, vp]
    L.rqae_hip_codebook_sims.restype = i
    L.rqae_hip_subfeatures.argtypes = [vp, vp, vp, i, i, i, i, vp, vp]
    L.rqae_hip_subfeatures.restype = i
    L.rqae_hip_tables_scratch_bytes.argtypes = [i, i, i]
    L.rqae_hip_tables_scratch_bytes.restype = sz
    L.rqae_hip_subfeature_sims.argtypes = [vp, vp, vp, i, i, i, i, vp, vp, vp, i, vp]
    L.rqae_hip_subfeature_sims.restype = i
    L.rqae_hip_kernel_info.argtypes = [i, i, i, i, ctypes.c_char_p, sz]
    L.rqae_hip_kernel_info.restype = i
    L.rqae_hip_gram_supported.argtypes = [i, i, i, i, i]
    L.rqae_hip_gram_supported.restype = i
    L.rqae_hip_encode_gram_supported.argtypes = [i, i, i, i, i, i, i]
    L.rqae_hip_encode_gram_supported.restype = i
    L.rqae_hip_gram_packed_bytes.argtypes = [i, i, i, i, i]
    L.rqae_hip_gram_packed_bytes.restype = sz
    L.rqae_hip_gram_pack.argtypes = [vp, i, i, i, i, vp, vp, vp, vp, vp, i, vp, vp]
    L.rqae_hip_gram_pack.restype = i
    L.rqae_hip_normalize_codebooks.argtypes = [vp, i, i, i, i, vp, vp]
    L.rqae_hip_normalize_codebooks.restype = i
    L.rqae_hip_gram_repack_codebooks.argtypes = [vp, i, i, i, i, vp, vp]
    L.rqae_hip_gram_repack_codebooks.restype = i
    L.rqae_hip_gram_lattice.argtypes = [vp, vp]
    L.rqae_hip_gram_lattice.restype = i
    L.rqae_hip_gram_workspace_bytes.argtypes = [i, i, i, i, i64, i]
    L.rqae_hip_gram_workspace_bytes.restype = sz
    L.rqae_hip_encode_gram.argtypes = [vp, i, i, i, i, i, vp, i, i64, i, vp, vp, vp, ctypes.POINTER(GramOptions), vp]
    L.rqae_hip_encode_gram.restype = i
    L.rqae_hip_encode_gram_gemma_supported.argtypes = [i, i, i, i, i, i]
    L.rqae_hip_encode_gram_gemma_supported.restype = i
    L.rqae_hip_encode_gram_gemma.argtypes = [vp, i, i, i, i, i, vp, i, ctypes.c_float, i64, i, vp, vp, vp, vp, ctypes.POINTER(GramOptions), vp]
    L.rqae_hip_encode_gram_gemma.restype = i
    L.rqae_hip_encode_gram_gemma_patch_supported.argtypes = [i, i, i, i, i, i, i, i]
    L.rqae_hip_encode_gram_gemma_patch_supported.restype = i
    L.rqae_hip_encode_gram_gemma_patch.argtypes = [vp, i, i, i, i, i, vp, i, ctypes.c_float, vp, i, i, i64, i, vp, vp, vp,
                                                   ctypes.POINTER(GramOptions), vp]
    L.rqae_hip_encode_gram_gemma_patch.restype = i
    L.rqae_hip_encode_gram_phases.argtypes = [vp, i, i, i, i, i, vp, i, i64, i, vp, vp, vp, ctypes.POINTER(GramOptions), vp,
                                              ctypes.POINTER(ctypes.c_float)]
    L.rqae_hip_encode_gram_phases.restype = i
    L.rqae_hip_decode_gram_supported.argtypes = [i, i, i, i, i, i]
    L.rqae_hip_decode_gram_supported.restype = i
    L.rqae_hip_decode_gram.argtypes = [vp, i, i, i, i, i, vp, i, i64, i, vp, vp, vp]
    L.rqae_hip_decode_gram.restype = i
    L.rqae_hip_search_supported.argtypes = [i, i]
    L.rqae_hip_search_supported.restype = i
    L.rqae_hip_search_table_bytes.argtypes = [i, i]
    L.rqae_hip_search_table_bytes.restype = sz
    L.rqae_hip_search_query_table.argtypes = [vp, i, i, vp, i, i64, i, i, vp, vp]
    L.rqae_hip_search_query_table.restype = i
    L.rqae_hip_search_accumulate.argtypes = [vp, i, i64, i, vp, i, i, i, i, vp, i, vp, vp]
    L.rqae_hip_search_accumulate.restype = i
    L.rqae_hip_search_rank_workspace_bytes.argtypes = [i64, i]
    L.rqae_hip_search_rank_workspace_bytes.restype = sz
    L.rqae_hip_search_rank.argtypes = [vp, i64, i, i, vp, vp, vp, sz, vp]
    L.rqae_hip_search_rank.restype = i
    L.rqae_hip_search_gather.argtypes = [vp, i64, i, i, vp, i, vp, vp, vp]
    L.rqae_hip_search_gather.restype = i
    if L.rqae_hip_abi_version() != ABI_VERSION:
        raise NativeLibraryError(
            f"librqae_hip.so ABI {L.rqae_hip_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().rqae_hip_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"rqae_amd: `{name}` is on {t.device}; the RQAE hot path runs on a ROCm device only "
            "(no CPU fallback). Move the module and its input to 'cuda'.")


_X_DTYPES = {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float16: DT_F16}
_IDX_DTYPES = {torch.int64: IDX_I64, torch.int32: IDX_I32, torch.uint16: IDX_U16}


def packed_bytes(d: int, c: int, K: int, L: int) -> int:
    return int(lib().rqae_hip_packed_bytes(d, c, K, L))


def pack(w_in, b_in, codebook, w_out, b_out, codebook_shared: bool) -> torch.Tensor:
    """Stacked fp32 parameters -> one device block in the kernels' layout."""
    L_, c, d = w_in.shape
    K = codebook.shape[1]
    for n, t in (("w_in", w_in), ("b_in", b_in), ("codebook", codebook), ("w_out", w_out), ("b_out", b_out)):
        _require_cuda(t, n)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"pack: {n} must be contiguous float32")
    nbytes = packed_bytes(d, c, K, L_)
    if nbytes == 0:
        _check(1, "rqae_hip_packed_bytes")
    out = torch.empty(nbytes, dtype=torch.uint8, device=w_in.device)
    with torch.cuda.device(w_in.device):
        rc = lib().rqae_hip_pack(out.data_ptr(), d, c, K, L_, w_in.data_ptr(), b_in.data_ptr(),
                                 codebook.data_ptr(), w_out.data_ptr(), b_out.data_ptr(),
                                 int(bool(codebook_shared)), _stream(w_in.device))
    _check(rc, "rqae_hip_pack")
    return out


def encode(packed, dims, x2d: torch.Tensor, levels: int, want_recon: bool = True):
    """x2d [N,d] {f32,bf16,f16} -> (q_out [N,d] f32 | None, indices [N,levels] i64)."""
    d, c, K, L_ = dims
    _require_cuda(x2d, "x")
    if x2d.dtype not in _X_DTYPES:
        raise RuntimeError(f"encode: unsupported input dtype {x2d.dtype}")
    if x2d.dim() != 2 or x2d.shape[1] != d or not x2d.is_contiguous():
        raise RuntimeError("encode: x must be contiguous [N, d]")
    n = x2d.shape[0]
    q = torch.empty((n, d), dtype=torch.float32, device=x2d.device) if want_recon else None
    idx = torch.empty((n, levels), dtype=torch.int64, device=x2d.device)
    if n == 0:
        return q, idx
    with torch.cuda.device(x2d.device):
        rc = lib().rqae_hip_encode(packed.data_ptr(), d, c, K, L_, x2d.data_ptr(), _X_DTYPES[x2d.dtype],
                                   n, levels, q.data_ptr() if want_recon else None, idx.data_ptr(),
                                   _stream(x2d.device))
    _check(rc, "rqae_hip_encode")
    return q, idx


# ---- the matrix-core arm (csrc/gram.hip; include/rqae_hip.h "gram arm") -------------------------------------------

def gram_supported(dims, codebook_shared: bool) -> bool:
    d, c, K, L_ = dims
    return bool(lib().rqae_hip_gram_supported(d, c, K, L_, int(bool(codebook_shared))))


def encode_gram_supported(dims, codebook_shared: bool, levels: int, want_recon: bool = True) -> bool:
    """One call with ``levels`` levels on the matrix-core arm (the reconstruction has its own depth limit)."""
    d, c, K, L_ = dims
    return bool(lib().rqae_hip_encode_gram_supported(d, c, K, L_, int(bool(codebook_shared)), int(levels), int(bool(want_recon))))


def gram_pack(w_in, b_in, codebook, w_out, b_out, codebook_shared: bool, in_scale: torch.Tensor | None = None) -> torch.Tensor:
    """Stacked fp32 parameters -> the gram arm's own device block (bf16 operand images, Gram blocks).  ``in_scale`` [d] fp32:
    a diagonal map in front of lin_in folded into the W_in image (Gemma's ``1 + norm.weight`` for encode_gram_gemma)."""
    L_, c, d = w_in.shape
    K = codebook.shape[1]
    for n, t in (("w_in", w_in), ("b_in", b_in), ("codebook", codebook), ("w_out", w_out), ("b_out", b_out)):
        _require_cuda(t, n)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"gram_pack: {n} must be contiguous float32")
    nbytes = int(lib().rqae_hip_gram_packed_bytes(d, c, K, L_, int(bool(codebook_shared))))
    if nbytes == 0:
        _check(1, "rqae_hip_gram_packed_bytes")
    out = torch.empty(nbytes, dtype=torch.uint8, device=w_in.device)
    with torch.cuda.device(w_in.device):
        if in_scale is not None:
            in_scale = in_scale.detach().to(device=w_in.device, dtype=torch.float32).contiguous()
            if in_scale.numel() != d:
                raise RuntimeError("gram_pack: in_scale must have d elements")
        rc = lib().rqae_hip_gram_pack(out.data_ptr(), d, c, K, L_, w_in.data_ptr(), b_in.data_ptr(),
                                      codebook.data_ptr(), w_out.data_ptr(), b_out.data_ptr(),
                                      int(bool(codebook_shared)), in_scale.data_ptr() if in_scale is not None else None,
                                      _stream(w_in.device))
    _check(rc, "rqae_hip_gram_pack")
    # an FSQ lattice?  Asked once, here (the one blocking call of the arm); the encode calls then launch the lattice form of the
    # level chain without the scan kernel behind it (include/rqae_hip.h: lean_levels = 5)
    with torch.cuda.device(w_in.device):
        out.lattice_n = int(lib().rqae_hip_gram_lattice(out.data_ptr(), _stream(w_in.device)))
    if out.lattice_n < 0:
        _check(1, "rqae_hip_gram_lattice")
    return out


def normalize_codebooks(codebook: torch.Tensor, levels: int) -> torch.Tensor:
    """RQAE.normalize_codebooks as a ``levels``-level forward applies it (include/rqae_hip.h: rqae_hip_normalize_codebooks): ``codebook``
    [L,K,c] fp32 is renormalised in place ``levels`` times; returns the per-level snapshots [L,K,c] (entries >= levels unspecified)."""
    _require_cuda(codebook, "codebook")
    if codebook.dtype != torch.float32 or not codebook.is_contiguous() or codebook.dim() != 3:
        raise RuntimeError("normalize_codebooks: codebook must be contiguous float32 [L, K, c]")
    L_, K, c = codebook.shape
    used = torch.empty_like(codebook)
    with torch.cuda.device(codebook.device):
        rc = lib().rqae_hip_normalize_codebooks(codebook.data_ptr(), L_, K, c, int(levels), used.data_ptr(), _stream(codebook.device))
    _check(rc, "rqae_hip_normalize_codebooks")
    return used


def gram_repack_codebooks(gpacked: torch.Tensor, dims, codebook_levels: torch.Tensor):
    """The codebook part of a block packed with codebook_shared = False rebuilt from per-level rows [L,K,c] (in place, one launch)."""
    d, c, K, L_ = dims
    if codebook_levels.shape != (L_, K, c) or codebook_levels.dtype != torch.float32 or not codebook_levels.is_contiguous():
        raise RuntimeError("gram_repack_codebooks: per-level rows must be contiguous float32 [L, K, c]")
    with torch.cuda.device(gpacked.device):
        rc = lib().rqae_hip_gram_repack_codebooks(gpacked.data_ptr(), d, c, K, L_, codebook_levels.data_ptr(), _stream(gpacked.device))
    _check(rc, "rqae_hip_gram_repack_codebooks")


def _gram_default_options(gpacked, options):
    """No options from the caller: the defaults, with the scan kernel behind the lattice form left out when ``gram_pack`` has
    learnt that the block holds a lattice."""
    if options is None and getattr(gpacked, "lattice_n", 0) >= 2:
        return gram_options(lean_levels=5)
    return options


def gram_workspace_bytes(dims, n_tokens: int, levels: int) -> int:
    d, c, K, L_ = dims
    return int(lib().rqae_hip_gram_workspace_bytes(d, c, K, L_, n_tokens, levels))


def encode_gram(gpacked, dims, codebook_shared: bool, x2d: torch.Tensor, levels: int, want_recon: bool = True,
                workspace: torch.Tensor | None = None, options: GramOptions | None = None):
    """x2d [N,d] {f32,bf16,f16} -> (q_out [N,d] f32 | None, indices [N,levels] i64) on the matrix-core arm."""
    options = _gram_default_options(gpacked, options)
    d, c, K, L_ = dims
    _require_cuda(x2d, "x")
    if x2d.dtype not in _X_DTYPES:
        raise RuntimeError(f"encode_gram: unsupported input dtype {x2d.dtype}")
    if x2d.dim() != 2 or x2d.shape[1] != d or not x2d.is_contiguous():
        raise RuntimeError("encode_gram: x must be contiguous [N, d]")
    n = x2d.shape[0]
    q = torch.empty((n, d), dtype=torch.float32, device=x2d.device) if want_recon else None
    idx = torch.empty((n, levels), dtype=torch.int64, device=x2d.device)
    if n == 0:
        return q, idx
    need = gram_workspace_bytes(dims, n, levels)
    if workspace is None or workspace.numel() * workspace.element_size() < need or workspace.device != x2d.device:
        workspace = torch.empty(need, dtype=torch.uint8, device=x2d.device)
    with torch.cuda.device(x2d.device):
        rc = lib().rqae_hip_encode_gram(gpacked.data_ptr(), d, c, K, L_, int(bool(codebook_shared)), x2d.data_ptr(),
                                        _X_DTYPES[x2d.dtype], n, levels, q.data_ptr() if want_recon else None,
                                        idx.data_ptr(), workspace.data_ptr(), ctypes.byref(options) if options is not None else None,
                                        _stream(x2d.device))
    _check(rc, "rqae_hip_encode_gram")
    return q, idx


def encode_gram_gemma_supported(dims, hs_dtype) -> bool:
    d, c, K, L_ = dims
    return hs_dtype in _X_DTYPES and bool(lib().rqae_hip_encode_gram_gemma_supported(d, c, K, L_, 1, _X_DTYPES[hs_dtype]))


def encode_gram_gemma(gpacked_scaled, dims, hs: torch.Tensor, eps: float, levels: int, workspace: torch.Tensor | None = None,
                      options: GramOptions | None = None):
    """hs [B,S,d] {bf16,f16} (the hooked layer's output) -> (q_out [B,S,d] f32, indices [B,S,levels] i64, inv_rms [B,S] f32):
    Gemma RMSNorm + RQAE.forward on the matrix-core arm (rqae/model.py:279-281); ``gpacked_scaled`` = gram_pack(...,
    in_scale=1 + norm.weight)."""
    _require_cuda(hs, "hidden states")
    if hs.dtype not in (torch.bfloat16, torch.float16) or not hs.is_contiguous() or hs.dim() != 3:
        raise RuntimeError("encode_gram_gemma: hidden states must be contiguous [B,S,d] bf16/f16")
    options = _gram_default_options(gpacked_scaled, options)
    d, c, K, L_ = dims
    B, S, _ = hs.shape
    n = B * S
    q = torch.empty((B, S, d), dtype=torch.float32, device=hs.device)
    idx = torch.empty((B, S, levels), dtype=torch.int64, device=hs.device)
    rs = torch.empty((B, S), dtype=torch.float32, device=hs.device)
    if n:
        need = gram_workspace_bytes(dims, n, levels)
        if workspace is None or workspace.numel() * workspace.element_size() < need or workspace.device != hs.device:
            workspace = torch.empty(need, dtype=torch.uint8, device=hs.device)
        with torch.cuda.device(hs.device):
            rc = lib().rqae_hip_encode_gram_gemma(gpacked_scaled.data_ptr(), d, c, K, L_, 1, hs.data_ptr(), _X_DTYPES[hs.dtype],
                                                  float(eps), n, levels, q.data_ptr(), idx.data_ptr(), rs.data_ptr(),
                                                  workspace.data_ptr(), ctypes.byref(options) if options is not None else None,
                                                  _stream(hs.device))
        _check(rc, "rqae_hip_encode_gram_gemma")
    return q, idx, rs


def encode_gram_gemma_patch_supported(dims, hs_dtype, levels: int, lattice_vouched: bool) -> bool:
    """include/rqae_hip.h: rqae_hip_encode_gram_gemma_patch_supported (the in-place hook is gated on the conditions the
    C side checks, not on the depth alone)."""
    d, c, K, L_ = dims
    return hs_dtype in _X_DTYPES and bool(lib().rqae_hip_encode_gram_gemma_patch_supported(d, c, K, L_, 1, int(levels), _X_DTYPES[hs_dtype],
                                                                                          int(bool(lattice_vouched))))


def encode_gram_gemma_patch(gpacked_scaled, dims, hs: torch.Tensor, inv_1pw: torch.Tensor, eps: float, levels: int, skip_bos: bool,
                            workspace: torch.Tensor | None = None, options: GramOptions | None = None):
    """RQAE.hook for a Gemma-2 layer output end to end, in place (include/rqae_hip.h: rqae_hip_encode_gram_gemma_patch): hs [B,S,d]
    {bf16,f16} is overwritten with denorm(forward(norm(hs))) (BOS rows kept when skip_bos); returns (indices [B,S,levels] i64,
    inv_rms [B,S] f32).  ``inv_1pw``: 1 / (1 + norm.weight), fp32, padded to a whole KiB (``gemma_inv_1pw``)."""
    _require_cuda(hs, "hidden states")
    if hs.dtype not in (torch.bfloat16, torch.float16) or not hs.is_contiguous() or hs.dim() != 3:
        raise RuntimeError("encode_gram_gemma_patch: hidden states must be contiguous [B,S,d] bf16/f16")
    options = _gram_default_options(gpacked_scaled, options)
    d, c, K, L_ = dims
    B, S, _ = hs.shape
    n = B * S
    idx = torch.empty((B, S, levels), dtype=torch.int64, device=hs.device)
    rs = torch.empty((B, S), dtype=torch.float32, device=hs.device)
    if n:
        need = gram_workspace_bytes(dims, n, levels)
        if workspace is None or workspace.numel() * workspace.element_size() < need or workspace.device != hs.device:
            workspace = torch.empty(need, dtype=torch.uint8, device=hs.device)
        with torch.cuda.device(hs.device):
            rc = lib().rqae_hip_encode_gram_gemma_patch(gpacked_scaled.data_ptr(), d, c, K, L_, 1, hs.data_ptr(), _X_DTYPES[hs.dtype],
                                                        float(eps), inv_1pw.data_ptr(), S, int(bool(skip_bos)), n, levels,
                                                        idx.data_ptr(), rs.data_ptr(), workspace.data_ptr(),
                                                        ctypes.byref(options) if options is not None else None, _stream(hs.device))
        _check(rc, "rqae_hip_encode_gram_gemma_patch")
    return idx, rs


def gemma_inv_1pw(norm_weight: torch.Tensor) -> torch.Tensor:
    """1 / (1 + w) as fp32 on w's device, zero-padded to a whole KiB (the reconstruction kernel stages it by 1-KiB pieces)."""
    w = norm_weight.detach().float()
    d = w.numel()
    out = torch.zeros(((d * 4 + 1023) // 1024) * 256, dtype=torch.float32, device=w.device)
    out[:d] = 1.0 / (1.0 + w)
    return out


def decode_gram_supported(dims, codebook_shared: bool, levels: int) -> bool:
    d, c, K, L_ = dims
    return bool(lib().rqae_hip_decode_gram_supported(d, c, K, L_, int(bool(codebook_shared)), levels))


def decode_gram(gpacked, dims, codebook_shared: bool, idx2d: torch.Tensor) -> torch.Tensor:
    """idx2d [N, levels] {i64,i32,u16} -> out [N,d] f32: every level selected, on the matrix-core arm (include/rqae_hip.h)."""
    d, c, K, L_ = dims
    _require_cuda(idx2d, "indices")
    if idx2d.dtype not in _IDX_DTYPES or idx2d.dim() != 2 or not idx2d.is_contiguous():
        raise RuntimeError("decode_gram: indices must be contiguous int64 / int32 / uint16 [N, levels]")
    n, levels = idx2d.shape
    out = torch.empty((n, d), dtype=torch.float32, device=idx2d.device)
    if n:
        with torch.cuda.device(idx2d.device):
            rc = lib().rqae_hip_decode_gram(gpacked.data_ptr(), d, c, K, L_, int(bool(codebook_shared)), idx2d.data_ptr(),
                                            _IDX_DTYPES[idx2d.dtype], n, levels, out.data_ptr(), _status_ptr(idx2d.device),
                                            _stream(idx2d.device))
        _check(rc, "rqae_hip_decode_gram")
        _codes_checked(idx2d.device, f"decode: index out of range for the codebook of {K} rows")
    return out


def encode_gram_phases(gpacked, dims, x2d: torch.Tensor, levels: int, want_recon: bool = True, options: GramOptions | None = None):
    """Measurement only: one encode_gram call timed per kernel -> [ms projection, ms level chain, ms reconstruction]."""
    options = _gram_default_options(gpacked, options)
    d, c, K, L_ = dims
    n = x2d.shape[0]
    q = torch.empty((n, d), dtype=torch.float32, device=x2d.device) if want_recon else None
    idx = torch.empty((n, levels), dtype=torch.int64, device=x2d.device)
    ws = torch.empty(gram_workspace_bytes(dims, n, levels), dtype=torch.uint8, device=x2d.device)
    ms = (ctypes.c_float * 3)()
    with torch.cuda.device(x2d.device):
        rc = lib().rqae_hip_encode_gram_phases(gpacked.data_ptr(), d, c, K, L_, 1, x2d.data_ptr(), _X_DTYPES[x2d.dtype], n,
                                               levels, q.data_ptr() if want_recon else None, idx.data_ptr(), ws.data_ptr(),
                                               ctypes.byref(options) if options is not None else None, _stream(x2d.device), ms)
    _check(rc, "rqae_hip_encode_gram_phases")
    return [float(v) for v in ms]


def quantize(p2d: torch.Tensor, codebook_l: torch.Tensor):
    """p2d [N,c] f32, codebook_l [K,c] f32 -> (cv [N,c] f32, idx [N] i64)."""
    _require_cuda(p2d, "x")
    _require_cuda(codebook_l, "codebook")
    n, c = p2d.shape
    K = codebook_l.shape[0]
    cv = torch.empty((n, c), dtype=torch.float32, device=p2d.device)
    idx = torch.empty((n,), dtype=torch.int64, device=p2d.device)
    if n == 0:
        return cv, idx
    with torch.cuda.device(p2d.device):
        rc = lib().rqae_hip_quantize(p2d.data_ptr(), n, c, K, codebook_l.data_ptr(), cv.data_ptr(),
                                     idx.data_ptr(), _stream(p2d.device))
    _check(rc, "rqae_hip_quantize")
    return cv, idx


def _mask_arg(level_mask):
    if level_mask is None:
        return None, None
    buf = (ctypes.c_uint8 * len(level_mask))(*[1 if m else 0 for m in level_mask])
    return buf, ctypes.cast(buf, ctypes.c_void_p)


# ---- bad code indices: a status word instead of a device assertion (include/rqae_hip.h "Status word") -----------------------
# The kernels that gather by caller-supplied codes never fault on a bad one: they read row 0 (or LDS) and OR a bit into a device
# int32.  The reference raises a recoverable IndexError for such input (rqae/model.py:234, rqae/feature.py:120); so does this
# wrapper, from the word.  RQAE_HIP_CHECK_CODES (or set_code_check()) selects when the word is read:
#   "sync"  (default) after the call, with one stream synchronisation: IndexError is raised BY the call, as in the reference;
#   "lazy"  the word is copied to pinned host memory behind the call (no synchronisation); a set bit raises IndexError at the
#           next rqae_amd call that takes codes or at check_status() -- for loops that queue many calls;
#   "off"   never read (the kernels still clamp).
_CHECK_MODE = os.environ.get("RQAE_HIP_CHECK_CODES", "sync").lower()
if _CHECK_MODE in ("1", "true", "on", ""):
    _CHECK_MODE = "sync"
if _CHECK_MODE in ("0", "false"):
    _CHECK_MODE = "off"
_status_words = {}     # device index -> (device int32[1], pinned host int32[1], [message of the last flagged-able call])


def set_code_check(mode: str):
    """"sync" | "lazy" | "off" (see above)."""
    global _CHECK_MODE
    if mode not in ("sync", "lazy", "off"):
        raise ValueError(f"code check mode {mode!r}")
    _CHECK_MODE = mode


def _status_entry(device):
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ent = _status_words.get(key)
    if ent is None:
        with torch.inference_mode(False):      # plain tensors: they are zeroed in place from either mode later on
            ent = (torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", key)), torch.zeros(1, dtype=torch.int32).pin_memory(), [""])
        _status_words[key] = ent
    return ent


def _status_ptr(device):
    """Device pointer of this device's status word for a call that takes codes; in lazy mode a bit left by an EARLIER call
    surfaces here, before the new call is queued."""
    if _CHECK_MODE == "off":
        return None
    ent = _status_entry(device)
    if _CHECK_MODE == "lazy" and int(ent[1][0]) != 0 and not torch.cuda.is_current_stream_capturing():
        _raise_flagged(ent)
    return ent[0].data_ptr()


def _raise_flagged(ent):
    msg = ent[2][0] or "code index out of range"
    ent[0].zero_()
    ent[1].zero_()
    raise IndexError(f"rqae_amd: {msg}")


def _codes_checked(device, message: str):
    """Call after queuing a kernel that was handed the status word."""
    if _CHECK_MODE == "off":
        return
    ent = _status_entry(device)
    ent[2][0] = message
    if torch.cuda.is_current_stream_capturing():
        return                             # inside a HIP graph capture nothing may synchronise: a replay's bit shows at check_status()
    if _CHECK_MODE == "sync":
        if int(ent[0].item()) != 0:      # synchronises the stream
            _raise_flagged(ent)
    else:
        ent[1].copy_(ent[0], non_blocking=True)


def check_status(device=None):
    """Synchronise and raise IndexError if any call since the last check met a code outside its table (lazy mode's barrier)."""
    ents = list(_status_words.values()) if device is None else [_status_entry(device)]
    for ent in ents:
        if int(ent[0].item()) != 0:
            _raise_flagged(ent)


def codes_in_range(idx: torch.Tensor, K: int, what: str, allow_negative: bool = False):
    """Host-side range check of a SMALL code tensor (a query, a feature centre): raises IndexError as the reference's indexing
    would.  One tiny reduction and a synchronisation -- not for dataset-sized inputs (those are checked by the kernels)."""
    if idx.numel() == 0:
        return
    v = idx.to(torch.int64) if idx.dtype != torch.uint16 else (idx.view(torch.int16).to(torch.int64) & 0xFFFF)
    lo = -K if allow_negative else 0
    if bool(((v < lo) | (v >= K)).any()):
        raise IndexError(f"rqae_amd: {what}: code outside {lo}..{K - 1} (index out of range for a table of {K} rows)")


def decode(packed, dims, idx2d: torch.Tensor, level_mask=None) -> torch.Tensor:
    """idx2d [N,NQ] {i64,i32,u16} -> out [N,d] f32 (sum over unmasked levels)."""
    d, c, K, L_ = dims
    _require_cuda(idx2d, "indices")
    if idx2d.dtype not in _IDX_DTYPES:
        raise RuntimeError(f"decode: unsupported index dtype {idx2d.dtype}")
    n, nq = idx2d.shape
    out = torch.empty((n, d), dtype=torch.float32, device=idx2d.device)
    if n == 0:
        return out
    keep, maskp = _mask_arg(level_mask)
    with torch.cuda.device(idx2d.device):
        rc = lib().rqae_hip_decode(packed.data_ptr(), d, c, K, L_, idx2d.data_ptr(), _IDX_DTYPES[idx2d.dtype],
                                   n, nq, maskp, out.data_ptr(), _status_ptr(idx2d.device), _stream(idx2d.device))
    _check(rc, "rqae_hip_decode")
    _codes_checked(idx2d.device, f"decode: index out of range for the codebook of {K} rows")
    return out


def decode_values(packed, dims, cv3d: torch.Tensor, level_mask=None) -> torch.Tensor:
    """cv3d [N,NQ,c] f32 -> out [N,d] f32."""
    d, c, K, L_ = dims
    _require_cuda(cv3d, "codebook_values")
    n, nq, cc = cv3d.shape
    if cc != c or cv3d.dtype != torch.float32 or not cv3d.is_contiguous():
        raise RuntimeError("decode_values: codebook_values must be contiguous float32 [N, NQ, c]")
    out = torch.empty((n, d), dtype=torch.float32, device=cv3d.device)
    if n == 0:
        return out
    keep, maskp = _mask_arg(level_mask)
    with torch.cuda.device(cv3d.device):
        rc = lib().rqae_hip_decode_values(packed.data_ptr(), d, c, K, L_, cv3d.data_ptr(), n, nq, maskp,
                                          out.data_ptr(), _stream(cv3d.device))
    _check(rc, "rqae_hip_decode_values")
    return out


def encode_gemma_supported(dims) -> bool:
    d, c, K, L_ = dims
    return bool(lib().rqae_hip_encode_gemma_supported(d, c, K, L_))


def encode_gemma(packed, dims, hs: torch.Tensor, norm_weight: torch.Tensor, eps: float, levels: int):
    """hs [B,S,d] {f16,bf16,f32} (the hooked layer's output) -> (q_out [B,S,d] f32, indices [B,S,levels] i64,
    inv_rms [B,S] f32): Gemma RMSNorm + RQAE.forward in one launch (rqae/model.py:279-281)."""
    _require_cuda(hs, "hidden states")
    if hs.dtype not in _X_DTYPES or not hs.is_contiguous() or hs.dim() != 3:
        raise RuntimeError("encode_gemma: hidden states must be contiguous [B,S,d] f16/bf16/f32")
    d, c, K, L_ = dims
    B, S, _ = hs.shape
    w = norm_weight.detach().float().contiguous()
    q = torch.empty((B, S, d), dtype=torch.float32, device=hs.device)
    idx = torch.empty((B, S, levels), dtype=torch.int64, device=hs.device)
    rs = torch.empty((B, S), dtype=torch.float32, device=hs.device)
    if B * S:
        with torch.cuda.device(hs.device):
            rc = lib().rqae_hip_encode_gemma(packed.data_ptr(), d, c, K, L_, hs.data_ptr(), _X_DTYPES[hs.dtype],
                                             w.data_ptr(), float(eps), B * S, levels, q.data_ptr(), idx.data_ptr(),
                                             rs.data_ptr(), _stream(hs.device))
        _check(rc, "rqae_hip_encode_gemma")
    return q, idx, rs


def gemma_norm(hs: torch.Tensor, norm_weight: torch.Tensor, eps: float):
    """hs [..., d] {f16,bf16,f32} -> (normed [..., d] f32, inv_rms [...] f32); rqae/llm.py:65-66."""
    _require_cuda(hs, "hidden states")
    if hs.dtype not in _X_DTYPES or not hs.is_contiguous():
        raise RuntimeError("gemma_norm: hidden states must be contiguous f16/bf16/f32")
    d = hs.shape[-1]
    n = hs.numel() // d
    w = norm_weight.detach().float().contiguous()
    normed = torch.empty(hs.shape, dtype=torch.float32, device=hs.device)
    rs = torch.empty(hs.shape[:-1], dtype=torch.float32, device=hs.device)
    if n:
        with torch.cuda.device(hs.device):
            rc = lib().rqae_hip_gemma_norm(hs.data_ptr(), _X_DTYPES[hs.dtype], w.data_ptr(), float(eps), n, d,
                                           normed.data_ptr(), rs.data_ptr(), _stream(hs.device))
        _check(rc, "rqae_hip_gemma_norm")
    return normed, rs


def gemma_denorm_patch(q_out: torch.Tensor, norm_weight: torch.Tensor, inv_rms: torch.Tensor, hs_inout: torch.Tensor,
                       skip_bos: bool, want_new: bool):
    """Writes denorm(q_out) back into hs_inout [B,S,d] in place (BOS rows kept when skip_bos);
    returns the fp32 `new` tensor when want_new.  rqae/llm.py:68-73, rqae/model.py:284-289."""
    _require_cuda(q_out, "q_out")
    B, S, d = hs_inout.shape
    if hs_inout.dtype not in _X_DTYPES or not hs_inout.is_contiguous() or not q_out.is_contiguous():
        raise RuntimeError("gemma_denorm_patch: tensors must be contiguous; hidden states f16/bf16/f32")
    w = norm_weight.detach().float().contiguous()
    new = torch.empty((B, S, d), dtype=torch.float32, device=q_out.device) if want_new else None
    if B * S:
        with torch.cuda.device(q_out.device):
            rc = lib().rqae_hip_gemma_denorm_patch(q_out.data_ptr(), w.data_ptr(), inv_rms.data_ptr(), B * S, d, S,
                                                   int(bool(skip_bos)), hs_inout.data_ptr(), _X_DTYPES[hs_inout.dtype],
                                                   new.data_ptr() if want_new else None, _stream(q_out.device))
        _check(rc, "rqae_hip_gemma_denorm_patch")
    return new


def intensity_prepare(codebook_sims: torch.Tensor, center: torch.Tensor, layer_weights: torch.Tensor, M: int):
    """(table [M,K] f16 = codebook_sims[center[m]], wcum [M] f16 = layer_weights[:M].cumsum() in torch-CPU order);
    rqae/feature.py:119-126.  Inputs on the device: codebook_sims [K,K] f16, center int32, layer_weights f16."""
    _require_cuda(codebook_sims, "codebook_sims")
    K = codebook_sims.shape[0]
    if codebook_sims.dtype != torch.float16 or codebook_sims.shape != (K, K) or not codebook_sims.is_contiguous():
        raise RuntimeError("intensity_prepare: codebook_sims must be a contiguous [K, K] float16 tensor")
    dev = codebook_sims.device
    c = center[:M].to(device=dev, dtype=torch.int32).contiguous()
    w = layer_weights[:M].to(device=dev, dtype=torch.float16).contiguous()
    if c.numel() != M or w.numel() != M:
        raise RuntimeError("intensity_prepare: center / layer_weights shorter than max(layers) + 1")
    codes_in_range(c, K, "intensity: feature centre")      # rows of codebook_sims (rqae/feature.py:120 indexes by them)
    n16 = (M * K * 2 + 15) // 16
    table = torch.empty(n16 * 8, dtype=torch.float16, device=dev)   # whole 16-byte pieces
    wcum = torch.empty(M, dtype=torch.float16, device=dev)
    with torch.cuda.device(dev):
        rc = lib().rqae_hip_intensity_prepare(codebook_sims.data_ptr(), c.data_ptr(), w.data_ptr(), M, K,
                                              table.data_ptr(), wcum.data_ptr(), _stream(dev))
    _check(rc, "rqae_hip_intensity_prepare")
    return table, w, wcum


def intensity(table, w, wcum, M: int, K: int, idx2d: torch.Tensor, layers_dev: torch.Tensor) -> torch.Tensor:
    """idx2d [n, nq] int64/int32/uint16 on the device -> [n, n_layers] f16; rqae/feature.py:119-128."""
    _require_cuda(idx2d, "token_indices")
    if idx2d.dtype not in _IDX_DTYPES or not idx2d.is_contiguous() or idx2d.dim() != 2:
        raise RuntimeError("intensity: token indices must be a contiguous [n, nq] int64/int32/uint16 tensor")
    n, nq = idx2d.shape
    nl = layers_dev.numel()
    out = torch.empty((n, nl), dtype=torch.float16, device=idx2d.device)
    if n:
        with torch.cuda.device(idx2d.device):
            rc = lib().rqae_hip_intensity(table.data_ptr(), w.data_ptr(), wcum.data_ptr(), M, K, idx2d.data_ptr(),
                                          _IDX_DTYPES[idx2d.dtype], n, nq, layers_dev.data_ptr(), nl, out.data_ptr(),
                                          _status_ptr(idx2d.device), _stream(idx2d.device))
        _check(rc, "rqae_hip_intensity")
        _codes_checked(idx2d.device, f"intensity: token index out of range for the similarity table of {K} rows")
    return out


def _f32_dev(t: torch.Tensor, what: str) -> torch.Tensor:
    _require_cuda(t, what)
    return t.detach().to(torch.float32).contiguous()


def codebook_sims(codebook0: torch.Tensor) -> torch.Tensor:
    """[K, c] fp32 on the device -> [K, K] f16 cosine table; rqae/model.py:133-143 (the reference's CPU bits)."""
    cb = _f32_dev(codebook0, "codebook")
    K, c = cb.shape
    out = torch.empty((K, K), dtype=torch.float16, device=cb.device)
    with torch.cuda.device(cb.device):
        rc = lib().rqae_hip_codebook_sims(cb.data_ptr(), K, c, out.data_ptr(), _stream(cb.device))
    _check(rc, "rqae_hip_codebook_sims")
    return out


def subfeatures(codebook: torch.Tensor, w_out: torch.Tensor, b_out: torch.Tensor) -> torch.Tensor:
    """codebook [L, K, c], w_out [L, d, c], b_out [L, d] on the device -> [L, K, d] fp32; rqae/model.py:145-156."""
    cb, wo, bo = _f32_dev(codebook, "codebook"), _f32_dev(w_out, "w_out"), _f32_dev(b_out, "b_out")
    L, K, c = cb.shape
    d = wo.shape[1]
    if wo.shape != (L, d, c) or bo.shape != (L, d):
        raise RuntimeError("subfeatures: w_out must be [L, d, c] and b_out [L, d]")
    out = torch.empty((L, K, d), dtype=torch.float32, device=cb.device)
    with torch.cuda.device(cb.device):
        rc = lib().rqae_hip_subfeatures(cb.data_ptr(), wo.data_ptr(), bo.data_ptr(), L, K, c, d, out.data_ptr(),
                                        _stream(cb.device))
    _check(rc, "rqae_hip_subfeatures")
    return out


def subfeature_sims(codebook: torch.Tensor, w_out: torch.Tensor, b_out: torch.Tensor, want_sims: bool = True,
                    scale_by_layer_norms: bool = False):
    """-> (sims [L, K, K] f16 or None, layer_norms [L] fp32); rqae/model.py:158-178, from the Gram matrices in fp64.
    scale_by_layer_norms: the table as the demo server keeps it (`sims *= layer_norms`, demo/server/server.py:108)."""
    cb, wo, bo = _f32_dev(codebook, "codebook"), _f32_dev(w_out, "w_out"), _f32_dev(b_out, "b_out")
    L, K, c = cb.shape
    d = wo.shape[1]
    if wo.shape != (L, d, c) or bo.shape != (L, d):
        raise RuntimeError("subfeature_sims: w_out must be [L, d, c] and b_out [L, d]")
    nbytes = lib().rqae_hip_tables_scratch_bytes(L, K, c)
    if nbytes == 0:
        _check(1, "rqae_hip_tables_scratch_bytes")
    scratch = torch.empty(nbytes // 8, dtype=torch.float64, device=cb.device)
    sims = torch.empty((L, K, K), dtype=torch.float16, device=cb.device) if want_sims else None
    norms = torch.empty(L, dtype=torch.float32, device=cb.device)
    with torch.cuda.device(cb.device):
        rc = lib().rqae_hip_subfeature_sims(cb.data_ptr(), wo.data_ptr(), bo.data_ptr(), L, K, c, d, scratch.data_ptr(),
                                            sims.data_ptr() if want_sims else None, norms.data_ptr(),
                                            1 if scale_by_layer_norms else 0, _stream(cb.device))
    _check(rc, "rqae_hip_subfeature_sims")
    return sims, norms


def kernel_info(d: int, c: int, K: int, L: int) -> str:
    """Which kernel variant encode() would launch for this shape (for bench/logs)."""
    buf = ctypes.create_string_buffer(256)
    rc = lib().rqae_hip_kernel_info(d, c, K, L, buf, 256)
    _check(rc, "rqae_hip_kernel_info")
    return buf.value.decode()


# ---- similarity search over resident codes (csrc/search.hip; include/rqae_hip.h "Similarity search") ---------------

def search_supported(K: int, Q: int) -> bool:
    return bool(lib().rqae_hip_search_supported(K, Q))


def search_query_table(sims16: torch.Tensor, query_codes: torch.Tensor, l0: int, l1: int, out: torch.Tensor | None = None):
    """sims16 [L,K,K] fp16, query_codes [Q, >= l1] int32 -> the range's table [l1-l0, KP, 128] fp16 (as a flat byte tensor)."""
    _require_cuda(sims16, "sims")
    _require_cuda(query_codes, "query codes")
    L_, K, K2 = sims16.shape
    if sims16.dtype != torch.float16 or not sims16.is_contiguous() or K != K2:
        raise RuntimeError("search_query_table: sims must be contiguous float16 [L, K, K]")
    if query_codes.dtype != torch.int32 or query_codes.dim() != 2 or query_codes.stride(1) != 1:
        raise RuntimeError("search_query_table: query codes must be int32 [Q, levels] with unit level stride")
    Q = query_codes.shape[0]
    need = int(lib().rqae_hip_search_table_bytes(K, l1 - l0))
    if out is None or out.numel() < need:
        out = torch.empty(max(need, 16), dtype=torch.uint8, device=sims16.device)
    with torch.cuda.device(sims16.device):
        rc = lib().rqae_hip_search_query_table(sims16.data_ptr(), L_, K, query_codes.data_ptr(), Q, query_codes.stride(0), l0, l1,
                                               out.data_ptr(), _stream(sims16.device))
    _check(rc, "rqae_hip_search_query_table")
    return out


def search_accumulate(codes: torch.Tensor, table: torch.Tensor, K: int, Q: int, l0: int, l1: int, acc: torch.Tensor, first: bool):
    """codes [N,S,L] int32/uint16 (resident), table from search_query_table, acc [N,S,128] fp16 updated in place."""
    _require_cuda(codes, "codes")
    if codes.dtype not in (torch.int32, torch.uint16) or not codes.is_contiguous() or codes.dim() != 3:
        raise RuntimeError("search_accumulate: codes must be contiguous int32 or uint16 [N, S, L]")
    N, S, L_ = codes.shape
    if acc.dtype != torch.float16 or tuple(acc.shape) != (N, S, 128) or not acc.is_contiguous() or acc.device != codes.device:
        raise RuntimeError("search_accumulate: acc must be contiguous float16 [N, S, 128] on the codes' device")
    with torch.cuda.device(codes.device):
        rc = lib().rqae_hip_search_accumulate(codes.data_ptr(), _IDX_DTYPES[codes.dtype], N * S, L_, table.data_ptr(), K, Q, l0, l1,
                                              acc.data_ptr(), int(bool(first)), _status_ptr(codes.device), _stream(codes.device))
    _check(rc, "rqae_hip_search_accumulate")
    _codes_checked(codes.device, f"search_accumulate: dataset code out of range for the table of {K} rows")


def search_rank(acc: torch.Tensor, Q: int, workspace: torch.Tensor | None = None):
    """acc [N,S,128] fp16 -> (max values [Q,N] fp16 descending, sequence numbers [Q,N] int32, workspace)."""
    _require_cuda(acc, "acc")
    N, S, _ = acc.shape
    need = int(lib().rqae_hip_search_rank_workspace_bytes(N, Q))
    if need == 0:
        raise RuntimeError(f"search_rank: unsupported shape N={N} Q={Q}")
    if workspace is None or workspace.numel() < need or workspace.device != acc.device:
        workspace = torch.empty(need, dtype=torch.uint8, device=acc.device)
    mx = torch.empty((Q, N), dtype=torch.float16, device=acc.device)
    order = torch.empty((Q, N), dtype=torch.int32, device=acc.device)
    with torch.cuda.device(acc.device):
        rc = lib().rqae_hip_search_rank(acc.data_ptr(), N, S, Q, mx.data_ptr(), order.data_ptr(), workspace.data_ptr(), workspace.numel(),
                                        _stream(acc.device))
    _check(rc, "rqae_hip_search_rank")
    return mx, order, workspace


def search_gather(acc: torch.Tensor, Q: int, examples: torch.Tensor) -> torch.Tensor:
    """acc [N,S,128] fp16, examples [Q,E] int32 -> [Q,E,S] fp16."""
    _require_cuda(acc, "acc")
    N, S, _ = acc.shape
    if examples.dtype != torch.int32 or examples.dim() != 2 or examples.shape[0] != Q or not examples.is_contiguous():
        raise RuntimeError("search_gather: examples must be contiguous int32 [Q, E]")
    E = examples.shape[1]
    out = torch.empty((Q, E, S), dtype=torch.float16, device=acc.device)
    if E:
        with torch.cuda.device(acc.device):
            rc = lib().rqae_hip_search_gather(acc.data_ptr(), N, S, Q, examples.data_ptr(), E, out.data_ptr(),
                                              _status_ptr(acc.device), _stream(acc.device))
        _check(rc, "rqae_hip_search_gather")
        _codes_checked(acc.device, f"search_gather: example outside the dataset of {N} sequences")
    return out
