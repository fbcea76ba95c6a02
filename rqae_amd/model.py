"""Host-side mirror of the reference ``rqae.model.RQAE`` for the MI355X path.

The class keeps the reference's public surface so its callers
(scripts/1_create_activations.py:144,176; scripts/3_make_rqae_features.py:184;
rqae/feature.py:95-137; demo/server/server.py:101-112) run unchanged:

* constructor arguments and state-dict keys      -- rqae/model.py:18-73
* ``forward`` / ``quantize`` / ``decode`` / ``hook`` -- rqae/model.py:187-291
* cached metadata properties                     -- rqae/model.py:133-178

What differs is where the arithmetic happens: the level loop of ``forward``,
``quantize`` and ``decode`` are single launches of the gfx950 kernels in
rqae_amd/csrc behind the C ABI of include/rqae_hip.h.  The hot path is
inference-only and device-only; it raises instead of falling back to CPU.
"""
from __future__ import annotations

import json
import math
import operator
import os
from itertools import product

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native

_FSQ_METHODS = ("fsq", "round_fsq")
_TENSOR_VERSION = operator.attrgetter("_version")


def _fsq_grid(codebook_dim: int, codebook_size: int, unit_rows: bool) -> torch.Tensor:
    """The (K, c) FSQ lattice in float64 (K = codebook_size ** codebook_dim).

    Same numbers as rqae/model.py:64-71: the Cartesian power of
    linspace(-1, 1, codebook_size); with ``unit_rows`` each row is scaled to unit
    L2 norm and the all-zero row (odd codebook_size) stays zero.
    """
    axis = np.linspace(-1, 1, codebook_size)
    lattice = np.array(list(product(axis, repeat=codebook_dim)))
    if unit_rows:
        length = np.linalg.norm(lattice, axis=-1, keepdims=True)
        length = np.where(length == 0, 1.0, length)
        lattice = np.divide(lattice, length, where=length != 0)
    return torch.from_numpy(lattice)


class RQAE(nn.Module):
    """Residual-quantisation autoencoder; see module docstring for the mapping."""

    GRAM_MIN_TOKENS = 16384   # arm="auto": below this the exact arm's single launch wins
    GRAM_MIN_TOKENS_DEEP = 8192    # ... for calls of more than 64 levels (the blocked path: ~100 launches, 5.6 ms of them at any batch; measured at depth
                                   # 1024: 4096 tokens exact 4.8 / gram 6.2 ms, 8192: 8.9 / 6.9, 16384: 17.9 / 8.0 -- DESIGN.md 13.4)

    PRETRAINED = {
        "google/gemma-2-2b": "harish-kamath/rqae/gemma-2-2b",
        "rqae-rqae-round_fsq-cbd4-cbs5-nq1024": "harish-kamath/rqae/gemma-2-2b",
    }

    def __init__(self, dim: int = 2304, codebook_dim: int = 4, codebook_size: int = 5,
                 num_quantizers: int = 1024, quantization_method: str = "round_fsq",
                 name: str = "", **kwargs):
        super().__init__()
        # layers.{l}.0 = lin_in (d -> c), layers.{l}.1 = lin_out (c -> d)
        self.layers = nn.ModuleList(
            nn.ModuleList((nn.Linear(dim, codebook_dim), nn.Linear(codebook_dim, dim)))
            for _ in range(num_quantizers))
        is_fsq = quantization_method in _FSQ_METHODS
        rows = codebook_size ** codebook_dim if is_fsq else codebook_size
        self.codebook = nn.Parameter(torch.randn(num_quantizers, rows, codebook_dim))
        self.register_buffer("codebook_counts", torch.zeros(num_quantizers, rows), persistent=True)

        self.quantization_method = quantization_method
        self.num_quantizers = num_quantizers
        self.codebook_dim = codebook_dim
        self.codebook_size = codebook_size
        self.name = name
        self._dim = dim
        self._packed = None
        self._packed_key = None
        self._gram = None            # (key, packed block, shared flag) of the matrix-core arm
        self._gram_ws = None
        self.gram_options = None     # _native.gram_options(...): kernel selection of this module's matrix-core calls (None = defaults)
        # which encode arm forward()/encode() take (not a reference argument; rides in **kwargs so the reference's
        # constructor signature stays as it is):
        #   "exact" (default) -- rqae_hip_encode: the canonical fp32 order, bit-equal to the CPU oracle
        #   "gram"            -- rqae_hip_encode_gram: all-levels projection + c-space level chain + reconstruction on the
        #                        bf16 matrix cores (bf16x3 operand split), SURVEY 8(c) Tier-2 parity: same codes wherever
        #                        the top-1/top-2 margin is >= 1e-5, reconstruction within 1e-5 relative
        #   "auto"            -- gram for batches of >= GRAM_MIN_TOKENS tokens on shapes it serves, else exact
        self.arm = kwargs.pop("arm", os.environ.get("RQAE_HIP_ARM", "exact"))
        if self.arm not in ("exact", "gram", "auto"):
            raise ValueError(f"unknown encode arm {self.arm!r}")

        self.normalize_codebooks()
        if is_fsq:
            grid = _fsq_grid(codebook_dim, codebook_size, quantization_method == "round_fsq")
            self.codebook.data.copy_(grid)  # float64 -> float32, broadcast over levels
            self.codebook.requires_grad = False

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, model_name: str):
        """HF-hub loader with the reference's repo layout (rqae/model.py:75-99)."""
        from huggingface_hub import hf_hub_download
        from safetensors import safe_open

        model_name = cls.PRETRAINED.get(model_name, model_name)
        user, repo, *sub = model_name.split("/")
        folder = "/".join(sub)
        weights_file = hf_hub_download(f"{user}/{repo}", os.path.join(folder, "model.safetensors"))
        config_file = hf_hub_download(f"{user}/{repo}", os.path.join(folder, "config.json"))
        with open(config_file, "r") as fh:
            params = json.load(fh)
        name = (f"rqae-{repo}-{params['quantization_method']}-cbd{params['codebook_dim']}"
                f"-cbs{params['codebook_size']}-nq{params['num_quantizers']}")
        model = cls(**params, name=name)
        if weights_file.endswith(".safetensors"):
            with safe_open(weights_file, framework="pt") as fh:
                state = {k: fh.get_tensor(k) for k in fh.keys()}
            model.load_state_dict(state, strict=True)
        elif weights_file.endswith(".pt"):
            model.load_state_dict(torch.load(weights_file, weights_only=True))
        else:
            raise ValueError(f"Unknown file extension for {weights_file}")
        return model

    # ------------------------------------------------------- packed-weight cache
    def _apply(self, fn, *args, **kwargs):
        self._packed = None  # .to()/.cuda()/.float() move or retype the parameters
        self._gram = None
        self.__dict__.pop("_gram_scaled", None)
        self.__dict__.pop("_gram_free", None)
        self.__dict__.pop("_param_list", None)   # the key's snapshots restart with the caches (a stale snapshot
        self.__dict__.pop("_key_state", None)    # made each of the next 16 calls find a "changed" slice and repack)
        self._gram_ws = None
        self._drop_metadata_cache()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._packed = None
        self._gram = None
        self.__dict__.pop("_gram_scaled", None)
        self.__dict__.pop("_gram_free", None)
        self.__dict__.pop("_param_list", None)
        self.__dict__.pop("_key_state", None)
        self._drop_metadata_cache()
        return super().load_state_dict(*args, **kwargs)

    def _drop_metadata_cache(self):
        for attr in ("_codebook_sims", "_subfeatures", "_subfeature_sims", "_layer_norms", "_codebook_used"):
            self.__dict__.pop(attr, None)

    def __getstate__(self):
        # picklable (scripts/3_make_rqae_features.py:190-192 ships the module over RPC):
        # the packed device block is a cache, rebuilt on first use.
        state = self.__dict__.copy()
        state["_packed"] = None
        state["_packed_key"] = None
        state["_gram"] = None
        state["_gram_ws"] = None
        state.pop("_gram_scaled", None)
        state.pop("_gram_free", None)
        state.pop("_gemma_inv", None)
        state["gram_options"] = None      # a ctypes struct: not picklable, and a per-process tuning choice anyway
        state.pop("_param_list", None)
        state.pop("_key_state", None)
        return state

    def _dims(self):
        return (self._dim, self.codebook_dim, self.codebook.shape[1], self.num_quantizers)

    _KEY_WALK_ALL = 512      # parameter tensors up to which every call also re-reads every storage pointer
    _KEY_SLICES = 16         # beyond that: one sixteenth of the POINTERS per call, round robin (see _weights_key)
    key_check = "full"       # "full" (default): every tensor's version counter at every call; "sliced": opt-in, see _weights_key

    def _weights_key(self):
        """Fingerprint of the parameters the packed blocks were built from: device, mode, the version counters and the storage
        pointers of the parameter tensors (the reference reads the weights live at every call, rqae/model.py:199-230, so an
        in-place edit of any level must be seen by the NEXT call).

        Default (``key_check = "full"``): the version counter of EVERY tensor is read at EVERY call -- one C-level pass
        (``sum(map(attrgetter("_version"), ...))``: 0.29 ms for the 4097 tensors of the released shape, 0.6 us for C2's 33) -- so
        an in-place edit (``p.zero_()``, ``p.mul_()``, ``p.copy_()`` under ``no_grad``) is never missed.  Storage pointers change
        only when a parameter is re-bound (``.to()`` / ``_apply`` / ``load_state_dict`` drop the caches themselves; ``p.data = t`` is
        the rest): up to 512 tensors every pointer is re-read at every call as well, larger modules re-read a sixteenth of the
        pointers per call, round robin.  ``key_check = "sliced"`` (explicit opt-in, for the 1024-level module in a latency-critical
        loop whose weights are frozen): versions too are re-read a sixteenth per call -- an in-place edit is then seen within 16
        calls; ``invalidate_packed()`` forces it.  A write through ``.data`` that keeps the storage (``p.data.zero_()``) bumps no
        counter in either mode: call ``invalidate_packed()`` after it."""
        plist = self.__dict__.get("_param_list")
        if plist is None or len(plist) != 4 * len(self.layers) + 1:
            plist = [self.codebook]
            for lin_in, lin_out in self.layers:
                plist += [lin_in.weight, lin_in.bias, lin_out.weight, lin_out.bias]
            self.__dict__["_param_list"] = plist
            self.__dict__["_key_state"] = None
        n = len(plist)
        if n <= self._KEY_WALK_ALL:
            ptr = 0
            for t in plist:
                ptr ^= t.data_ptr()
            return (self.codebook.device, self.training, ptr, sum(map(_TENSOR_VERSION, plist)), n)
        sliced = self.key_check == "sliced" and not self.training
        # large module: per-tensor snapshot of (version, pointer), one slice re-read per call; the key carries a generation number
        # that moves whenever a re-read tensor differs from its snapshot -- and, by default, the sum of ALL version counters
        st = self.__dict__.get("_key_state")
        if st is None or len(st["snap"]) != n:
            st = {"snap": [(t._version, t.data_ptr()) for t in plist], "gen": 0, "turn": 0}
            self.__dict__["_key_state"] = st
        else:
            snap, k = st["snap"], st["turn"]
            changed = False
            for i in [0, *range(1 + k, n, self._KEY_SLICES)]:
                t = plist[i]
                cur = (t._version, t.data_ptr())
                if cur != snap[i]:
                    snap[i] = cur
                    changed = True
            st["turn"] = (k + 1) % self._KEY_SLICES
            if changed:
                st["gen"] += 1
        return (self.codebook.device, self.training, st["gen"], -1 if sliced else sum(map(_TENSOR_VERSION, plist)), n)

    def invalidate_packed(self):
        """Drop the packed parameter blocks (call after a ``.data`` write that the version counters cannot see)."""
        self._packed = None
        self._gram = None
        self.__dict__.pop("_gram_scaled", None)
        self.__dict__.pop("_gram_free", None)
        self.__dict__.pop("_param_list", None)
        self.__dict__.pop("_key_state", None)

    def _stacked(self):
        w_in = torch.stack([l[0].weight for l in self.layers]).float().contiguous()
        b_in = torch.stack([l[0].bias for l in self.layers]).float().contiguous()
        w_out = torch.stack([l[1].weight for l in self.layers]).float().contiguous()
        b_out = torch.stack([l[1].bias for l in self.layers]).float().contiguous()
        return w_in, b_in, w_out, b_out

    def _pack(self, codebook: torch.Tensor) -> torch.Tensor:
        dev = self.codebook.device
        if dev.type != "cuda":
            raise RuntimeError(
                f"rqae_amd: module parameters are on {dev}; the RQAE hot path runs on a ROCm "
                "device only (no CPU fallback). Call .to('cuda') first.")
        with torch.no_grad():
            w_in, b_in, w_out, b_out = self._stacked()
            cb = codebook.detach().float().contiguous()
            if not torch.isfinite(cb).all():
                raise RuntimeError("rqae_amd: codebook contains non-finite values")
            shared = bool((cb == cb[:1]).all())
            return _native.pack(w_in, b_in, cb, w_out, b_out, shared)

    def packed_weights(self) -> torch.Tensor:
        """Device block in the kernels' layout (include/rqae_hip.h: rqae_hip_pack).

        Rebuilt when parameters were moved, reloaded or modified in place.
        """
        key = self._weights_key()
        if self._packed is None or self._packed_key != key:
            self._packed = self._pack(self.codebook)
            self._packed_key = key
        return self._packed

    def _packed_for_encode(self, levels: int) -> torch.Tensor:
        """Parameter block for one forward call.

        FSQ lattices are constants: the cached block.  Any other method renormalises ALL
        codebooks in place once per level (rqae/model.py:196 -> 126-131), which is not a
        bit-wise no-op, so level l is quantised against the rows as they stand after l+1
        renormalisations; the per-level snapshots are packed for this call only and kept
        in ``_codebook_used`` for inspection.
        """
        if self.quantization_method in _FSQ_METHODS:
            return self.packed_weights()
        with torch.no_grad():
            used = self._renormalise_for_call(levels)
            return self._pack(used)

    def _renormalise_for_call(self, levels: int) -> torch.Tensor:
        """The in-place renormalisations of one ``levels``-level call (rqae/model.py:196 -> 126-131) as ONE kernel
        (rqae_hip_normalize_codebooks: every row through its ``levels`` steps, torch-CPU's arithmetic order -- the reference's bits):
        ``self.codebook`` ends as the reference leaves it, the returned [L,K,c] tensor holds what each level < levels is quantised
        against (kept in ``_codebook_used`` for inspection)."""
        cb = self.codebook.data
        if cb.dtype != torch.float32 or not cb.is_contiguous():
            raise RuntimeError("rqae_amd: the codebook parameter must be contiguous float32")
        used = _native.normalize_codebooks(cb, levels)
        self.__dict__["_codebook_used"] = used[:levels]
        self._packed = None  # the parameter moved under the cached exact-arm block (.data writes bump no version counter)
        return used

    # ------------------------------------------------------------ matrix-core arm
    def _use_gram(self, n_tokens: int, levels: int, want_recon: bool = True) -> bool:
        if self.arm == "exact":
            return False
        # FSQ lattices: one codebook for every level; any other method: one per level, streamed level by level
        ok = _native.encode_gram_supported(self._dims(), self.quantization_method in _FSQ_METHODS, levels, want_recon)
        if self.arm == "gram":
            if not ok:
                raise RuntimeError(f"rqae_amd: arm='gram' does not serve shape (d, c, K, L) = {self._dims()} at {levels} levels"
                                   f"{' with reconstruction' if want_recon else ''} "
                                   "(include/rqae_hip.h: rqae_hip_encode_gram_supported); use arm='exact' or 'auto'")
            return True
        return ok and n_tokens >= (self.GRAM_MIN_TOKENS_DEEP if levels > 64 else self.GRAM_MIN_TOKENS)

    def gram_packed_weights(self) -> torch.Tensor:
        """The matrix-core arm's own parameter block (include/rqae_hip.h: rqae_hip_gram_pack), cached like
        ``packed_weights``."""
        key = self._weights_key()
        if self._gram is None or self._gram[0] != key:
            dev = self.codebook.device
            if dev.type != "cuda":
                raise RuntimeError(f"rqae_amd: module parameters are on {dev}; the RQAE hot path runs on a ROCm "
                                   "device only (no CPU fallback). Call .to('cuda') first.")
            with torch.no_grad():
                w_in, b_in, w_out, b_out = self._stacked()
                cb = self.codebook.detach().float().contiguous()
                if not torch.isfinite(cb).all():
                    raise RuntimeError("rqae_amd: codebook contains non-finite values")
                if not bool((cb == cb[:1]).all()):
                    raise RuntimeError("rqae_amd: the matrix-core arm needs one codebook shared by all levels")
                self._gram = (key, _native.gram_pack(w_in, b_in, cb, w_out, b_out, True))
        return self._gram[1]

    def gram_packed_weights_scaled(self, norm_weight: torch.Tensor) -> torch.Tensor:
        """The matrix-core arm's block with Gemma's ``1 + norm.weight`` folded into the ``W_in`` image (include/rqae_hip.h:
        rqae_hip_gram_pack's in_scale), for ``rqae_hip_encode_gram_gemma``; cached per (parameters, norm weight)."""
        key = (self._weights_key(), norm_weight.data_ptr(), norm_weight._version)
        ent = self.__dict__.get("_gram_scaled")
        if ent is None or ent[0] != key or ent[2] is not norm_weight:
            with torch.no_grad():
                w_in, b_in, w_out, b_out = self._stacked()
                cb = self.codebook.detach().float().contiguous()
                if not bool((cb == cb[:1]).all()):
                    raise RuntimeError("rqae_amd: the matrix-core arm needs one codebook shared by all levels")
                scale = 1.0 + norm_weight.detach().float()
                ent = (key, _native.gram_pack(w_in, b_in, cb, w_out, b_out, True, in_scale=scale), norm_weight)
            self.__dict__["_gram_scaled"] = ent
        return ent[1]

    def _encode_gram(self, x2d, levels, want_recon):
        need = _native.gram_workspace_bytes(self._dims(), x2d.shape[0], levels)
        ws = self._gram_ws
        if ws is None or ws.device != x2d.device or ws.numel() < need:
            ws = self._gram_ws = torch.empty(need, dtype=torch.uint8, device=x2d.device)
        packed, shared = self._gram_packed_for_encode(levels)
        return _native.encode_gram(packed, self._dims(), shared, x2d, levels,
                                   want_recon=want_recon, workspace=ws, options=self.__dict__.get("gram_options"))

    def _gram_packed_for_encode(self, levels: int):
        """(block, codebook_shared) for one call on the matrix-core arm.  FSQ lattices: the cached block.  Any other method
        renormalises all codebooks in place once per level (rqae/model.py:196 -> 126-131), as ``_packed_for_encode`` spells out;
        the per-level snapshots are packed with one codebook per level, and the block is kept for as long as the snapshots come
        out the same (renormalising unit rows reaches a fixed point after a call or two)."""
        if self.quantization_method in _FSQ_METHODS:
            return self.gram_packed_weights(), True
        with torch.no_grad():
            used = self._renormalise_for_call(levels)
            key = self._weights_key()
            ent = self.__dict__.get("_gram_free")
            if ent is None or ent[0] != key:
                # the linear maps' images and the Gram blocks: once per parameter state
                if not torch.isfinite(used[:levels]).all():
                    raise RuntimeError("rqae_amd: codebook contains non-finite values")
                w_in, b_in, w_out, b_out = self._stacked()
                ent = self.__dict__["_gram_free"] = (key, None, _native.gram_pack(w_in, b_in, used, w_out, b_out, False))
            else:
                # ... the codebook part again at every call, on the device (one launch, no comparison on the host)
                _native.gram_repack_codebooks(ent[2], self._dims(), used)
        return ent[2], False

    # ------------------------------------------------------------- small mirrors
    def update_codebook_counts(self, indices):
        """No-op, as in the reference (rqae/model.py:101-104 returns before any work)."""
        return None

    def normalize_codebooks(self):
        """Row-normalise every level's codebook in place unless it is an FSQ lattice
        (rqae/model.py:126-131)."""
        if self.quantization_method in _FSQ_METHODS:
            return
        cb = self.codebook.data
        self.codebook.data.copy_(cb / cb.norm(dim=-1, keepdim=True))
        self._packed = None  # .data writes do not bump the parameter's version counter

    # -------------------------------------------------- cached tables
    # On a ROCm device the tables come from the HIP kernels of csrc/tables.hpp (no torch fallback there); a module
    # still on the CPU (before .to("cuda"), as RQAEFeature.from_quantizer may see it) computes them as the
    # reference does.  codebook_sims and subfeatures have the reference's CPU bits either way.
    def _on_device(self) -> bool:
        return self.codebook.device.type == "cuda"

    def _table_inputs(self):
        with torch.no_grad():
            _, _, w_out, b_out = self._stacked()
            return self.codebook.detach().float().contiguous(), w_out, b_out

    @property
    def codebook_sims(self):
        """(K, K) fp16 cosine similarities of the level-0 codebook (rqae/model.py:133-143)."""
        if "_codebook_sims" not in self.__dict__:
            if self.quantization_method != "round_fsq":
                raise ValueError("Codebook sims only supported for round_fsq for now")
            if self._on_device():
                self.__dict__["_codebook_sims"] = _native.codebook_sims(self.codebook.data[0])
            else:
                unit = F.normalize(self.codebook.data.detach().clone()[0], dim=-1)
                self.__dict__["_codebook_sims"] = (unit @ unit.T).to(torch.float16)
        return self.__dict__["_codebook_sims"]

    @property
    def subfeatures(self):
        """(L, K, d): lin_out_l applied to codebook[l] (rqae/model.py:145-156)."""
        if "_subfeatures" not in self.__dict__:
            if self._on_device():
                self.__dict__["_subfeatures"] = _native.subfeatures(*self._table_inputs())
            else:
                self.__dict__["_subfeatures"] = torch.stack(
                    [pair[1](self.codebook[l]) for l, pair in enumerate(self.layers)])
        return self.__dict__["_subfeatures"]

    @property
    def subfeature_sims(self):
        """(L, K, K) fp16 cosine similarities between subfeatures (rqae/model.py:158-168).

        On the device this is the quadratic form of the (c+1)x(c+1) Gram matrix in fp64 (csrc/tables.hpp); the
        (L, K, d) subfeatures are not materialised for it."""
        if "_subfeature_sims" not in self.__dict__:
            if self._on_device():
                sims, norms = _native.subfeature_sims(*self._table_inputs())
                self.__dict__["_subfeature_sims"] = sims
                self.__dict__.setdefault("_layer_norms", norms)
            else:
                unit = F.normalize(self.subfeatures, dim=-1)
                self.__dict__["_subfeature_sims"] = (unit @ unit.transpose(-1, -2)).to(torch.float16)
        return self.__dict__["_subfeature_sims"]

    @property
    def layer_norms(self):
        """(L,) mean column norm of each lin_out weight (rqae/model.py:170-178)."""
        if "_layer_norms" not in self.__dict__:
            if self._on_device():
                self.__dict__["_layer_norms"] = _native.subfeature_sims(*self._table_inputs(), want_sims=False)[1]
            else:
                self.__dict__["_layer_norms"] = torch.tensor(
                    [pair[1].weight.data.norm(dim=0).mean().item() for pair in self.layers],
                    device=self.layers[0][1].weight.device)
        return self.__dict__["_layer_norms"]

    # ------------------------------------------------------------------ hot path
    def _deterministic(self, temperature) -> bool:
        return temperature < 1e-7 or not self.training

    def gumbel_sample(self, cos_sim, temperature=0.0):
        """Index per row of cos_sim (rqae/model.py:180-185): argmax in eval mode or at temperature < 1e-7,
        otherwise the hard gumbel-softmax sample the reference draws while training."""
        if self._deterministic(temperature):
            return cos_sim.argmax(dim=-1)
        return F.gumbel_softmax(cos_sim, tau=temperature, hard=True).argmax(dim=-1)

    def _needs_autograd(self, x) -> bool:
        """True while training with gradients flowing: the HIP kernels are inference kernels (no autograd graph), so
        such a call is served by the torch ops of ``_forward_training`` -- the training-time branch of the reference
        (straight-through estimator, optional gumbel sampling), never the inference hot path."""
        return self.training and torch.is_grad_enabled() and (
            x.requires_grad or any(p.requires_grad for p in self.layers.parameters()))

    def _quantize_training(self, x, layer_idx, temperature):
        unit = x / x.norm(dim=-1, keepdim=True)
        rows = self.codebook[layer_idx]
        picked = self.gumbel_sample(unit @ rows.T, temperature=temperature)
        return rows[picked], picked

    def _forward_training(self, x, levels, temperature):
        """Training-time forward with torch ops (autograd + stochastic sampling; rqae/model.py:199-230)."""
        residual, total, picked_all = x, 0, []
        for level in range(levels):
            lin_in, lin_out = self.layers[level]
            proj = lin_in(residual)
            self.normalize_codebooks()
            rows, picked = self._quantize_training(proj, level, temperature)
            picked_all.append(picked)
            step = lin_out(proj + (rows - proj).detach())       # straight-through estimator
            residual = residual - step.detach()
            total = total + step
        codes = torch.stack(picked_all).permute(1, 2, 0)
        self.update_codebook_counts(codes)
        return total, codes

    def quantize_gumbel(self, x, layer_idx, temperature=0.0):
        """x[..., c] -> (nearest unit codebook row by cosine, its index) for one level
        (rqae/model.py:187-193); one launch of the HIP quantize kernel (torch ops while training with
        temperature >= 1e-7 or gradients)."""
        if not self._deterministic(temperature) or self._needs_autograd(x):
            return self._quantize_training(x, layer_idx, temperature)
        lead = x.shape[:-1]
        p = x.reshape(-1, x.shape[-1]).float().contiguous()
        cv, idx = _native.quantize(p, self.codebook[layer_idx].detach().float().contiguous())
        return cv.view(*lead, -1), idx.view(*lead)

    def quantize(self, x, layer_idx, temperature=0.0):
        self.normalize_codebooks()
        return self.quantize_gumbel(x, layer_idx, temperature)

    def forward(self, x, max_layers: int = float("inf"), temperature=0.0):
        """Encode: x[B,S,d] -> (quantized_out[B,S,d] f32, indices[B,S,levels] i64).

        Same values as the reference level loop (rqae/model.py:199-230); levels =
        min(num_quantizers, max_layers).  x may be fp32 (reference), bf16 or fp16
        (widened exactly to fp32 on load).  One launch of the exact arm, or the three kernels of the matrix-core
        arm when ``self.arm`` selects it (see __init__).  While training with gradients or temperature >= 1e-7 the
        call is served by torch ops (``_forward_training``): the kernels are inference kernels.
        """
        if x.dim() != 3:
            raise RuntimeError(f"RQAE.forward expects x of shape (B, S, dim), got {tuple(x.shape)}")
        levels = self.num_quantizers if max_layers >= self.num_quantizers else int(math.ceil(max_layers))
        if levels <= 0:
            raise RuntimeError("RQAE.forward: max_layers leaves no quantizer to run (stack of an empty list)")
        B, S, d = x.shape
        if d != self._dim:
            raise RuntimeError(f"RQAE.forward: last dim {d} != model dim {self._dim}")
        if not self._deterministic(temperature) or self._needs_autograd(x):
            return self._forward_training(x, levels, temperature)
        x2d = x.reshape(B * S, d).contiguous()
        if self._use_gram(B * S, levels):
            q, idx = self._encode_gram(x2d, levels, True)
        else:
            q, idx = _native.encode(self._packed_for_encode(levels), self._dims(), x2d, levels)
        idx = idx.view(B, S, levels)
        self.update_codebook_counts(idx)
        return q.view(B, S, d), idx

    def encode(self, x, max_layers: int = float("inf")):
        """Codes only (skips writing the reconstruction): x[B,S,d] -> indices[B,S,levels]."""
        if x.dim() != 3:
            raise RuntimeError(f"RQAE.encode expects x of shape (B, S, dim), got {tuple(x.shape)}")
        levels = self.num_quantizers if max_layers >= self.num_quantizers else int(math.ceil(max_layers))
        if levels <= 0:
            raise RuntimeError("RQAE.encode: max_layers leaves no quantizer to run")
        B, S, d = x.shape
        if d != self._dim:
            raise RuntimeError(f"RQAE.encode: last dim {d} != model dim {self._dim}")
        x2d = x.reshape(B * S, d).contiguous()
        if self._use_gram(B * S, levels, want_recon=False):
            _, idx = self._encode_gram(x2d, levels, False)
        else:
            _, idx = _native.encode(self._packed_for_encode(levels), self._dims(), x2d, levels, want_recon=False)
        return idx.view(B, S, levels)

    def indices_to_codebook_values(self, indices):
        """[B,S,NQ] -> [B,S,NQ,c] rows of the LEVEL-0 codebook (rqae/model.py:232-234)."""
        B, S, NQ = indices.shape
        return self.codebook[0][indices]

    def _level_mask(self, layers, nq):
        L = self.num_quantizers
        keep = [layers is None or (l in layers) for l in range(L)]
        for l in range(L):
            if keep[l] and l >= nq:  # reference: codebook_values[:, :, l] out of range
                raise IndexError(f"index {l} is out of bounds for dimension 2 with size {nq}")
        return keep

    def decode_from_codebook_values(self, codebook_values, layers=None):
        """Sum over (whitelisted) levels of lin_out_l(cv[:, :, l]) in level order
        (rqae/model.py:236-248); None when nothing is selected."""
        B, S, NQ, CD = codebook_values.shape
        keep = self._level_mask(layers, NQ)
        if not any(keep):
            return None
        out = _native.decode_values(self.packed_weights(), self._dims(),
                                    codebook_values.reshape(B * S, NQ, CD).float().contiguous(), keep)
        return out.view(B, S, -1)

    def decode(self, indices, layers=None):
        """indices[B,S,NQ] -> reconstruction[B,S,d] (rqae/model.py:250-252); one launch."""
        B, S, NQ = indices.shape
        keep = self._level_mask(layers, NQ)
        if not any(keep):
            return None
        idx = indices.reshape(B * S, NQ)
        if idx.dtype not in (torch.int64, torch.int32, torch.uint16):
            idx = idx.long()
        idx = idx.contiguous()
        if (all(keep) and self.quantization_method in _FSQ_METHODS and self._use_gram(B * S, NQ, want_recon=False)
                and _native.decode_gram_supported(self._dims(), True, NQ)):
            # the matrix-core arm's reconstruction kernel fed by the codes: same sum within 1e-6 relative, HBM-write-bound
            out = _native.decode_gram(self.gram_packed_weights(), self._dims(), True, idx)
        else:
            out = _native.decode(self.packed_weights(), self._dims(), idx, keep)
        return out.view(B, S, -1)

    # ----------------------------------------------------------------- LLM hook
    def hook(self, **kwargs):
        """Forward-hook factory with the reference's keyword protocol
        (rqae/model.py:254-291): llm= | norm=,denorm= ; store=, skip_bos=, replace=."""
        if "llm" in kwargs:
            llm = kwargs.pop("llm")
            assert hasattr(llm, "norm") and hasattr(llm, "denorm"), \
                "RQAE hook requires norm and denorm from LLM"
            fused = kwargs.pop("fused", True)
            generic = self.hook(norm=llm.norm, denorm=llm.denorm, **kwargs)
            gemma = _gemma_final_norm(llm)
            if gemma is not None and fused:
                return self._gemma_hook(gemma, generic, **kwargs)
            return generic
        if "norm" not in kwargs:
            raise ValueError("RQAE hook requires norm from LLM")
        if "denorm" not in kwargs:
            raise ValueError("RQAE hook requires denorm from LLM")
        norm, denorm = kwargs["norm"], kwargs["denorm"]
        store = kwargs.get("store", lambda name, value: None)
        skip_bos = kwargs.get("skip_bos", True)
        replace = kwargs.get("replace", True)

        def hook_fn(module, input, output):
            hs = output[0].float()
            store("original", hs.detach().clone())
            normed = norm(hs)
            store("normed", normed.detach().clone())
            q_out, codes = self(normed)
            store("quantized", q_out.detach().clone())
            store("indices", codes.detach().clone())
            q_out = denorm(q_out, hs)
            if skip_bos:
                q_out[:, 0] = hs[:, 0]
            store("new", q_out.detach().clone())
            if replace:
                output[0].data.copy_(q_out)

        return hook_fn

    def _gemma_hook(self, final_norm, generic_hook, **kwargs):
        """hook() for an LLM whose norm/denorm are Gemma-2's (rqae/llm.py:65-73): the RMSNorm before
        forward runs inside the encode kernel (rqae_hip_encode_gemma; as its own kernel when a store
        wants `normed`), the denorm + BOS skip + in-place write-back after it as one HIP kernel
        (rqae_hip_gemma_denorm_patch) -- instead of ~12 torch ops and their fp32 temporaries.  Same
        store() protocol and side effects as the generic hook."""
        store = kwargs.get("store", None)
        skip_bos = kwargs.get("skip_bos", True)
        replace = kwargs.get("replace", True)
        eps = float(getattr(final_norm, "eps", 1e-6))

        def hook_fn(module, input, output):
            hs_llm = output[0]
            if (not hs_llm.is_cuda or not hs_llm.is_contiguous() or hs_llm.dim() != 3
                    or hs_llm.dtype not in (torch.float16, torch.bfloat16, torch.float32)):
                return generic_hook(module, input, output)   # same values, torch ops around forward()
            if store is not None:
                store("original", hs_llm.detach().float().clone() if hs_llm.dtype == torch.float32 else hs_llm.detach().float())
            # (small batches run on 4-token workgroups, which only the plain encode kernel has: below 8 tokens per
            #  CU the norm as its own kernel + forward is the faster pair -- 3.06 vs 3.63 ms at 4 x 128 tokens, 1024 levels)
            n_tok = hs_llm.shape[0] * hs_llm.shape[1]
            fold = (store is None and hs_llm.shape[-1] == self._dim and self.quantization_method in _FSQ_METHODS
                    and n_tok > 8 * torch.cuda.get_device_properties(hs_llm.device).multi_processor_count
                    and _native.encode_gemma_supported(self._dims()))
            fold_gram = (store is None and hs_llm.shape[-1] == self._dim and self.quantization_method in _FSQ_METHODS
                         and self._use_gram(n_tok, self.num_quantizers)
                         and _native.encode_gram_gemma_supported(self._dims(), hs_llm.dtype))
            if fold_gram:
                # the matrix-core arm (self.arm = "gram" / "auto"): 1 + w folded into the W_in image, sum(hs^2) taken in the
                # projection kernel, projections scaled by 1/rms -- Tier-2 parity like every call on this arm
                need = _native.gram_workspace_bytes(self._dims(), n_tok, self.num_quantizers)
                ws = self._gram_ws
                if ws is None or ws.device != hs_llm.device or ws.numel() < need:
                    ws = self._gram_ws = torch.empty(need, dtype=torch.uint8, device=hs_llm.device)
                gp = self.gram_packed_weights_scaled(final_norm.weight)
                vouched = self.__dict__.get("gram_options") is None and getattr(gp, "lattice_n", 0) >= 2      # default options: lean_levels = 5
                if replace and _native.encode_gram_gemma_patch_supported(self._dims(), hs_llm.dtype, self.num_quantizers, vouched):
                    # ... and the way back folded into the reconstruction's stores: hs is patched in place, `quantized` never exists
                    inv = self.__dict__.get("_gemma_inv")
                    if inv is None or inv[0] is not final_norm.weight or inv[1] != final_norm.weight._version:
                        inv = self.__dict__["_gemma_inv"] = (final_norm.weight, final_norm.weight._version,
                                                             _native.gemma_inv_1pw(final_norm.weight))
                    codes, _ = _native.encode_gram_gemma_patch(self.gram_packed_weights_scaled(final_norm.weight), self._dims(), hs_llm,
                                                               inv[2], eps, self.num_quantizers, skip_bos, workspace=ws,
                                                               options=self.__dict__.get("gram_options"))
                    self.update_codebook_counts(codes)
                    return
                q_out, codes, inv_rms = _native.encode_gram_gemma(self.gram_packed_weights_scaled(final_norm.weight), self._dims(),
                                                                  hs_llm, eps, self.num_quantizers, workspace=ws,
                                                                  options=self.__dict__.get("gram_options"))
                self.update_codebook_counts(codes)
            elif fold:
                # nobody asked for `normed`: the norm runs inside the encode kernel's load (same bits)
                q_out, codes, inv_rms = _native.encode_gemma(self.packed_weights(), self._dims(), hs_llm,
                                                             final_norm.weight, eps, self.num_quantizers)
                self.update_codebook_counts(codes)
            else:
                normed, inv_rms = _native.gemma_norm(hs_llm, final_norm.weight, eps)
                if store is not None:
                    store("normed", normed.detach().clone())
                q_out, codes = self(normed)
            if store is not None:
                store("quantized", q_out.detach().clone())
                store("indices", codes.detach().clone())
            target = hs_llm if replace else hs_llm.clone()
            new = _native.gemma_denorm_patch(q_out, final_norm.weight, inv_rms, target, skip_bos,
                                             want_new=store is not None)
            if store is not None:
                store("new", new)

        return hook_fn


def _gemma_final_norm(llm):
    """The RMSNorm module behind Gemma2.norm (rqae/llm.py:65-66: ``self.model.model.norm``), or None."""
    norm = getattr(getattr(getattr(llm, "model", None), "model", None), "norm", None)
    if norm is not None and hasattr(norm, "weight") and type(norm).__name__.lower().startswith("gemma"):
        return norm
    return None
