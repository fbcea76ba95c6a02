"""Similarity search over a dataset of codes -- the demo server's ``IntensityEngine``.

Mirror of /root/reference/demo/server/server.py: ``get_intensities`` (:41-68) and ``IntensityEngine.find_examples``
(:160-325), with the Modal / FastAPI shell left out.  Same arguments, same yielded structure; the arithmetic runs in
csrc/search.hip through the C ABI (include/rqae_hip.h, "Similarity search").  What is different on purpose:

* the dataset's codes live in HBM for the engine's lifetime (uint16 when the codebook allows it: 266 MB per 1024-sequence
  shard of depth 1023; the reference keeps int32 on the host and copies a slice per layer range and shard);
* the ``(N, S, Q, levels)`` gather the reference materialises per chunk never exists: the kernel sums it as it gathers;
* equal maxima rank in ascending sequence order (``torch.argsort`` in the reference promises no order for ties).
"""
from typing import List, Optional, Sequence

import torch

from . import _native

DEFAULT_LAYERS = [4, 6, 8, 12, 16, 24, 32, 48, 64, 128, 256, 512, 1023]
QP = 128   # accumulator / table row width (query positions, padded)


def get_intensities(activations: torch.Tensor, sims: torch.Tensor) -> torch.Tensor:
    """server.py:41-68: ``out[b, s, q, l] = sims[q, l, activations[b, s, l]]``.

    Kept for callers of the reference's helper; a plain gather with no arithmetic (``find_examples`` below never forms this
    tensor)."""
    B, S, D = activations.shape
    S2, D2, _ = sims.shape
    if D != D2:
        raise RuntimeError(f"get_intensities: {D} levels of codes, {D2} levels of sims")
    lv = torch.arange(D, device=sims.device)
    return sims[:, lv, activations.long().reshape(B * S, D)].permute(1, 0, 2).reshape(B, S, S2, D)


class IntensityEngine:
    """``sims``: the table the server keeps (server.py:99-112) -- ``subfeature_sims * layer_norms[:, None, None]``,
    ``[num_quantizers, K, K]`` (any float dtype; held as fp16, which is what ``find_examples`` casts it to, :190).
    ``activations``: the dataset's codes, a list of shards ``[n_i, S, num_quantizers]`` or one tensor (BOS already
    dropped, as ``load_file`` does at :127-130)."""

    def __init__(self, sims: torch.Tensor, activations, device=None, shard_size: int = 1024):
        device = torch.device(device if device is not None else "cuda")
        if device.type != "cuda":
            raise RuntimeError("IntensityEngine runs on the GPU only (rqae_amd has no CPU fallback)")
        _native.lib()
        shards = [activations] if torch.is_tensor(activations) else list(activations)
        if not shards:
            raise ValueError("no activations")
        K = sims.shape[1]
        # the whole table in fp16 on the device: (1023, 625, 625) is 0.8 GB
        self.sims = sims.detach().to(device=device, dtype=torch.float16).contiguous()
        code_dtype = torch.uint16 if K <= 65536 else torch.int32
        parts = []
        for s in shards:
            if s.dim() != 3 or s.shape[1:] != shards[0].shape[1:]:
                raise ValueError("every shard must be [n, S, num_quantizers]")
            s = s.to(device)
            _native.check_codes(s, K, "IntensityEngine activations", always=False)
            s = s.to(torch.int32)
            # int32 -> uint16 by way of int16 (the wrap-around cast keeps the low 16 bits)
            parts.append(s.to(torch.int16).view(torch.uint16) if code_dtype == torch.uint16 else s)
        self.activations = torch.cat(parts, dim=0).contiguous()          # [N, S, L]
        self.shard_size = shard_size
        self.device = device
        self.K = K
        N, S, _ = self.activations.shape
        self._acc = torch.empty((N, S, QP), dtype=torch.float16, device=device)
        self._rank_ws = None
        self._table = None

    @property
    def num_sequences(self) -> int:
        return self.activations.shape[0]

    def _query(self, idx, activation) -> torch.Tensor:
        if activation is not None and idx is not None:
            raise ValueError("Cannot specify both idx and activation")
        if idx is not None:
            # server.py:175-176 looks the sequence up as activations[idx // 1024][idx % 1024]: with 1024-sequence shards that is
            # row idx of the concatenated dataset
            if not 0 <= idx < self.num_sequences:
                raise IndexError(f"idx {idx} outside the dataset of {self.num_sequences} sequences")
            q = self.activations[idx]
            return (q.view(torch.int16).to(torch.int32) & 0xFFFF) if q.dtype == torch.uint16 else q.to(torch.int32)
        if activation is not None:
            q = torch.as_tensor(activation).to(self.device)
            if q.dim() != 2:
                raise ValueError("activation must be [positions, num_quantizers]")
            return q.to(torch.int32)
        raise ValueError("Must specify either idx or activation")

    def find_examples(self, idx: Optional[int] = None, activation: Optional[torch.Tensor] = None, top_examples: int = 30,
                      middle_examples: int = 10, bottom_examples: int = 10, layers: Sequence[int] = tuple(DEFAULT_LAYERS)):
        """Generator, one item per entry of ``layers`` (server.py:160-325): ``({"top"|"middle"|"bottom": {"indices":
        int32 [Q, E], "intensities": fp16 [Q, E, S]}}, layer)`` on the CPU, like the reference's ``.cpu()`` results."""
        layers = [int(v) for v in layers]
        if not layers or any(b <= a for a, b in zip([0] + layers[:-1], layers)):
            raise ValueError("layers must be increasing and positive")
        N, S, L = self.activations.shape
        if layers[-1] > L or layers[-1] > self.sims.shape[0]:
            raise ValueError(f"layers reach level {layers[-1]}; the codes have {L}, the table {self.sims.shape[0]}")
        query = self._query(idx, activation)[:, : layers[-1]].contiguous()       # [Q, max(layers)]
        Q = query.shape[0]
        if query.shape[1] < layers[-1]:
            raise ValueError("the query has fewer levels than max(layers)")
        if not _native.search_supported(self.K, Q):
            raise RuntimeError(f"find_examples: K={self.K}, Q={Q} not served (Q <= 128, K <= 1280)")
        _native.check_codes(query, self.K, "find_examples query", always=True)
        bounds = [0] + layers
        for step, (layer, l0, l1) in enumerate(zip(layers, bounds[:-1], bounds[1:])):
            self._table = _native.search_query_table(self.sims, query, l0, l1, self._table)
            _native.search_accumulate(self.activations, self._table, self.K, Q, l0, l1, self._acc, first=(step == 0))
            _, order, self._rank_ws = _native.search_rank(self._acc, Q, self._rank_ws)
            mid0, mid1 = N // 2 - middle_examples // 2, N // 2 + middle_examples // 2
            picks = {"top": order[:, :top_examples], "middle": order[:, mid0:mid1], "bottom": order[:, N - bottom_examples:]}
            out = {}
            for name, ex in picks.items():
                ex = ex.contiguous()
                out[name] = {"indices": ex.cpu().int(), "intensities": _native.search_gather(self._acc, Q, ex).cpu()}
            yield out, layer

    def accumulated(self, Q: int) -> torch.Tensor:
        """The running fp16 accumulation ``[N, S, Q]`` after the last yielded layer (the reference's ``intensity_accumulation``)."""
        return self._acc[:, :, :Q]
