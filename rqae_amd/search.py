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
            if code_dtype == torch.uint16 and s.dtype != torch.uint16:
                # the wrap-around cast below keeps the low 16 bits: anything it would fold into range is rejected here, once, at
                # load (the accumulate kernel flags what is left: codes in K..65535)
                if bool(((s < 0) | (s > 65535)).any()):
                    raise IndexError(f"IntensityEngine: dataset code outside 0..{K - 1}")
            s = s.to(torch.int32) if s.dtype != torch.uint16 else s
            # int32 -> uint16 by way of int16 (the wrap-around cast keeps the low 16 bits)
            if code_dtype == torch.uint16 and s.dtype != torch.uint16:
                s = s.to(torch.int16).view(torch.uint16)
            parts.append(s)
        self.activations = torch.cat(parts, dim=0).contiguous()          # [N, S, L]
        self.shard_size = shard_size
        self.device = device
        self.K = K
        self._acc = None          # the accumulation of the most recent find_examples step (accumulated())

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
        int32 [Q, E], "intensities": fp16 [Q, E, S]}}, layer)`` on the CPU, like the reference's ``.cpu()`` results.

        The running accumulation, the range's table and the ranking workspace belong to THIS call (the reference keeps
        ``intensity_accumulation`` per call too, :206): several live generators of one engine -- the server streams them
        through ``find_examples.remote_gen`` -- do not disturb one another.  33 MB per 1024-sequence shard and call."""
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
        _native.codes_in_range(query, self.K, "find_examples query")    # server.py:195-198 indexes the table by these: IndexError there
        bounds = [0] + layers
        acc = torch.empty((N, S, QP), dtype=torch.float16, device=self.device)
        table = rank_ws = None
        for step, (layer, l0, l1) in enumerate(zip(layers, bounds[:-1], bounds[1:])):
            table = _native.search_query_table(self.sims, query, l0, l1, table)
            _native.search_accumulate(self.activations, table, self.K, Q, l0, l1, acc, first=(step == 0))
            _, order, rank_ws = _native.search_rank(acc, Q, rank_ws)
            # the reference's slices of the ranking, as written there (server.py:268-280): `[-bottom_examples:]` with
            # bottom_examples = 0 is the WHOLE ranking, and a middle window past the ends is clipped by the slice
            mid0, mid1 = N // 2 - middle_examples // 2, N // 2 + middle_examples // 2
            picks = {"top": order[:, :top_examples], "middle": order[:, mid0:mid1], "bottom": order[:, -bottom_examples:]}
            out = {}
            for name, ex in picks.items():
                ex = ex.contiguous()
                out[name] = {"indices": ex.cpu().int(), "intensities": _native.search_gather(acc, Q, ex).cpu()}
            self._acc = acc
            yield out, layer

    def accumulated(self, Q: int) -> torch.Tensor:
        """The running fp16 accumulation ``[N, S, Q]`` after the most recently yielded layer of the most recently advanced
        ``find_examples`` generator (the reference's ``intensity_accumulation``)."""
        if self._acc is None:
            raise RuntimeError("accumulated(): no find_examples step has run yet")
        return self._acc[:, :, :Q]
