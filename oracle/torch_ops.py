"""The level loop of RQAE.forward through PyTorch-CPU operators -- TEST INFRASTRUCTURE ONLY (never imported by rqae_amd/).

SURVEY.md 8(d) asks for the CPU baseline "through PyTorch-CPU ops": the reference's own path is ~10 ATen calls per level
(rqae/model.py:208-224 with 187-193 inside it).  This module issues the same operator sequence on the same shapes from the
stacked weights of ``synth.weights`` -- one affine map into c-space, the row normalisation, the [N,K] cosine product, argmax,
a row gather, the straight-through sum, one affine map back, one subtraction and one addition per level -- so that what
``bench.py`` times as ``cpu_baseline`` (kind "reference-ops") costs what the reference costs on the host, op for op.  The C
restatement (rqae_oracle.c) stays the parity checker; this one is pinned beside it by tests/test_oracle_golden.py (codes and
reconstruction rows bit-equal to the reference captures: the operators are the ones the capture ran).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


class TorchOpsModel:
    """Stacked parameters as CPU tensors; ``forward`` = rqae/model.py:199-230 in eval mode (temperature 0)."""

    def __init__(self, w: dict):
        self.w_in = torch.from_numpy(np.ascontiguousarray(w["w_in"]))        # [L, c, d]
        self.b_in = torch.from_numpy(np.ascontiguousarray(w["b_in"]))        # [L, c]
        self.w_out = torch.from_numpy(np.ascontiguousarray(w["w_out"]))      # [L, d, c]
        self.b_out = torch.from_numpy(np.ascontiguousarray(w["b_out"]))      # [L, d]
        self.codebook = torch.from_numpy(np.ascontiguousarray(w["codebook"]))  # [L, K, c]
        self.levels = self.w_in.shape[0]

    @torch.inference_mode()
    def forward(self, x: torch.Tensor, max_layers=None):
        """x [B, S, d] fp32 -> (reconstruction [B, S, d] fp32, codes [B, S, levels] int64)."""
        assert x.dim() == 3, "RQAE.forward takes [B, S, d] (rqae/model.py:226 permutes three axes)"
        depth = self.levels if max_layers is None else min(self.levels, int(max_layers))
        left = x                      # what is still to be explained
        total = 0                     # what the levels so far explain
        picked = []
        for level in range(depth):
            into_c = F.linear(left, self.w_in[level], self.b_in[level])          # :211
            direction = into_c / into_c.norm(dim=-1, keepdim=True)               # :188
            scores = torch.matmul(direction, self.codebook[level].T)             # :190
            which = scores.argmax(dim=-1)                                        # :182 (eval: first maximum)
            rows = self.codebook[level][which]                                   # :192
            through = into_c + (rows - into_c)                                   # :218-220 (value-affecting rounding)
            back = F.linear(through, self.w_out[level], self.b_out[level])       # :221
            left = left - back                                                   # :223
            total = total + back                                                 # :224
            picked.append(which)
        return total, torch.stack(picked).permute(1, 2, 0)                       # :226


def forward(x: np.ndarray, w: dict, max_layers=None, threads: int = 0):
    """numpy front end: x [n, d] fp32 -> (q [n, d] fp32, codes [n, levels] int64)."""
    if threads:
        torch.set_num_threads(int(threads))
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    n, d = xt.shape
    q, codes = TorchOpsModel(w).forward(xt.view(1, n, d), max_layers)
    return q.reshape(n, d).numpy(), codes.reshape(n, -1).contiguous().numpy()
