"""Host-side mirror of rqae/feature.py for the consumer of the codes: ``RQAEFeature.intensity``.

Same classes, constructor arguments, attributes and methods as the reference (rqae/feature.py:9-137);
``intensity`` (rqae/feature.py:102-129) runs as one HIP kernel behind the C ABI
(``rqae_hip_intensity_prepare`` / ``rqae_hip_intensity``, include/rqae_hip.h) instead of five torch ops over
a ``[..., max_layer]`` fp16 temporary.  Results are bit-equal to the reference's fp16 output.  There is no CPU
path: the model must live on a ROCm device; index tensors given on the CPU are copied to it and the result is
returned on the caller's device, as the reference's callers expect (scripts/3_make_rqae_features.py:107-113).
"""
from typing import List, Optional

import numpy as np
import torch

from . import _native


class Feature:
    """Plain record, as rqae/feature.py:9-39."""

    def __init__(self, id: str = "", explanation: str = "", scores: dict = {}, model: str = "",
                 activations: list = []):
        self.id = str(id)
        self.explanation = str(explanation)
        self.scores = scores
        self.model = str(model)
        self.activations = activations  # list of {"text": [str], "activations": [float]}

    def save(self, file_path: str):
        np.savez(file_path, **self.__dict__)

    @classmethod
    def load(cls, file_path: str):
        params = dict(np.load(file_path, allow_pickle=True))
        for k, v in params.items():
            try:
                params[k] = v.item()
            except Exception:
                pass
        return cls(**params)


class RQAEFeature:
    """rqae/feature.py:42-137."""

    def __init__(self, id: str = "", explanations: List[str] = None, scores: List[dict] = None,
                 activations: list = None, model: str = "", num_quantizers: int = 1024, dim: int = 4,
                 layers: List[int] = None, layer_weights: List[float] = None,
                 center: Optional[np.ndarray] = None, **kwargs):
        self.num_quantizers = num_quantizers
        self.dim = dim
        self.model = model
        self.id = id
        if layers is None:
            layers = [num_quantizers - 1]
        if layer_weights is None:
            layer_weights = np.ones(num_quantizers)
        if center is None:
            center = np.zeros((num_quantizers,))
        self.layers = layers
        self.layer_weights = torch.tensor(layer_weights)
        self.center = torch.tensor(center).int()
        if explanations is None:
            explanations = ["" for _ in layers]
        if scores is None:
            scores = [{} for _ in layers]
        if activations is None:
            activations = {k: [] for k in layers}
        self.explanations = explanations
        self.scores = scores
        self.activations = activations
        self.rqae = None
        self._prepared = None   # (M, table, w, wcum) on the model's device

    def to_feature(self, layer: int = 0):
        return Feature(id=self.id, model=self.model, explanation=self.explanations[layer],
                       scores=self.scores[layer], activations=self.activations[self.layers[layer]])

    def load_model(self, rqae):
        self.rqae = rqae
        if rqae.codebook.is_cuda:    # RQAE.layer_norms: one kernel instead of a sync per level
            self.layer_weights = rqae.layer_norms.to(torch.float16).cpu()
        else:
            self.layer_weights = torch.tensor(
                [l[1].weight.data.norm(dim=0).mean().item() for l in rqae.layers]).to(torch.float16)
        self._prepared = None
        return self

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_prepared"] = None   # device cache, rebuilt on first use (the object crosses RPC, scripts/3:190-192)
        return state

    def intensity(self, token_indices: torch.Tensor, layers=None):
        """(..., num_quantizers) code indices -> (..., len(layers)) fp16 intensities (rqae/feature.py:102-129)."""
        if layers is None:
            layers = self.layers
        if self.rqae is None:
            raise ValueError("Model not loaded. Needed for intensity calculation.")
        layers = [int(l) for l in layers]
        M = max(layers) + 1
        sims = self.rqae.codebook_sims
        if not sims.is_cuda:
            raise RuntimeError("rqae_amd.RQAEFeature.intensity needs the model on a ROCm device (no CPU path)")
        dev = sims.device
        if token_indices.shape[-1] < M:
            raise IndexError(f"token_indices has {token_indices.shape[-1]} levels, layer {M - 1} requested")
        prep = self._prepared
        if prep is None or prep[0] != M or prep[1].device != dev:
            table, w, wcum = _native.intensity_prepare(sims.contiguous(), self.center, self.layer_weights, M)
            prep = self._prepared = (M, table, w, wcum)
        _, table, w, wcum = prep
        idx = token_indices
        if idx.dtype not in (torch.int64, torch.int32, torch.uint16):
            idx = idx.to(torch.int32)
        lead = idx.shape[:-1]
        idx2d = idx.to(dev).reshape(-1, idx.shape[-1]).contiguous()
        layers_dev = torch.tensor(layers, dtype=torch.int32, device=dev)
        out = _native.intensity(table, w, wcum, M, sims.shape[0], idx2d, layers_dev)
        return out.view(*lead, len(layers)).to(token_indices.device)

    @classmethod
    def from_quantizer(cls, quantizer, **kwargs):
        return cls(num_quantizers=quantizer.num_quantizers, dim=quantizer.codebook_dim, **kwargs).load_model(quantizer)

    def save(self, file_path: str):
        np.savez(file_path, **{k: v for k, v in self.__dict__.items() if k not in ("rqae", "_prepared")})

    @classmethod
    def load(cls, file_path: str):
        """Read back what ``save`` wrote (rqae/feature.py:142-153): every stored array that holds exactly one element is
        unwrapped to the Python object (``ndarray.item()``), anything longer stays an array; ``explanations`` comes back
        as a list of ``str``.  The model is not part of the file: call ``load_model`` before ``intensity``."""
        stored = np.load(file_path, allow_pickle=True)
        fields = {}
        for key in stored.files:
            value = stored[key]
            fields[key] = [str(text) for text in value] if key == "explanations" and value.ndim > 0 else value
            if value.size == 1:
                fields[key] = value.item()
        return cls(**fields)
