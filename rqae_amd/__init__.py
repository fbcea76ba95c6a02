"""rqae_amd -- MI355X-native residual-quantisation encode/decode path for RQAE.

Drop-in for ``rqae.model.RQAE`` (reference rqae/model.py): same constructor,
state-dict layout, ``forward`` / ``quantize`` / ``decode`` / ``hook`` surface.
The per-level project -> nearest-codebook-row -> reconstruct -> subtract loop
runs in hand-written gfx950 HIP kernels behind the C ABI in include/rqae_hip.h.
There is no CPU fallback: calling the hot path without the HIP library or on a
CPU tensor raises.
"""
from .model import RQAE  # noqa: F401
from .feature import Feature, RQAEFeature  # noqa: F401
from . import ops  # noqa: F401  (registers torch.ops.rqae_hip.{encode, decode, quantize})

__all__ = ["RQAE", "RQAEFeature", "Feature"]
