"""CPU oracle for the RQAE hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  rqae_amd/ (the product) never does.
"""
from .oracle import (  # noqa: F401
    build,
    forward,
    adjudicate,
    decode,
    quantize,
    max_threads,
    gemma_norm,
    gemma_denorm,
    hook_reference,
    gemma_norm_canonical,
    gemma_denorm_canonical,
    intensity,
    codebook_sims,
    subfeatures,
    subfeature_sims,
    layer_norms,
)
