"""Forward of a named BASELINE config on both arms, with the matrix-core arm's per-kernel times.

    python tools/aux_bench/config_bench.py C5 65536 [--levels N] [--dtype bf16|f32] [--free-codebooks]

--free-codebooks: the config's shape with one learned codebook per level (quantization_method "vq", K rows each) instead of the
shared FSQ lattice.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rqae_amd import RQAE, _native, synth  # noqa: E402


def timed(fn, reps=5):
    with torch.inference_mode():
        fn(); fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("tokens", type=int)
ap.add_argument("--levels", type=int, default=0)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--free-codebooks", action="store_true")
ap.add_argument("--recon", type=int, default=-1, help="rqae_hip_gram_options.recon (3 = fp16 two-piece kernel, 2 / 1 = round 3's kernels)")
ap.add_argument("--proj", type=int, default=-1)
ap.add_argument("--lean", type=int, default=-1)
ap.add_argument("--arms", default="exact,gram")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
cfg = dict(synth.CONFIGS[a.config])
if a.levels:
    cfg["num_quantizers"] = a.levels
if a.free_codebooks:
    rows = cfg["codebook_size"] ** cfg["codebook_dim"]
    cfg.update(codebook_size=rows, quantization_method="vq")
    w = synth.weights(1, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], free_codebook_rows=rows)
else:
    w = synth.weights(1, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
shared = not a.free_codebooks
out = {"config": a.config, **cfg, "tokens": a.tokens, "dtype": a.dtype}
x = torch.randn(1, a.tokens, cfg["dim"], device="cuda")
x = (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16 if a.dtype == "bf16" else torch.float32)
opts = None
if a.recon >= 0 or a.proj >= 0 or a.lean >= 0:
    opts = _native.gram_options(recon=a.recon, proj=a.proj, lean_levels=a.lean if a.lean >= 0 else 5)
    out["options"] = {"recon": a.recon, "proj": a.proj, "lean_levels": a.lean}
for arm in a.arms.split(","):
    m = RQAE(**cfg, arm=arm).eval()
    m.load_state_dict(synth.state_dict_from_weights(w))
    m = m.to("cuda")
    m.gram_options = opts
    K = m.codebook.shape[1]
    if arm == "gram" and not _native.gram_supported(m._dims(), shared):
        out["gram"] = "not served"
        continue
    ms = timed(lambda: m(x), a.reps)
    out[arm] = {"ms": round(ms, 3), "Mtok_per_s": round(a.tokens / ms / 1e3, 2)}
    if arm == "exact":
        out[arm]["kernel"] = _native.kernel_info(cfg["dim"], cfg["codebook_dim"], K, cfg["num_quantizers"])
    else:
        if not shared:
            continue
        ph = _native.encode_gram_phases(m.gram_packed_weights(), m._dims(), x.view(-1, cfg["dim"]), cfg["num_quantizers"], options=opts)
        out[arm]["kernel_ms"] = {"project": round(ph[0], 3), "levels": round(ph[1], 3), "reconstruct": round(ph[2], 3)}
print(json.dumps(out))
