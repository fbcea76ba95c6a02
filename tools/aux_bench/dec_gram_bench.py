"""RQAE.decode of BASELINE configs[1]'s codes (262 144 tokens x 8 levels, d 2304): exact kernel vs the matrix-core arm's."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rqae_amd import RQAE, synth  # noqa: E402


def timed(fn, reps=20):
    with torch.inference_mode():
        fn(); fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


cfg = synth.CONFIGS["C2"]
w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
n = 262144
idx = torch.randint(0, 1024, (1, n, 8), device="cuda")
out = {"workload": f"decode, {n} tokens x 8 codes, d 2304, K 1024", "bytes_per_token": 2304 * 4 + 8 * 8}
for arm in ("exact", "gram"):
    m = RQAE(**cfg, arm=arm).eval()
    m.load_state_dict(synth.state_dict_from_weights(w))
    m = m.to("cuda")
    for name, ix in (("int64", idx), ("uint16", idx.to(torch.int32).to(torch.int16).view(torch.uint16))):
        ms = timed(lambda: m.decode(ix))
        b = 2304 * 4 + 8 * ix.element_size()
        out[f"{arm}_{name}"] = {"ms": round(ms, 4), "Mtok_per_s": round(n / ms / 1e3, 1), "hbm_GBps": round(n * b / ms / 1e6, 0),
                                "frac_of_8TBps": round(n * b / ms / 1e6 / 8000, 3)}
print(json.dumps(out))
