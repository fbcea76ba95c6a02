import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {k: z[k] for k in z.files}
    if "meta" in out:
        out["meta"] = json.loads(str(out["meta"]))
    return out


def case_inputs(g):
    """Regenerate the inputs of a forward fixture from its seeds (and check the guard hash)."""
    from rqae_amd import synth

    meta = g["meta"]
    cfg = meta["cfg"]
    w = synth.weights(meta["seed_w"], cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    x = synth.activations(meta["seed_x"], meta["n"], cfg["dim"])
    if meta.get("bf16"):
        x = synth.round_to_bf16(x)
    assert synth.sha256_of(x) == str(g["x_sha256"]), "synth.activations is not bit-stable on this host"
    assert synth.sha256_of(w["codebook"]) == str(g["codebook_sha256"]), "FSQ codebook differs on this host"
    return cfg, w, x


def hook_4x128_inputs():
    """The production-shaped hook capture (tests/golden/make_golden_r4.py: B=4, S=128, scripts/1_create_activations.py:152): the fp16
    hidden states regenerate from their seed (sha256 guard); returns (fixture, hs [4,128,d] fp16)."""
    from rqae_amd import synth

    h = load_golden("hook_c2_4x128")
    m = h["meta"]
    cfg = synth.CONFIGS[m["cfgname"]]
    hs = (synth.activations(m["seed_hs"], m["B"] * m["S"], cfg["dim"]) * m["hs_scale"]).astype(np.float16).reshape(m["B"], m["S"], cfg["dim"])
    assert synth.sha256_of(hs) == str(h["hs_sha256"]), "synth.activations is not bit-stable on this host"
    return h, hs


FORWARD_CASES = ["fwd_c1", "fwd_c2", "fwd_c2_bf16", "fwd_c2_max3", "fwd_c4", "fwd_c5", "fwd_p", "fwd_ragged", "fwd_p128", "fwd_p1024"]


def rel_err(a, b):
    """max |a-b| relative to the largest reference magnitude (norm-wise fp32 tolerance)."""
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(np.abs(b).max(), 1e-30))
