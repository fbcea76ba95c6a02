"""Round-4 additions to the reference captures: run ONLY in the build container, like make_golden.py
(imports /root/reference/rqae/model.py; stores outputs, never source).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r4.py

* fwd_c4        -- C4's shape at depth 4 on 1024 tokens ("C4-small (N=1024, L=4)"; it was 512)
* fwd_p128      -- the released shape (d 2304, c 4, n 5) at depth 128 on 512 tokens: deep-level accumulation pinned by the reference itself
* fwd_p1024     -- the released shape at its full depth 1024 (rqae/model.py:18-24) on 128 tokens
* hook_c2_4x128 -- RQAE.hook's own capture at the production call shape B=4, S=128 (scripts/1_create_activations.py:152): every code,
                   and normed / quantized / new / patched on sampled token rows (the inputs regenerate from seeds)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts the repo root and /root/reference on the path, imports the reference)
from rqae_amd import synth  # noqa: E402


def hook_case_4x128():
    cfg = dict(synth.CONFIGS["C2"])
    w = synth.weights(mg.SEED_W, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    m = mg.ref_model(cfg, w)
    B, S, d = 4, 128, cfg["dim"]
    rng = np.random.Generator(np.random.PCG64(4243))
    hs = (synth.activations(556, B * S, d) * 3.0).astype(np.float16).reshape(B, S, d)
    nw = (0.1 * rng.standard_normal(d)).astype(np.float32)
    llm = mg.FakeGemma(torch.from_numpy(nw))
    store = {}
    out0 = torch.from_numpy(hs.copy())
    fn = m.hook(llm=llm, store=lambda k, v: store.update({k: v}))
    with torch.inference_mode():
        fn(None, None, (out0,))
    rows = mg.sample_rows(B * S)
    rows = np.unique(np.concatenate([rows, np.arange(0, B * S, S), np.arange(1, B * S, S)]))     # + every BOS row and the row after it
    flat = lambda t: t.reshape(B * S, -1).numpy()[rows]      # noqa: E731
    np.savez_compressed(
        os.path.join(HERE, "hook_c2_4x128.npz"),
        meta=json.dumps(dict(cfgname="C2", B=B, S=S, seed_w=mg.SEED_W, seed_hs=556, hs_scale=3.0, seed_nw=4243)),
        hs_sha256=synth.sha256_of(hs), norm_weight=nw, rows=rows,
        indices=store["indices"].numpy().astype(np.uint16),
        normed_rows=flat(store["normed"]), quantized_rows=flat(store["quantized"]), new_rows=flat(store["new"]),
        patched_rows=flat(out0))
    print("hook_c2_4x128 written", os.path.getsize(os.path.join(HERE, "hook_c2_4x128.npz")) // 1024, "KB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    mg.forward_case("fwd_c4", "C4", 1024, 1003, levels=4)
    mg.forward_case("fwd_p128", "P", 512, 1008, levels=128)
    mg.forward_case("fwd_p1024", "P", 128, 1009, levels=1024)
    hook_case_4x128()
