#!/usr/bin/env python3
"""The packed-fp32 mis-scoring of round 2, one controlled run: every scan form of the level chain -- 16 waves per
workgroup (128 registers per lane: the form that spills with the SLP vectoriser on), 8 waves (256 registers: no spills), the first
form -- on a build of gram.hip WITH the SLP vectoriser (tools/_build/slp, RQAE_HIP_LIBRARY) and on the product build, against the
exact arm.  Prints per form: launches with a differing token that is not low-margin, and the lanes (token % 64) of such tokens.

    RQAE_HIP_LIBRARY=tools/_build/slp/librqae_hip.so python tools/slp_experiment.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (a tool, not the product)
from rqae_amd import RQAE, _native, synth  # noqa: E402

cfg = synth.CONFIGS["C2"]
w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
models = {}
for arm in ("gram", "exact"):
    m = RQAE(**cfg, arm=arm).eval()
    m.load_state_dict(synth.state_dict_from_weights(w))
    models[arm] = m.cuda()
xall = torch.from_numpy(synth.activations(7, 65536, cfg["dim"])).cuda()
_, _, mg_all = oracle.forward(xall.cpu().numpy(), w, want_margins=True)
low = mg_all.min(axis=1) < 1e-5
out = {"library": os.environ.get("RQAE_HIP_LIBRARY", "product")}
for lean, name in ((2, "scan, 16 waves x 128 registers"), (1, "scan, 8 waves x 256 registers"), (0, "scan, first form")):
    models["gram"].gram_options = _native.gram_options(lean_levels=lean)
    res = {}
    for n, reps in ((4096, 30), (65536, 6)):
        x = xall[:n].view(1, n, -1)
        bad_launches, lanes = 0, []
        with torch.inference_mode():
            _, ie = models["exact"](x)
            for _ in range(reps):
                _, ig = models["gram"](x)
                bad = torch.nonzero((ig[0] != ie[0]).any(dim=1)).flatten().cpu().numpy()
                hard = bad[~low[bad]]
                if len(hard):
                    bad_launches += 1
                    lanes += [int(t % 64) for t in hard]
        res[f"{n} tokens x {reps} launches"] = {"launches_with_a_wrong_token": bad_launches, "wrong_tokens": len(lanes),
                                                "lanes": sorted(set(lanes))}
    out[name] = res
print(json.dumps(out, indent=1))
