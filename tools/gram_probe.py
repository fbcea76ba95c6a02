#!/usr/bin/env python3
"""Measurement tool: the matrix-core ("gram") arm of forward against the exact arm on the same inputs.

    python tools/gram_probe.py [--tokens N] [--shapes C2,P,C1] [--dtypes bf16,f32,f16] [--reps R] [--adjudicate]

Prints, per shape and input dtype: tokens whose codes differ between the two arms, the largest relative
reconstruction difference on tokens with identical codes, and the time per launch of each arm (HIP events on the
launching stream).  --adjudicate runs the CPU oracle (test infrastructure) on the differing tokens and prints
their top-1/top-2 margins.  Not part of the product path.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from rqae_amd import _native, synth  # noqa: E402

DT = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def fp64_sims(x, w, codes, level):
    """The reference level loop in float64 with the given codes up to `level`; returns that level's cosine row and q."""
    r = x.astype(np.float64)
    for l in range(level + 1):
        q = w["w_in"][l].astype(np.float64) @ r + w["b_in"][l].astype(np.float64)
        cb = w["codebook"][l].astype(np.float64)
        if l == level:
            return cb @ (q / np.linalg.norm(q)), q
        cv = q + (cb[codes[l]] - q)
        r = r - (w["w_out"][l].astype(np.float64) @ cv + w["b_out"][l].astype(np.float64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=262144)
    ap.add_argument("--shapes", default="C2")
    ap.add_argument("--dtypes", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--adjudicate", action="store_true")
    ap.add_argument("--codes-only", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for shape in args.shapes.split(","):
        cfg = synth.CONFIGS[shape]
        d, c, L = cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"]
        w = synth.weights(1234, d, c, L, cfg["codebook_size"])
        K = w["codebook"].shape[1]
        dims = (d, c, K, L)
        t = {k: torch.from_numpy(v).to(dev) for k, v in w.items()}
        packed = _native.pack(t["w_in"], t["b_in"], t["codebook"], t["w_out"], t["b_out"], True)
        if not _native.gram_supported(dims, True):
            print(json.dumps({"shape": shape, "gram": "unsupported"}))
            continue
        gpacked = _native.gram_pack(t["w_in"], t["b_in"], t["codebook"], t["w_out"], t["b_out"], True)
        n = args.tokens
        xs = np.concatenate([synth.activations(1000 + i, min(65536, n - s), d) for i, s in enumerate(range(0, n, 65536))])
        for dt in args.dtypes.split(","):
            x = torch.from_numpy(xs).to(dev).to(DT[dt]).contiguous()
            ws = torch.empty(_native.gram_workspace_bytes(dims, n, L), dtype=torch.uint8, device=dev)
            recon = not args.codes_only
            q0, i0 = _native.encode(packed, dims, x, L, want_recon=recon)
            q1, i1 = _native.encode_gram(gpacked, dims, True, x, L, want_recon=recon, workspace=ws)
            torch.cuda.synchronize()
            bad = (i0 != i1).any(dim=1)
            nbad = int(bad.sum())
            rel = None
            if recon:
                same = ~bad
                scale = q0[same].abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
                rel = float(((q1[same] - q0[same]).abs() / scale).max())
            t_exact = timed(lambda: _native.encode(packed, dims, x, L, want_recon=recon), max(2, args.reps // 4))
            t_gram = timed(lambda: _native.encode_gram(gpacked, dims, True, x, L, want_recon=recon, workspace=ws), args.reps)
            rec = {"shape": shape, "dtype": dt, "tokens": n, "codes_differ_tokens": nbad, "recon_max_rel_rowmax": rel,
                   "ms_exact": t_exact, "ms_gram": t_gram, "Mtok_s_gram": n / t_gram / 1e3}
            if args.adjudicate and nbad:
                import oracle

                rows = torch.nonzero(bad).flatten()[:64].cpu().numpy()
                xr = x[rows].float().cpu().numpy()
                _, io, mg = oracle.forward(xr, w, want_margins=True)
                i0c, i1c = i0[rows].cpu().numpy(), i1[rows].cpu().numpy()
                lv = []
                for k, r in enumerate(rows):
                    l1 = int(np.argmax(i0c[k] != i1c[k]))
                    s64, q64 = fp64_sims(xr[k], w, i0c[k], l1)
                    lv.append(dict(row=int(r), level=l1, oracle_margin_at_level=float(mg[k, l1]), exact_eq_oracle=bool((io[k] == i0c[k]).all()),
                                   exact=int(i0c[k, l1]), gram=int(i1c[k, l1]), fp64_argmax=int(np.argmax(s64)),
                                   fp64_sim_exact=float(s64[i0c[k, l1]]), fp64_sim_gram=float(s64[i1c[k, l1]]),
                                   qnorm=float(np.linalg.norm(q64))))
                rec["differing"] = lv
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
