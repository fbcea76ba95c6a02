#!/usr/bin/env python3
"""Timeline of the level chain (gram_levels2_kernel) from s_memtime stamps -- tools build with -DRQAE_LEV_TIMELINE; run ON the GPU box.

    python tools/lev_timeline.py [--tokens 262144] [--define MACRO]...

Per wave: kernel entry, tables staged, projections loaded, end of each level, end of the tile, plus s_memrealtime at both ends
(the in-kernel clock) and the XCC / hardware ids.  Prints where a wave's time goes, how far apart the waves start and end, and
what fraction of the kernel's span the average wave is alive for.
"""
import argparse
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "rqae_amd", "csrc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=262144)
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--lean", type=int, default=2)
    ap.add_argument("--build-only", action="store_true", help="compile the tools library (works without a GPU) and stop")
    ap.add_argument("--no-build", action="store_true", help="use the tools library already in tools/_build")
    args = ap.parse_args()
    import __graft_entry__ as ge
    out = os.path.join(ROOT, "tools", "_build")
    objdir = os.path.join(out, "lt_obj")
    os.makedirs(objdir, exist_ok=True)
    so = os.path.join(out, "librqae_hip_lt.so")
    if not args.no_build:
        for f in ("rqae_hip.o", "encode_main.o", "search.o"):
            shutil.copy(os.path.join(CSRC, f), os.path.join(objdir, f))
        ge.build_library(so, extra_flags=["-DRQAE_LEV_TIMELINE", "-DRQAE_GRAM_TOOLS", *[f"-D{m}" for m in args.define]], objdir=objdir, units=("gram.hip",))
    if args.build_only:
        return
    import numpy as np
    import torch
    from rqae_amd import _native, synth
    _native.SO_PATH = so
    cfg = synth.CONFIGS["C2"]
    d, c, L = cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"]
    w = synth.weights(1234, d, c, L, cfg["codebook_size"])
    K = w["codebook"].shape[1]
    dims = (d, c, K, L)
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in w.items()}
    gp = _native.gram_pack(t["w_in"], t["b_in"], t["codebook"], t["w_out"], t["b_out"], True)
    n = args.tokens
    x = torch.randn(n, d, device=dev).to(torch.bfloat16)
    ws = torch.zeros(_native.gram_workspace_bytes(dims, n, L), dtype=torch.uint8, device=dev)
    opts = _native.gram_options(lean_levels=args.lean)
    for _ in range(200):      # clocks settled
        _native.encode_gram(gp, dims, True, x, L, workspace=ws, options=opts)
    torch.cuda.synchronize()
    ph = [_native.encode_gram_phases(gp, dims, x, L, options=opts) for _ in range(5)]
    print("kernel ms (project, levels, reconstruct):", [round(float(np.median(v)), 4) for v in zip(*ph)])
    _native.encode_gram(gp, dims, True, x, L, workspace=ws, options=opts)
    torch.cuda.synchronize()
    raw = ws[-(1 << 20):].cpu().numpy().view(np.uint64).astype(np.int64)
    nwaves = (n + 63) // 64
    st = raw[: nwaves * 16].reshape(nwaves, 16)
    st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    span = st[:, 11].max() - t0
    life = st[:, 11] - st[:, 0]
    clk = (st[:, 11] - st[:, 0]) / ((st[:, 13] - st[:, 12]) * 10e-9) / 1e9
    print(f"{len(st)} waves; kernel span (first entry -> last end) {span} cycles; wave lifetime mean {life.mean():.0f} min {life.min()} max {life.max()}")
    print(f"in-kernel clock (s_memtime / s_memrealtime): median {np.median(clk):.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f})")
    print(f"entry after the first: mean {np.mean(st[:,0]-t0):.0f} p50 {np.percentile(st[:,0]-t0,50):.0f} p90 {np.percentile(st[:,0]-t0,90):.0f} max {np.max(st[:,0]-t0)}")
    print(f"end before the last:   mean {np.mean(t0+span-st[:,11]):.0f} p50 {np.percentile(t0+span-st[:,11],50):.0f} p90 {np.percentile(t0+span-st[:,11],90):.0f}")
    names = ["tables", "P load"] + [f"lvl{i}" for i in range(8)] + ["stores"]
    seg = np.stack([st[:, i + 1] - st[:, i] for i in range(11)], axis=1)
    print("mean cycles per segment: " + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, seg.mean(axis=0))))
    print("p90  cycles per segment: " + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, np.percentile(seg, 90, axis=0))))
    xcc = st[:, 14] & 15
    for xx in sorted(set(xcc.tolist())):
        m = xcc == xx
        print(f"  XCC {xx}: {m.sum()} waves, entry {np.mean(st[m,0]-t0):.0f}, lifetime {life[m].mean():.0f}, end {np.mean(st[m,11]-t0):.0f}")


if __name__ == "__main__":
    main()
