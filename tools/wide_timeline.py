#!/usr/bin/env python3
"""Where a wave of gram_levels_wide_kernel (C4: c 13, K 8192, depth 16) spends its cycles -- tools build with -DW_TIMELINE.

    python tools/wide_timeline.py --build-only      (anywhere)
    python tools/wide_timeline.py --no-build        (on the GPU box)
"""
import argparse
import ctypes
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=65536)
    ap.add_argument("--config", default="C4", help="C4 (c 13, K 8192, depth 16) or C5 (c 4, K 4096, depth 32)")
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--no-build", action="store_true")
    ap.add_argument("--lib", default="", help="a tools library already built (tools/_build/...)")
    args = ap.parse_args()
    import __graft_entry__ as ge
    out = os.path.join(ROOT, "tools", "_build")
    objdir = os.path.join(out, "wt_obj")
    os.makedirs(objdir, exist_ok=True)
    so = os.path.join(out, "librqae_wt.so")
    if not args.no_build:
        for f in ("rqae_hip.o", "encode_main.o", "search.o"):
            shutil.copy(os.path.join(ge.CSRC, f), os.path.join(objdir, f))
        ge.build_library(so, extra_flags=["-DW_TIMELINE", *[f"-D{m}" for m in args.define]], objdir=objdir, units=("gram.hip",))
    if args.build_only:
        return
    os.environ["RQAE_HIP_LIBRARY"] = args.lib or so
    import numpy as np
    import torch
    from rqae_amd import RQAE, _native, synth
    cfg = synth.CONFIGS[args.config]
    w = synth.weights(1, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    m = RQAE(**cfg, arm="gram").eval()
    m.load_state_dict(synth.state_dict_from_weights(w))
    m = m.to("cuda")
    n = args.tokens
    x = torch.randn(1, n, cfg["dim"], device="cuda")
    x = (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16)
    with torch.inference_mode():
        for _ in range(5):
            m(x)
    torch.cuda.synchronize()
    ph = _native.encode_gram_phases(m.gram_packed_weights(), m._dims(), x.view(n, -1), cfg["num_quantizers"])
    print("kernel ms (project, levels, reconstruct):", [round(v, 3) for v in ph])
    nw = (n + 63) // 64
    buf = (ctypes.c_ulonglong * (nw * 8))()
    rc = _native.lib().rqae_hip_debug_wide_stamps(buf, nw * 8)
    assert rc == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 8).astype(np.float64)
    a = a[a[:, 0] > 0]
    names = ["total", "P load", "B rows", "scan", "hand-over", "rescoring", "straight-through + Gram", "later groups"]
    tot = a[:, 0].mean()
    for i, nm in enumerate(names):
        print(f"  {nm:26s} {a[:, i].mean():12.0f} cycles per wave  ({100 * a[:, i].mean() / tot:5.1f} %)   min {a[:, i].min():.0f} max {a[:, i].max():.0f}")
    K, c = cfg["codebook_size"] ** cfg["codebook_dim"], cfg["codebook_dim"]
    per_tile = (K // 16) * (3 if c > 5 else 1) * 4           # matrix instructions per level and 64-token tile
    tiles = nw / len(a)                                       # 64-token tiles per (persistent) wave
    print(f"  {len(a)} waves, {tiles:.2f} tiles each; scan per level and tile {a[:, 3].mean() / cfg['num_quantizers'] / tiles:.0f} cycles = "
          f"{a[:, 3].mean() / cfg['num_quantizers'] / tiles / per_tile:.1f} cycles per matrix instruction")


if __name__ == "__main__":
    main()
