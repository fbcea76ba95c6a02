#!/usr/bin/env python3
"""Phase timeline of the fused matrix-core kernel (tools build with -DRQAE_FUSED_TIMELINE; run ON the GPU box).

    python3 tools/fused_timeline.py [--tokens N]

Rebuilds gram.hip with the stamp macro into a scratch copy of the library, runs one forward and prints, for the first
workgroups / waves / tiles, the s_memtime distance between the phase boundaries (shader cycles):
A = x stream + projection, b1 = barrier after A, T = transpose to one token per lane, B = level chain, V = cv pieces to
B operands, b2 = barrier, C = reconstruction + stores."""
import argparse, ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "rqae_amd", "csrc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=262144)
    ap.add_argument("--define", action="append", default=[], help="extra -D macros of the tools build (FQ_NOW, FQ_NOX, ...)")
    args = ap.parse_args()
    import __graft_entry__ as ge
    out = os.path.join(ROOT, "tools", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "librqae_hip_tl.so")
    objdir = os.path.join(out, "tl_obj")
    os.makedirs(objdir, exist_ok=True)
    for f in ("rqae_hip.o", "encode_main.o", "search.o"):       # unchanged units: the product build's objects
        shutil.copy(os.path.join(CSRC, f), os.path.join(objdir, f))
    ge.build_library(so, extra_flags=["-DRQAE_FUSED_TIMELINE", "-DRQAE_GRAM_TOOLS", *[f"-D{m}" for m in args.define]], objdir=objdir,
                     units=("gram.hip",))
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
    nbytes = _native.gram_workspace_bytes(dims, n, L)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    for _ in range(3):
        _native.encode_gram(gp, dims, True, x, L, workspace=ws)
    torch.cuda.synchronize()
    raw = ws[-((2048 + 8 * 36 * 4) * 8):].cpu().numpy().view(np.uint64).astype(np.int64)
    st = raw[:2048].reshape(8, 8, 4, 8)
    ca = raw[2048:].reshape(8, 36, 4)
    names = ["A", "b1", "T", "B", "V", "b2", "C"]
    print("cycles per phase (wg, wave, tile): " + " ".join(f"{k:>7s}" for k in names) + "   total")
    for wg in (0, 1, 5):
        for wave in (0, 3, 7):
            for tile in range(4):
                s = st[wg, wave, tile]
                if s[0] == 0:
                    continue
                dd = [int(s[i + 1] - s[i]) for i in range(7)]
                print(f"   wg {wg} wave {wave} tile {tile}:            " + " ".join(f"{v:7d}" for v in dd) + f"  {int(s[7] - s[0]):7d}")
    tot = st[:, :, :, 7] - st[:, :, :, 0]
    ok = st[:, :, :, 0] != 0
    mean = [float(((st[:, :, :, i + 1] - st[:, :, :, i])[ok]).mean()) for i in range(7)]
    print("mean over recorded waves:              " + " ".join(f"{v:7.0f}" for v in mean) + f"  {float(tot[ok].mean()):7.0f}")
    print("phase A of workgroup 0, tile 1, per chunk: wait+barrier | DMA issue | multiply (cycles; wave 0 and wave 5)")
    for wv in (0, 5):
        rows = []
        for sidx in range(0, 35):
            a = ca[wv, sidx]
            nxt = ca[wv, sidx + 1][0]
            rows.append((int(a[2] - a[0]), int(a[3] - a[2]), int(nxt - a[3])))
        print(f"  wave {wv}: " + " ".join(f"{r[0]}|{r[1]}|{r[2]}" for r in rows[:12]))
        arr = np.array(rows[2:])
        print(f"  wave {wv} mean over chunks 2..34: wait+barrier {arr[:,0].mean():.0f} dma-issue {arr[:,1].mean():.0f} multiply {arr[:,2].mean():.0f}")
    print("phase B, level 1 of workgroup 0 tile 1 (cycles): pieces+exchange | scan | hand-over | rescoring | straight-through | Gram updates")
    for wv in (0, 3, 5):
        a, b = ca[wv, 30], ca[wv, 31]
        print(f"  wave {wv}: {int(a[1]-a[0])} | {int(a[2]-a[1])} | {int(a[3]-a[2])} | {int(b[0]-a[3])} | {int(b[1]-b[0])} | {int(b[2]-b[1])}")
    t0 = st[:, :, 0, 0][ok[:, :, 0]].min()
    print("first tile starts (cycles after the earliest):", sorted(set(int(v - t0) for v in st[:, 0, 0, 0] if v)))


if __name__ == "__main__":
    main()
