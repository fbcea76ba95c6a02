#!/usr/bin/env python3
"""Benchmark of the RQAE encode hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--tokens T] [--arm gram|exact|auto] [--codes-only]
                    [--spinup-ms MS] [--no-extra-legs] [--no-parity]

A "step" is one RQAE.forward (codes + reconstruction, depth 8) over one batch of synthetic Gemma-2-2B-shaped
activations already resident in HBM.  N = 1: BASELINE.json configs[1] -- 262144 tokens x d=2304 in bf16, K=1024 (c=5),
depth 8.  N > 1: configs[2]'s sharding -- every rank encodes its own 524288-token batch (4M tokens over 8 GPUs), weights
replicated, no collective on the data path: weak scaling.  `python bench.py --gpus N` launches its own ranks (one
process per GPU, RCCL process group for the barrier and the MAX over ranks); under torch.distributed.run it uses the
ranks it is given.  Rank 0 prints ONE JSON line.

What the line holds besides the contract's fields:
  * `spinup_ms` / `spinup_steps`: an untimed, disclosed run of the same step BEFORE the W warm-up steps (clocks and caches
    settle; the K timed steps are still exactly K, bracketed by barrier + synchronize);
  * `step_ms`: first / min / median / max of the per-step HIP-event times inside the timed region;
  * `legs`: the same measurement protocol repeated in-process at other batch sizes -- 524288 tokens per GPU (configs[2]'s
    shard) and, at N = 1, north_star's 1048576 tokens (`roofline.at_1M`) -- so 1 -> N ratios can be taken at equal per-GPU work;
  * `parity`: after the timed region, a slice of the timed batch through the CPU oracle (codes compared token by token);
  * `roofline.traffic_source`: where `traffic` comes from (a committed rocprofv3 --pmc pass of the same sources, replayed).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = dict(dim=2304, codebook_dim=5, codebook_size=4, num_quantizers=8)  # K = 4**5 = 1024


def workload_name(tokens, world):
    """config.workload for the batch actually timed (BASELINE.json configs[1] / configs[2] when the sizes are theirs)."""
    shape = "d=2304, K=1024, depth=8, bf16 in"
    if world == 1 and tokens == 262144:
        return f"C2: Gemma-2-2B layer-12 residual stream (synthetic), 256k tokens, {shape}"
    if world > 1 and tokens == 524288:
        return (f"C3: same d/K/depth as C2, 512k tokens per GPU token-sharded ({tokens * world // 1048576}M over {world} GPUs), "
                "codebooks replicated, no collective")
    return (f"C2 shape at a non-default batch: Gemma-2-2B layer-12 residual stream (synthetic), {tokens} tokens per GPU x {world} GPU(s), "
            f"{shape}")


# algorithmic work per token
D, C, K, L = 2304, 5, 1024, 8
BYTES_PER_TOKEN = D * 2 + D * 4 + L * 8                      # bf16 x in, fp32 recon + int64 codes out = 13888
BYTES_PER_TOKEN_CODES = D * 2 + L * 8                        # 4672
FLOPS_PER_TOKEN = L * (2 * D * C + 2 * C * K + 2 * C * D + 2 * D)  # 487424, SURVEY 8(d)
# codes only on the exact arm: no `out += y`, and the last level needs neither lin_out nor `r -= y`     = 443648
FLOPS_PER_TOKEN_CODES_EXACT = L * (2 * D * C + 2 * C * K) + (L - 1) * (2 * C * D + D)
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (about 6.3 TB/s is achievable)
PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 (vector = fp32 MFMA) peak
PEAK_BF16_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
# matrix-core arm: matrix-instruction flops actually issued per token at C2: projection W_in as two fp16 pieces
# x one x piece over 64 output columns; no scan on the FSQ lattice; reconstruction 48 values x 3 piece products
GRAM_MFMA_FLOPS_PER_TOKEN = 2 * D * 64 * 2 + 2 * D * 48 * 3

PROFILE_ROUND = "r04"     # profiles/<round>_traffic.json, <round>_pmc_summary.json: the committed counter passes of this round

KERNEL_SOURCES = ["gram.hip", "gram_fused.hpp", "gram_stream.hpp", "gram_wide.hpp", "gram_recon3.hpp", "gram_small.hpp", "gram_pieces.hpp", "recon3.hip", "gram_proj3.hpp", "proj3.hip", "gram_deep3.hpp", "deep3.hip", "gram_sign3.hpp", "sign3.hip", "rqae_hip.hip", "encode_main.hip", "encode_lds.hpp", "rqae_common.hpp"]


def sources_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "rqae_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def synth_batch(torch, n_tokens, device, seed, d=None, dtype=None):
    """Gemma-like residual rows generated ON DEVICE (no H2D in the bench): N(0,1) x heavy-tailed
    channel scale, 4 massive channels, RMS-normalised, x (1+w); rounded to bf16 (or kept fp32)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    d = d or CFG["dim"]
    dtype = dtype or torch.bfloat16
    scale = 0.3 + torch.randn(d, device=device, generator=g) ** 2
    scale[torch.randperm(d, device=device, generator=g)[:4]] *= 30.0
    w = 0.1 * torch.randn(d, device=device, generator=g)
    x = torch.empty(n_tokens, d, device=device, dtype=dtype)
    step = 32768
    for s in range(0, n_tokens, step):
        z = torch.randn(min(step, n_tokens - s), d, device=device, generator=g) * scale
        z = z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + 1e-6) * (1.0 + w)
        x[s:s + z.shape[0]] = z.to(dtype)
    return x


def cpu_baseline_reference_ops(w, n=32768):
    """SURVEY 8(d)'s CPU baseline: the reference's operator sequence (rqae/model.py:199-230 -- per level addmm, norm, div,
    [N,K] matmul, argmax, gather, straight-through sum, addmm, sub, add) through PyTorch-CPU ops on all host cores, ONE pass over
    N = 32768 tokens of the same workload (oracle/torch_ops.py: bit-equal to the reference captures, tests/test_oracle_golden.py)."""
    import torch
    from oracle import torch_ops
    from rqae_amd import synth

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    x = synth.round_to_bf16(synth.activations(1001, n, CFG["dim"]))
    torch_ops.forward(x[:2048], w)        # warm: thread pool, allocator
    t0 = time.perf_counter()
    torch_ops.forward(x, w)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "reference-ops",
            "sample": f"1 x {n} tokens of the same workload (bf16-rounded synthetic rows as fp32, d=2304 K=1024 depth=8, codes + "
                      f"reconstruction) through the reference's own PyTorch-CPU operator sequence (oracle/torch_ops.py restates "
                      f"rqae/model.py:199-230 op for op; torch.set_num_threads({cores}) = os.cpu_count()), {dt:.1f} s"}


def cpu_baseline(w, target_seconds=8.0):
    """The CPU oracle (a port of the reference's math, pinned to its golden vectors) timed on
    this host's cores on a bounded sample of the same workload (~10 s of CPU work)."""
    import oracle
    from rqae_amd import synth

    threads = oracle.max_threads()
    n = 32768
    x = synth.round_to_bf16(synth.activations(1001, n, CFG["dim"]))
    oracle.forward(x[:1024], w, threads=threads)  # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.forward(x, w, threads=threads)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or reps >= 200:
            break
    return {"value": n * reps / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x {n} tokens of the same workload (bf16-rounded synthetic rows, d=2304 K=1024 depth=8, "
                      f"codes + reconstruction), oracle/rqae_oracle.c (C restatement, {threads} OpenMP threads, {dt:.1f} s); "
                      "the reference's own torch-CPU path ran ~40x slower than this port in the survey container "
                      "(BASELINE.md section 2: 4.3 k tok/s on 8 threads)"}


def parity_check(np, torch, model, x, w, gram, codes_only, n_check=65536):
    """After the timed region: the first `n_check` tokens of the timed batch through the product path once more and through the
    CPU oracle (the checker, never the thing measured).  Exact arm: every code must equal the oracle's.  Matrix-core arm:
    SURVEY 8(c) Tier 2 -- codes equal wherever the oracle's top-1/top-2 margin is >= 1e-5; tokens that differ are counted
    (`listed_low_margin`), and none may differ at a larger margin (`mismatch_at_margin_ge_1e-5` must be 0)."""
    import oracle

    n = min(n_check, x.shape[1])
    xs = x[:, :n]
    with torch.inference_mode():
        if codes_only:
            idx, q = model.encode(xs), None
        else:
            q, idx = model(xs)
    torch.cuda.synchronize()
    xf = xs.float().cpu().numpy().reshape(n, -1)
    q_o, idx_o, mg = oracle.forward(xf, w, want_margins=True)
    idx = idx.cpu().numpy().reshape(n, -1)
    differ = (idx != idx_o).any(axis=1)
    bad = differ & (mg.min(axis=1) >= 1e-5)
    out = {"tokens_checked": int(n), "arm": "gram" if gram else "exact", "tokens_with_other_codes": int(differ.sum()),
           "listed_low_margin": int((differ & ~bad).sum()), "mismatch_at_margin_ge_1e-5": int(bad.sum()),
           "oracle": "oracle/rqae_oracle.c (pinned to captures of the reference, tests/test_oracle_golden.py)"}
    if q is not None:
        same = ~differ
        qg = q.cpu().numpy().reshape(n, -1)
        rel = np.abs(qg[same] - q_o[same]).max(axis=1) / np.abs(q_o[same]).max(axis=1)
        out["recon_max_rel_err"] = float(rel.max()) if rel.size else 0.0
    out["ok"] = bool(out["mismatch_at_margin_ge_1e-5"] == 0 and (gram or out["tokens_with_other_codes"] == 0)
                     and out.get("recon_max_rel_err", 0.0) < 1e-5)
    return out


# ---- the other BASELINE configs, timed in the same process after the C2 legs ---------------------------
# (name, synth config, tokens, x dtype, what it is)
CONFIG_LEGS = [
    ("C4/bf16", "C4", 65536, "bf16", "configs[3]: d=3584, K=8192 (c=13), depth=16, 65536 tokens, bf16 in"),
    ("C4/f32", "C4", 65536, "f32", "configs[3] with fp32 activations"),
    ("C5/bf16", "C5", 262144, "bf16", "configs[4]'s shape: d=2304, K=4096 (c=4), depth=32, 262144 tokens, bf16 in (fp32-accurate operands; "
                                      "the fp8 codebook was measured and rejected, DESIGN.md 12.6)"),
    ("P16/bf16", "P", 262144, "bf16", "released shape cut to depth 16: d=2304, K=625 (c=4, n=5), 262144 tokens, bf16 in"),
]


def algorithmic_per_token(d, c, K, L, x_bytes):
    """bytes = x in + fp32 reconstruction + int64 codes; flops = L (2dc + 2cK + 2cd + 2d)."""
    return d * x_bytes + d * 4 + L * 8, L * (2 * d * c + 2 * c * K + 2 * c * d + 2 * d)


def issued_mfma_flops_per_token(d, c, K, L, dtype, lattice):
    """Matrix-instruction flops the matrix-core arm issues per token (DESIGN.md sections 9, 12, 13): every fp32 operand as exact
    low-precision pieces -- projection: W_in pieces x x pieces over d x (level groups x 16-wide output tiles); the cosine scan (none on
    an FSQ lattice codebook); reconstruction over the k-slots of the straight-through values' pieces."""
    cp = 4 if c <= 4 else 5 if c == 5 else 13
    gl = 8 if cp <= 8 else 4
    gp = (gl * cp + 15) // 16 * 16
    nlg = (L + gl - 1) // gl
    # projection: gram_proj3_kernel -- W_in as two fp16 pieces x (x as one fp16 piece: 2 products; fp32 x as two: 3), output tiles of 32
    nog = nlg * gp // 16
    ot = (nog + 1) // 2
    if ot <= 8:
        otk = 1 if ot <= 1 else 2 if ot <= 2 else 4 if ot <= 4 else 8
        proj = 2 * d * otk * 32 * (3 if dtype == "f32" else 2)
    else:
        proj = 2 * d * nlg * gp * {"bf16": 3, "f16": 6, "f32": 6}[dtype]
    nkc = (6 * cp + 31) // 32
    scan = 0 if lattice else L * 2 * ((K + 15) // 16 * 16) * 32 * nkc
    # reconstruction: gram_recon3_kernel -- two fp16 pieces per value, three products, k-steps of 16 values (instantiated depths)
    ks = (L * c + 15) // 16
    ks_t = next((v for v in (2, 3, 4, 8, 13) if ks <= v), None)
    if ks_t is not None:
        recon = 2 * d * 16 * ks_t * 3
    else:
        full, rem = divmod(L, gl)
        gs = (6 * cp * gl + 31) // 32 * 32
        recon = 2 * d * (full * gs + (6 * cp * rem + 31) // 32 * 32)
    # the level chain of a two-point lattice at C4's shape (gram_sign3.hpp): 68 (level, later column tile) pairs x 3 products of 32 x 32 x 16
    chain = 68 * 3 * 2 * 32 * 16 if (lattice and c == 13 and K == 2 ** 13 and L == 16) else 0
    return proj + scan + recon + chain


def config_leg(np, torch, device, name, cfg_name, tokens, dtype, what, steps=20, warmup=3, n_check=4096, options=None,
               free_codebooks=False, arm="gram", levels=None, spinup_ms=150.0):
    """One BASELINE config through the product path: `steps` timed forwards (HIP events on the launch stream, synchronised both
    sides), the per-kernel split, both roofline fractions and a parity count against the oracle on the first `n_check` tokens.
    As for the headline leg, an untimed and disclosed spin-up of the same step (`spinup_ms`) runs before the warm-up steps: a window of
    a few ms right after another workload is 5-10 % slow (clocks, caches: DESIGN.md 12.1)."""
    import oracle
    from rqae_amd import RQAE, _native, synth

    cfg = dict(synth.CONFIGS[cfg_name])
    if levels:
        cfg["num_quantizers"] = levels
    if free_codebooks:
        rows = cfg["codebook_size"] ** cfg["codebook_dim"]
        w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], free_codebook_rows=rows)
        cfg.update(codebook_size=rows, quantization_method="vq")
    else:
        w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    model = RQAE(**cfg, arm=arm).eval()
    model.load_state_dict(synth.state_dict_from_weights(w), strict=True)
    model = model.to(device)
    if options is not None:
        model.gram_options = options
    d, c, K, L_ = model._dims()
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    x = synth_batch(torch, tokens, device, seed=4321, d=d, dtype=tdt).view(1, tokens, d)
    evs = []
    spin_steps = 0
    with torch.inference_mode():
        model(x)
        torch.cuda.synchronize()
        time.sleep(0.02)      # (the legs before this one have freed GBs: let the runtime finish with them outside the timed region)
        ts = time.perf_counter()
        while spinup_ms > 0 and (time.perf_counter() - ts) * 1e3 < spinup_ms:
            for _ in range(4):
                model(x)
            torch.cuda.synchronize()
            spin_steps += 4
        spin_s = time.perf_counter() - ts
        for _ in range(warmup):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            q, idx = model(x)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    ev_ms = [a.elapsed_time(b) for a, b in evs]
    ms = wall / steps * 1e3
    bytes_tok, flops_tok = algorithmic_per_token(d, c, K, L_, 2 if dtype == "bf16" else 4)
    tps = tokens / (ms * 1e-3)
    out = {"what": what, "tokens": tokens, "dtype_in": dtype, "steps": steps, "warmup": warmup, "spinup_ms": spin_s * 1e3 if spinup_ms > 0 else 0.0,
           "spinup_steps": spin_steps, "ms": ms,
           "ms_events": {"min": float(np.min(ev_ms)), "median": float(np.median(ev_ms)), "max": float(np.max(ev_ms))},
           "value": tps, "unit": "tokens/s", "bytes_per_token": bytes_tok, "algorithmic_flops_per_token": flops_tok}
    hbm_frac = tps * bytes_tok / 1e9 / PEAK_HBM_GBS
    roof = {"hbm_frac": hbm_frac, "hbm_GBs": tps * bytes_tok / 1e9,
            "algorithmic_flops_frac_of_bf16_peak": tps * flops_tok / 1e12 / PEAK_BF16_TFLOPS}
    shared = not free_codebooks
    if arm != "exact" and shared and L_ > 64:
        # the blocked path of the deep shapes (gram_deep3.hpp): projection 2 pieces x d x L c, cross-block corrections L^2 c^2 / 2 x 3
        # products (whole blocks of 32 levels), reconstruction L c x d x 3 products
        try:
            with torch.inference_mode():
                ph = [_native.encode_gram_phases(model.gram_packed_weights(), model._dims(), x.view(tokens, -1)[:131072], L_, options=options)
                      for _ in range(2)]
            out["kernel_ms_per_131072_tokens"] = dict(zip(["project", "level chain + corrections", "reconstruct"], [float(np.median(v)) for v in zip(*ph)]))
        except Exception as e:
            out["kernel_ms_per_131072_tokens"] = f"unavailable: {e}"
        nb = L_ // 32
        issued = 2 * d * L_ * c * 2 + 2 * 128 * 128 * (nb * (nb - 1) // 2) * 3 + 2 * d * L_ * c * 3
        roof["mfma_issued_flops_per_token"] = issued
        roof["mfma_frac"] = tps * issued / 1e12 / PEAK_BF16_TFLOPS
    elif arm != "exact" and shared:
        try:
            with torch.inference_mode():
                ph = [_native.encode_gram_phases(model.gram_packed_weights(), model._dims(), x.view(tokens, -1), L_, options=options)
                      for _ in range(3)]
            out["kernel_ms"] = dict(zip(["project", "levels", "reconstruct"], [float(np.median(v)) for v in zip(*ph)]))
        except Exception as e:
            out["kernel_ms"] = f"unavailable: {e}"
        lattice = getattr(model.gram_packed_weights(), "lattice_n", 0) >= 2 and (options is None or options.lean_levels >= 4 or options.lean_levels < 0)
        issued = issued_mfma_flops_per_token(d, c, K, L_, dtype, lattice)
        roof["mfma_issued_flops_per_token"] = issued
        roof["mfma_frac"] = tps * issued / 1e12 / PEAK_BF16_TFLOPS
    else:
        roof["mfma_frac"] = None
    fr = [(roof["hbm_frac"], "hbm"), (roof.get("mfma_frac") or 0.0, "mfma")]
    roof["frac"], roof["bound"] = max(fr)
    out["roofline"] = roof
    # parity: the first n_check tokens, product path vs oracle (SURVEY 8(c) Tier 2)
    n = min(n_check, tokens)
    if free_codebooks:
        wchk = dict(w, codebook=model._codebook_used.float().cpu().numpy()) if hasattr(model, "_codebook_used") else w
    else:
        wchk = w
    xf = x[:, :n].float().cpu().numpy().reshape(n, -1)
    q_o, idx_o, mg = oracle.forward(xf, wchk, want_margins=True)
    ig = idx[0, :n].cpu().numpy().reshape(n, -1)
    qg = q[0, :n].cpu().numpy().reshape(n, -1)
    differ = (ig != idx_o).any(axis=1)
    bad = differ & (np.nanmin(mg, axis=1) >= 1e-5)
    same = ~differ
    rel = np.abs(qg[same] - q_o[same]).max(axis=1) / np.abs(q_o[same]).max(axis=1)
    out["parity"] = {"tokens_checked": int(n), "tokens_with_other_codes": int(differ.sum()), "mismatch_at_margin_ge_1e-5": int(bad.sum()),
                     "recon_max_rel_err": float(rel.max()) if rel.size else 0.0,
                     "ok": bool(bad.sum() == 0 and (not rel.size or rel.max() < 1e-5))}
    del model, x, q, idx
    torch.cuda.empty_cache()
    return out


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N > 1 outside a launcher: one child process per GPU.  This parent never touches a GPU (no HIP call, no
    torch.cuda call) and never execs: it starts the ranks, relays rank 0's JSON line and exits with the worst code."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RQAE_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for ln in out.splitlines():   # ONE JSON line on stdout; anything else a library printed there goes to stderr
        print(ln, file=sys.stdout if ln.lstrip().startswith("{") else sys.stderr, flush=True)
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tokens", type=int, default=0, help="tokens per GPU per step (default: 262144 at N=1, 524288 at N>1)")
    ap.add_argument("--arm", choices=["gram", "exact", "auto"], default="gram",
                    help="gram = matrix-core arm (Tier-2 parity, the fast path); exact = canonical fp32 order (bit-exact)")
    ap.add_argument("--codes-only", action="store_true", help="time RQAE.encode (no reconstruction written)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spinup-burst", type=int, default=8, help="steps enqueued between two synchronisations of the spin-up")
    ap.add_argument("--spinup-ms", type=float, default=300.0,
                    help="untimed run of the same step before the warm-up steps (disclosed as spinup_ms in the line); 0 = none")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the 524288- and 1048576-token legs")
    ap.add_argument("--no-parity", action="store_true", help="skip the post-run oracle check of a 65536-token slice of the timed batch")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (the other BASELINE configs, N = 1 only)")
    ap.add_argument("--stub", action="store_true", help="CPU/gloo rehearsal of the multi-rank orchestration (tests only)")
    args = ap.parse_args()

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    tokens = args.tokens or (262144 if world == 1 else 524288)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # one rank per GPU over RCCL
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from rqae_amd import sharding

    if args.stub:
        # rehearsal of everything around the kernel: rendezvous, barrier-bracketed timing, MAX over ranks, one JSON line
        def step():
            time.sleep(0.002 * (rank + 1))

        elapsed = sharding.timed_steps(step, args.steps, args.warmup, dist.barrier if dist else (lambda: None))
        elapsed = sharding.max_over_ranks(elapsed, dist, None)
        if rank == 0:
            print(json.dumps({"metric": "stub", "value": sharding.aggregate_throughput(tokens, world, args.steps, elapsed),
                              "unit": "tokens/s", "n_gpus": world, "world_size_seen": dist.get_world_size() if dist else 1,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "config": {"tokens_per_gpu_per_step": tokens}}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    from rqae_amd import RQAE, _native, synth

    w = synth.weights(1234, CFG["dim"], CFG["codebook_dim"], CFG["num_quantizers"], CFG["codebook_size"])
    model = RQAE(**CFG, arm=args.arm).eval()
    model.load_state_dict(synth.state_dict_from_weights(w), strict=True)
    model = model.to(device)
    fn = model.encode if args.codes_only else model.forward
    gram = model._use_gram(tokens, L)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(n_tok, steps, warmup, spinup_ms):
        """The contract's protocol on one batch size: [spin-up, disclosed] W warm-up steps, barrier + synchronize, EXACTLY
        `steps` timed steps, barrier + synchronize; MAX over ranks.  Every step is also bracketed by HIP events on the stream
        the kernels are launched on."""
        x = synth_batch(torch, n_tok, device, seed=1000 + rank).view(1, n_tok, CFG["dim"])
        evs = []

        def step():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()              # same stream the kernels are launched on (torch's current stream)
            fn(x)
            b.record()
            evs.append((a, b))

        spin_steps, spin_s = 0, 0.0
        with torch.inference_mode():
            if spinup_ms > 0:
                t0 = time.perf_counter()
                while (time.perf_counter() - t0) * 1e3 < spinup_ms:
                    for _ in range(args.spinup_burst):
                        fn(x)
                    torch.cuda.synchronize()
                    spin_steps += args.spinup_burst
                spin_s = time.perf_counter() - t0
            elapsed = sharding.timed_steps(step, steps, warmup, barrier)
        ev_ms = [a.elapsed_time(b) for a, b in evs[warmup:]]
        per_rank = [elapsed]
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per_rank = [float(v.item()) for v in allt]
        elapsed = sharding.max_over_ranks(elapsed, dist, device)
        return {"x": x, "elapsed": elapsed, "per_rank": per_rank, "ev_ms": ev_ms, "spinup_steps": spin_steps,
                "spinup_ms": spin_s * 1e3}

    def leg_summary(n_tok, m, steps):
        bytes_tok = BYTES_PER_TOKEN_CODES if args.codes_only else BYTES_PER_TOKEN
        ms = m["elapsed"] / steps * 1e3
        return {"tokens_per_gpu_per_step": n_tok, "steps": steps, "ms_per_step": ms,
                "value": sharding.aggregate_throughput(n_tok, world, steps, m["elapsed"]),
                "hbm_frac_per_gpu": n_tok * bytes_tok / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "step_ms": step_stats(m["ev_ms"])}

    def step_stats(ev_ms):
        return {"first": ev_ms[0], "min": float(np.min(ev_ms)), "median": float(np.median(ev_ms)), "max": float(np.max(ev_ms)),
                "mean": float(np.mean(ev_ms))}

    main_m = measure(tokens, args.steps, args.warmup, args.spinup_ms)
    x = main_m["x"]
    elapsed, per_rank = main_m["elapsed"], main_m["per_rank"]
    kern_ms = float(np.mean(main_m["ev_ms"]))

    # ---- further legs of the same protocol (every rank takes part: barriers and the MAX are collective)
    legs = {}
    if not args.no_extra_legs and not args.tokens:
        for name, n_tok in (("262144", 262144), ("524288", 524288)) + ((("1048576", 1048576),) if world == 1 else ()):
            if n_tok == tokens:
                legs[name] = leg_summary(n_tok, main_m, args.steps)
                continue
            m = measure(n_tok, args.steps, args.warmup, 0.0)
            del m["x"]
            legs[name] = leg_summary(n_tok, m, args.steps)
        torch.cuda.empty_cache()

    if rank == 0:
        value = sharding.aggregate_throughput(tokens, world, args.steps, elapsed)
        bytes_tok = BYTES_PER_TOKEN_CODES if args.codes_only else BYTES_PER_TOKEN
        tok_per_s_kernel = tokens / (kern_ms * 1e-3)
        gbs = tok_per_s_kernel * bytes_tok / 1e9
        traffic, traffic_source = None, None
        tname = f"profiles/{PROFILE_ROUND}_traffic.json"
        tpath = os.path.join(ROOT, tname)
        if os.path.exists(tpath) and not args.codes_only:
            try:
                tj = json.load(open(tpath))
                # measured by tools/pmc.py for THIS build and THIS launch shape, or not reported
                if tj.get("tokens") == tokens and tj.get("arm") == ("gram" if gram else "exact") and tj.get("sources_sha256_16") == sources_hash():
                    traffic = tj["hbm_bytes_per_step"]
                    traffic_source = (f"{tname} (replayed: separate rocprofv3 --pmc passes of this command on these kernel sources, "
                                      "FETCH_SIZE x 2 + WRITE_SIZE per MI355X_MICROARCH.md; not measured in this run)")
            except Exception:
                traffic = None
        if traffic is None:
            traffic_source = f"none: no {tname} for these kernel sources and this launch shape"
        if gram:
            # achieved / frac follow from the contract's own clock (ms_per_step: K steps between barrier + synchronize), as VERDICT
            # round 3 asked; the mean of the per-step HIP-event times gives frac_from_step_events
            gbs_step = tokens * bytes_tok / (elapsed / args.steps) / 1e9
            roof = {"bound": "hbm", "achieved": gbs_step, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs_step / PEAK_HBM_GBS,
                    "frac_from_step_events": gbs / PEAK_HBM_GBS,
                    "traffic": traffic, "traffic_source": traffic_source,
                    "launch": "one forward = 3 kernels back to back on one stream (projection, level chain, "
                    "reconstruction)", "launch_ms": kern_ms, "bytes_per_token": bytes_tok,
                    "frac_from_ms_per_step": tokens * bytes_tok / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS,
                    "mfma_f16_TFLOPs": tok_per_s_kernel * GRAM_MFMA_FLOPS_PER_TOKEN / 1e12,
                    "mfma_f16_frac": tok_per_s_kernel * GRAM_MFMA_FLOPS_PER_TOKEN / 1e12 / PEAK_BF16_TFLOPS,
                    "algorithmic_fp32_TFLOPs": tok_per_s_kernel * FLOPS_PER_TOKEN / 1e12}
            try:    # MFMA-busy and per-kernel counters: measured by tools/pmc.py for THIS build (same source hash) or absent
                pj = json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_summary.json")))
                if traffic is not None:
                    roof["mfma_busy_frac_by_kernel"] = {k: round(v["mfma_busy_frac"], 3) for k, v in pj["kernels"].items() if "mfma_busy_frac" in v}
                    roof["mfma_busy_source"] = f"profiles/{PROFILE_ROUND}_pmc_summary.json (replayed)"
            except Exception:
                pass
            try:
                with torch.inference_mode():
                    ph = [_native.encode_gram_phases(model.gram_packed_weights(), model._dims(), x.view(tokens, -1), L,
                                                     want_recon=not args.codes_only) for _ in range(5)]
                roof["kernel_ms"] = dict(zip(["project", "levels", "reconstruct"], [float(np.median(v)) for v in zip(*ph)]))
                if not args.codes_only:
                    # the dominant kernel on its own: the reconstruction writes 4 d bytes per token (its algorithmic bytes)
                    rms = roof["kernel_ms"]["reconstruct"]
                    rg = tokens * 4.0 * D / (rms * 1e-3) / 1e9
                    roof["dominant_kernel"] = {"name": "reconstruct (gram_recon3_kernel<3, 5>)", "ms": rms, "algorithmic_bytes": tokens * 4 * D,
                                               "achieved_GBs": rg, "frac": rg / PEAK_HBM_GBS}
            except Exception as e:  # measurement extra only
                roof["kernel_ms"] = f"unavailable: {e}"
        else:
            flops_tok = FLOPS_PER_TOKEN_CODES_EXACT if args.codes_only else FLOPS_PER_TOKEN
            tflops = tok_per_s_kernel * flops_tok / 1e12
            roof = {"bound": "fp32-valu", "achieved": tflops, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": tflops / PEAK_F32_TFLOPS, "traffic": traffic, "traffic_source": traffic_source, "launch_ms": kern_ms,
                    "flops_per_token": flops_tok,
                    "flops_per_token_survey_8d": FLOPS_PER_TOKEN,
                    "frac_survey_8d": tok_per_s_kernel * FLOPS_PER_TOKEN / 1e12 / PEAK_F32_TFLOPS,
                    "hbm_achieved_GBs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS, "bytes_per_token": bytes_tok}
        if "1048576" in legs:
            roof["at_1M"] = {k: legs["1048576"][k] for k in ("tokens_per_gpu_per_step", "ms_per_step", "value", "step_ms")}
            roof["at_1M"]["frac"] = legs["1048576"]["hbm_frac_per_gpu"]
        line = {
            "metric": "tokens encoded/sec, d=2304, K=1024, depth=8",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "spinup_ms": main_m["spinup_ms"], "spinup_steps": main_m["spinup_steps"],
            "step_ms": step_stats(main_m["ev_ms"]),
            "config": {"workload": workload_name(tokens, world), "tokens_per_gpu_per_step": tokens,
                       "outputs": "codes int64" + ("" if args.codes_only else " + reconstruction fp32"),
                       "sharding": f"token-sharded x{world}, weights replicated, no collective",
                       "world_size_seen": dist.get_world_size() if dist is not None else 1,
                       "per_rank_seconds": per_rank,
                       "arm": ("gram: all-levels projection and reconstruction on v_mfma_f32_32x32x16_f16 (fp32 operands as 2 fp16 "
                               "pieces at a power-of-two scale, 3 products, fp32 accumulate), c-space level chain between them -- for this FSQ lattice codebook the "
                               "candidate rows from the signs and the order of the projection, scored in the reference's fp32 order, "
                               "no codebook scan (Tier-2 parity: codes equal wherever the top-1/top-2 margin >= 1e-5, "
                               "reconstruction <= 1e-5 rel)") if gram
                       else "exact: canonical fp32 order, bit-equal to the CPU oracle; " + _native.kernel_info(D, C, K, L)},
            "roofline": roof,
        }
        if legs:
            line["legs"] = {"note": "same protocol (W warm-up, K timed steps, barrier + synchronize, MAX over ranks) at other per-GPU "
                                    "batch sizes; `value` = whole-job tokens/s; take 1 -> N ratios at equal tokens_per_gpu_per_step",
                            **legs}
        if not args.no_parity:
            line["parity"] = parity_check(np, torch, model, x, w, gram, args.codes_only)
        if world == 1 and not args.no_configs and not args.tokens and not args.codes_only and args.arm == "gram":
            cfgs = {"note": "the other BASELINE configs through the same product path in this process, after the C2 legs: `ms` = wall time "
                            "per forward over `steps` forwards bracketed by torch.cuda.synchronize(), after an untimed spin-up of the same step (`spinup_ms`, as for the headline leg); roofline.frac = the larger of hbm_frac "
                            "(algorithmic bytes / s over 8 TB/s) and mfma_frac (matrix-instruction flops issued / s over the dense bf16 peak)"}
            del x
            torch.cuda.empty_cache()
            for name, cname, ntok, dt, what in CONFIG_LEGS:
                try:
                    cfgs[name] = config_leg(np, torch, device, name, cname, ntok, dt, what)
                except Exception as e:      # a leg that fails is reported, never hidden
                    cfgs[name] = {"what": what, "error": f"{type(e).__name__}: {e}"}
            try:
                cfgs["C2/scan"] = config_leg(np, torch, device, "C2/scan", "C2", 262144, "bf16",
                                             "configs[1]'s shape with the codebook SCANNED on the matrix cores (lean_levels=2: the path for any "
                                             "shared codebook that is not an FSQ lattice)", options=_native.gram_options(lean_levels=2))
            except Exception as e:
                cfgs["C2/scan"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                cfgs["C2/vq"] = config_leg(np, torch, device, "C2/vq", "C2", 262144, "bf16",
                                           "configs[1]'s shape with one learned codebook per level (quantization_method 'vq', renormalised in "
                                           "place once per level as rqae/model.py:126-131,196 does)", free_codebooks=True)
            except Exception as e:
                cfgs["C2/vq"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                cfgs["P1024/bf16"] = config_leg(np, torch, device, "P1024/bf16", "P", 262144, "bf16",
                                                "the released model's shape at its full depth (rqae/model.py:18-24: d=2304, c=4, n=5, 1024 levels), 262144 "
                                                "tokens, bf16 in: the matrix-core arm's blocked path (32-level blocks, corrections across blocks and the "
                                                "reconstruction as K-streamed matrix products)", steps=3, warmup=1, n_check=512, levels=1024, spinup_ms=0.0)
                ex = config_leg(np, torch, device, "P1024/bf16 exact arm", "P", 32768, "bf16", "the same shape on the exact arm (canonical fp32 order)",
                                steps=2, warmup=1, n_check=256, levels=1024, arm="exact", spinup_ms=0.0)
                cfgs["P1024/bf16"]["exact_arm"] = {k: ex[k] for k in ("tokens", "ms", "value", "parity")}
                cfgs["P1024/bf16"]["speedup_vs_exact_arm"] = cfgs["P1024/bf16"]["value"] / ex["value"]
            except Exception as e:
                cfgs["P1024/bf16"] = {"error": f"{type(e).__name__}: {e}"}
            line["configs"] = cfgs
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_reference_ops(w)
            line["cpu_baseline_port"] = cpu_baseline(w)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
