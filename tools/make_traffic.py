#!/usr/bin/env python3
"""profiles/<round>_traffic.json and profiles/<round>_pmc_summary.json (round = bench.PROFILE_ROUND) from a tools/pmc.py run of bench.py.

    (on the GPU box)  python3 tools/pmc.py --out gpurun_out/r03/pmc.json --kernels gram_ --skip-launches 2 -- python3 bench.py --no-cpu-baseline --no-extra-legs --no-parity --spinup-ms 0 --steps 4 --warmup 2
    (anywhere)        python3 tools/make_traffic.py gpurun_out/r03/pmc.json

HBM bytes per forward = sum over the arm's three kernels of (2 x 1024 x FETCH_SIZE + 1024 x WRITE_SIZE): FETCH_SIZE counts half of
the bytes of 16-byte-per-lane streaming reads on gfx950 (MI355X_MICROARCH.md, HBM; every read of these kernels is of that form:
LDS-DMA pieces and dwordx4 loads), WRITE_SIZE counts stores exactly.  The file carries a hash of the kernel sources; bench.py
reports `traffic` only when it matches the build it runs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    return k.split("(")[0].strip()


def main():
    src = sys.argv[1]
    j = json.load(open(src))
    ks = {k: v for k, v in j["kernels"].items() if any(t in k for t in ("gram_fused_kernel", "gram_levels", "gram_proj", "gram_recon", "deep_", "sign_chain"))}
    configs = "--configs" in sys.argv      # the pass over bench.py's `configs` block: a per-kernel summary only (no C2 traffic file)
    rd = sum(v.get("hbm_read_bytes", 0.0) for v in ks.values())
    wr = sum(v.get("hbm_write_bytes", 0.0) for v in ks.values())
    tokens = 262144
    out = {"tokens": tokens, "arm": "gram", "sources_sha256_16": bench.sources_hash(), "hbm_bytes_per_step": rd + wr,
           "hbm_read_bytes": rd, "hbm_write_bytes": wr, "algorithmic_bytes_per_step": bench.BYTES_PER_TOKEN * tokens,
           "per_kernel": {short(k): {"read": v.get("hbm_read_bytes"), "write": v.get("hbm_write_bytes")}
                          for k, v in ks.items()},
           "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python3 bench.py --no-cpu-baseline --no-extra-legs --no-parity --spinup-ms 0 --steps 4 --warmup 2` by tools/pmc.py; "
                   "KiB x 1024, reads x 2 (gfx950 counts 128-byte requests of 16-byte-per-lane streams at 64 bytes).  Above the algorithmic bytes by the "
                   "intermediates of the three-kernel path (projections written + read, fp32 straight-through values written + read) and the weight images' first touch."}
    if not configs:
        json.dump(out, open(os.path.join(ROOT, "profiles", f"{bench.PROFILE_ROUND}_traffic.json"), "w"), indent=1)
    summ = {}
    for k, v in ks.items():
        name = short(k)
        summ[name] = {c: v[c] for c in sorted(v) if c not in ("mean_ns_by_pass",)}
        ns = v.get("mean_ns_by_pass", {})
        if ns:
            summ[name]["mean_kernel_ns"] = sum(ns.values()) / len(ns)
        if "hbm_read_bytes" in v or "hbm_write_bytes" in v:
            summ[name]["hbm_bytes"] = v.get("hbm_read_bytes", 0.0) + v.get("hbm_write_bytes", 0.0)
            if ns:
                summ[name]["hbm_TBs"] = summ[name]["hbm_bytes"] / (sum(ns.values()) / len(ns)) / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            # busy cycles summed over the 1024 SIMDs against the kernel's shader cycles (GRBM_GUI_ACTIVE sums 8 XCDs)
            summ[name]["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)
    json.dump({"command": j["command"], "passes": j["passes"], "sources_sha256_16": bench.sources_hash(), "kernels": summ},
              open(os.path.join(ROOT, "profiles", f"{bench.PROFILE_ROUND}_pmc_{'configs' if configs else 'summary'}.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: out[k] for k in ("hbm_bytes_per_step", "algorithmic_bytes_per_step", "sources_sha256_16")}))
    for n, v in summ.items():
        print(n, "mfma_busy_frac", round(v.get("mfma_busy_frac", 0), 3), "ms", round(v.get("mean_kernel_ns", 0) / 1e6, 4))


if __name__ == "__main__":
    main()
