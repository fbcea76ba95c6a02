#!/usr/bin/env python3
"""Per-kernel hardware counters of one command, by separate rocprofv3 --pmc passes (run ON the GPU box).

    python3 tools/pmc.py --out gpurun_out/pmc.json [--kernels REGEX] [--pass "C1 C2 ..."]... -- python3 bench.py ...

Each --pass is one rocprofv3 run of the command with that counter set (gfx950: 8 SQ slots, FETCH_SIZE and WRITE_SIZE do
not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"); the program goes directly after `--` (no env / bash -c hop).
Without --pass the default passes below are taken.  Output: {kernel: {counter: mean per launch, "launches": n,
"mean_ns": kernel duration in the counter runs}} plus the corrected HBM byte counts:
FETCH_SIZE and WRITE_SIZE are reported in KiB (the tool's unit); on gfx950 FETCH_SIZE counts half of the bytes of 16-byte-per-lane
streaming reads (MI355X_MICROARCH.md, HBM), so hbm_read_bytes = 2 * 1024 * FETCH_SIZE for kernels whose reads are all of that
form (the gram arm's x, operand and image loads are) and hbm_write_bytes = 1024 * WRITE_SIZE.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

DEFAULT_PASSES = [
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE",
    "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM",
    "FETCH_SIZE",
    "WRITE_SIZE",
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--kernels", default="gram_|rqae_", help="regex on kernel names to keep")
    ap.add_argument("--pass", dest="passes", action="append", default=None)
    ap.add_argument("--pass-timeout", type=int, default=150, help="seconds per rocprofv3 pass")
    ap.add_argument("--skip-launches", type=int, default=0, help="ignore the first N launches of each kernel (warm-up)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cmd = args.cmd[1:] if args.cmd and args.cmd[0] == "--" else args.cmd
    if not cmd:
        sys.exit("no command")
    passes = args.passes or DEFAULT_PASSES
    keep = re.compile(args.kernels)
    out_abs = os.path.abspath(args.out)
    cmd = [os.path.abspath(c) if os.path.exists(c) else c for c in cmd]
    agg = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for pi, counters in enumerate(passes):
        d = tempfile.mkdtemp(prefix=f"pmc{pi}_", dir="/tmp")
        full = ["rocprofv3", "--pmc", *counters.split(), "--output-format", "csv", "-d", d, "-o", "p", "--", *cmd]
        print(f"pass {pi}: {counters}", flush=True)
        try:
            r = subprocess.run(full, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=args.pass_timeout)
        except subprocess.TimeoutExpired:
            print(f"pass {pi} timed out after {args.pass_timeout} s -- skipped", flush=True)
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            print(r.stdout[-1500:], flush=True)
            print(f"pass {pi} ({counters}) failed rc={r.returncode}, csv files: {files} -- skipped", flush=True)
            continue
        per = {}   # (kernel, dispatch) -> {counter: value}
        dur = {}
        for f in files:
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"]
                    if not keep.search(k):
                        continue
                    key = (k, int(row["Dispatch_Id"]))
                    per.setdefault(key, {}).setdefault(row["Counter_Name"], 0.0)
                    per[key][row["Counter_Name"]] += float(row["Counter_Value"])
                    if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                        dur[key] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        by_kernel = {}
        for (k, disp), vals in sorted(per.items(), key=lambda kv: kv[0][1]):
            by_kernel.setdefault(k, []).append((vals, dur.get((k, disp))))
        for k, lst in by_kernel.items():
            lst = lst[args.skip_launches:] if len(lst) > args.skip_launches else lst
            rec = agg.setdefault(k, {})
            rec["launches"] = len(lst)
            names = set().union(*[v.keys() for v, _ in lst])
            for c in names:
                rec[c] = sum(v.get(c, 0.0) for v, _ in lst) / len(lst)
            ds = [t for _, t in lst if t]
            if ds:
                rec.setdefault("mean_ns_by_pass", {})[counters.split()[0]] = sum(ds) / len(ds)
    for k, rec in agg.items():
        if "FETCH_SIZE" in rec:
            rec["hbm_read_bytes"] = 2.0 * 1024.0 * rec["FETCH_SIZE"]
        if "WRITE_SIZE" in rec:
            rec["hbm_write_bytes"] = 1024.0 * rec["WRITE_SIZE"]
    os.makedirs(os.path.dirname(out_abs) or ".", exist_ok=True)
    with open(out_abs, "w") as fh:
        json.dump({"command": " ".join(cmd), "passes": passes, "kernels": agg}, fh, indent=1, sort_keys=True)
    print(f"wrote {out_abs}: {len(agg)} kernels")


if __name__ == "__main__":
    main()
