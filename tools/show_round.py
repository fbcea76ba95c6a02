#!/usr/bin/env python3
"""Print the figures DESIGN.md / README.md quote from a measurement pass (gpurun_out/r04 by default)."""
import json, sys
R = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04"
for f in ["bench_driver_cmd", "bench", "bench_codes_only", "bench_exact_arm", "bench_under_rocprof"]:
    try:
        j = json.loads(open(f"{R}/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "missing", e); continue
    r = j["roofline"]
    a1 = r.get("at_1M")
    print(f"{f}: ms {j['ms_per_step']:.4f}  {j['value']/1e6:.1f} M tok/s  frac {r['frac']:.4f}  at_1M {a1.get('frac') if isinstance(a1, dict) else a1}  "
          f"parity {j.get('parity', {}).get('tokens_with_other_codes')}  kernel_ms {r.get('kernel_ms')}  traffic {r.get('traffic')}")
    if f == "bench_driver_cmd":
        for k, v in j["configs"].items():
            if isinstance(v, dict):
                km = v.get("kernel_ms")
                km = {a: round(b, 3) for a, b in km.items()} if km else None
                print(f"   {k}: {v['ms']:.3f} ms  {v['value']/1e6:.1f} M  frac {v['roofline']['frac']:.3f} ({v['roofline']['bound']}; hbm {v['roofline']['hbm_frac']:.3f} mfma {v['roofline'].get('mfma_frac')})  {km}  "
                      f"other {v['parity']['tokens_with_other_codes']} bad {v['parity']['mismatch_at_margin_ge_1e-5']}  x{v.get('speedup_vs_exact_arm')}")
        print("   cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["kind"], j["cpu_baseline"].get("cores"), "| port", j.get("cpu_baseline_port", {}).get("value"))
        print("   legs", {k: (round(v.get("ms_per_step", 0), 4)) for k, v in j.get("legs", {}).items() if isinstance(v, dict)})
        print("   step_ms", j.get("step_ms"))
