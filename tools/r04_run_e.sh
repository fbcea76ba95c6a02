#!/bin/bash
R=gpurun_out/r04e
mkdir -p $R
( timeout -k 10 900 python -m pytest tests -x -q -m gpu > $R/pytest_gpu.txt 2>&1; rc=$?; tail -6 $R/pytest_gpu.txt; [ $rc -le 1 ] ) || exit 1
python tools/aux_bench/config_bench.py P 262144 --levels 1024 --reps 3 --arms gram > $R/p1024.txt 2> $R/p1024.err || { tail -5 $R/p1024.err; exit 1; }
python tools/aux_bench/config_bench.py P 524288 --levels 1024 --reps 2 --arms gram >> $R/p1024.txt 2>> $R/p1024.err || { tail -5 $R/p1024.err; exit 1; }
cat $R/p1024.txt
python bench.py --steps 20 --warmup 5 > $R/bench.json 2> $R/bench.err || { tail -5 $R/bench.err; exit 1; }
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04e/bench.json'))
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('frac_from_ms_per_step'), j['roofline']['kernel_ms'], j['roofline'].get('at_1M',{}).get('frac'))
for k,v in j['configs'].items():
    if isinstance(v,dict):
        print(k, {kk:v[kk] for kk in v if kk in('ms','value','kernel_ms','error','speedup_vs_exact_arm','kernel_ms_per_131072_tokens')}, {a:round(b,3) if isinstance(b,float) else b for a,b in (v.get('roofline') or {}).items()}, v.get('parity'))
print(j['cpu_baseline']['value'], j['cpu_baseline_port']['value'])
PY
