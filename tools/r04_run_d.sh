#!/bin/bash
R=gpurun_out/r04d
mkdir -p $R
( timeout -k 10 600 python -m pytest tests -x -q -m gpu > $R/pytest_gpu.txt 2>&1; rc=$?; tail -4 $R/pytest_gpu.txt; [ $rc -le 1 ] ) || exit 1
python bench.py --steps 20 --warmup 5 > $R/bench.json 2> $R/bench.err || exit 1
for SPEC in "C2 262144" "C4 65536" "C5 262144" "P 262144"; do
  echo "#### $SPEC" >> $R/cfg.txt
  python tools/aux_bench/config_bench.py $SPEC --arms gram --reps 20 >> $R/cfg.txt 2>> $R/cfg.err || exit 1
done
grep -o '#### .*\|"options": {[^}]*}\|"kernel_ms": {[^}]*}' $R/cfg.txt
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04d/bench.json'))
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('frac_from_ms_per_step'), j['roofline']['kernel_ms'], j['roofline'].get('at_1M',{}).get('frac'))
print(j['parity'])
for k,v in j['configs'].items():
    if isinstance(v,dict):
        print(k, {kk:v[kk] for kk in v if kk in('ms','value','kernel_ms','error')}, {a:round(b,3) if isinstance(b,float) else b for a,b in (v.get('roofline') or {}).items()}, v.get('parity'))
PY
