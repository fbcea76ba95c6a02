#!/bin/bash
# lattice_fast (approximate scores choose, the exact enumeration behind them): parity, then the chains' times
R=gpurun_out/${OUT:-r04r}
mkdir -p $R
( timeout -k 10 900 python -m pytest tests/test_gram_arm.py tests/test_hip_parity.py -x -q -m gpu > $R/pytest.txt 2>&1; rc=$?; tail -3 $R/pytest.txt; [ $rc -eq 0 ] ) || { grep -n "Error\|assert" $R/pytest.txt | head -20; exit 1; }
for spec in "C5 262144" "P 262144 --levels 16" "P 262144 --levels 32" "C2 262144" "P 262144 --levels 1024"; do
  set -- $spec
  timeout -k 10 300 python tools/aux_bench/config_bench.py $1 $2 $3 $4 --arms gram --reps 10 2>> $R/err.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['config'], j['num_quantizers'], j['gram'])" >> $R/times.txt || { tail -5 $R/err.txt; exit 1; }
done
cat $R/times.txt
