#!/bin/bash
# gram_recon3_kernel with two token sets per wave (TS = 2): parity on the tools build with R3_FORCE=ts2, then times of both forms
R=gpurun_out/r04o
mkdir -p $R
export RQAE_HIP_LIBRARY=tools/_build/r3v/ts/librqae_hip.so
( R3_FORCE=ts2 timeout -k 10 900 python -m pytest tests/test_gram_arm.py tests/test_hip_parity.py -x -q -m gpu > $R/pytest_ts2.txt 2>&1; rc=$?; tail -3 $R/pytest_ts2.txt; [ $rc -eq 0 ] ) || exit 1
for f in ts1 ts2; do
  for spec in "C2 262144" "C2 1048576" "C5 262144" "P 262144 --levels 16"; do
    set -- $spec
    echo "== $f $spec" >> $R/times.txt
    R3_FORCE=$f timeout -k 10 200 python tools/aux_bench/config_bench.py $1 $2 $3 $4 --arms gram --reps 20 2>> $R/err.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['gram'])" >> $R/times.txt || { tail -5 $R/err.txt; exit 1; }
  done
done
cat $R/times.txt
