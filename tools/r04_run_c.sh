#!/bin/bash
# round 4: proj3 + recon3: full GPU suite, then per-config kernel times (new defaults vs round-3 kernels)
R=gpurun_out/r04c
mkdir -p $R
( timeout -k 10 600 python -m pytest tests -x -q -m gpu > $R/pytest_gpu.txt 2>&1; rc=$?; tail -4 $R/pytest_gpu.txt; [ $rc -le 1 ] ) || exit 1
for SPEC in "C2 262144" "C4 65536" "C4 65536 --dtype f32" "C5 262144" "P 262144"; do
  echo "#### $SPEC" >> $R/cfg.txt
  python tools/aux_bench/config_bench.py $SPEC --arms gram --reps 20 >> $R/cfg.txt 2>> $R/cfg.err || exit 1
  python tools/aux_bench/config_bench.py $SPEC --arms gram --reps 20 --proj 3 --recon 2 >> $R/cfg.txt 2>> $R/cfg.err || exit 1
done
grep -o '#### .*\|"options": {[^}]*}\|"kernel_ms": {[^}]*}' $R/cfg.txt
