#!/bin/bash
# proj3 with the register ring of W pairs: parity tests of the matrix-core arm, then the per-kernel times of the configs
R=gpurun_out/r04j
mkdir -p $R
( timeout -k 10 900 python -m pytest tests/test_gram_arm.py -x -q -m gpu > $R/pytest.txt 2>&1; rc=$?; tail -3 $R/pytest.txt; [ $rc -eq 0 ] ) || exit 1
for spec in "C2 262144 bf16" "C4 65536 bf16" "C4 65536 f32" "C5 262144 bf16" "P 262144 bf16 --levels 16"; do
  set -- $spec
  python tools/aux_bench/config_bench.py $1 $2 --dtype $3 $4 $5 --arms gram --reps 10 >> $R/configs.txt 2>> $R/configs.err || { tail -5 $R/configs.err; exit 1; }
done
python tools/aux_bench/config_bench.py P 262144 --levels 1024 --reps 3 --arms gram >> $R/configs.txt 2>> $R/configs.err || { tail -5 $R/configs.err; exit 1; }
cat $R/configs.txt
