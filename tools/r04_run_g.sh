#!/bin/bash
R=gpurun_out/r04g
mkdir -p $R
( timeout -k 10 900 python -m pytest tests -x -q -m gpu > $R/pytest_gpu.txt 2>&1; rc=$?; tail -6 $R/pytest_gpu.txt; [ $rc -le 1 ] ) || exit 1
python tools/aux_bench/hook_bench.py --released > $R/hook_released.json 2> $R/hook.err || { tail -5 $R/hook.err; exit 1; }
cat $R/hook_released.json
python tools/aux_bench/config_bench.py C2 262144 --free-codebooks --arms gram --reps 20 > $R/vq.txt 2>> $R/hook.err || { tail -5 $R/hook.err; exit 1; }
python tools/aux_bench/config_bench.py C5 65536 --free-codebooks --arms gram --reps 10 >> $R/vq.txt 2>> $R/hook.err || { tail -5 $R/hook.err; exit 1; }
cat $R/vq.txt
