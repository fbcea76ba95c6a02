#!/bin/bash
# The measurement pass behind profiles/r04_*: run on a GPU box from the repo root (gpurun -- bash tools/measure_round.sh), then
# `python3 tools/make_traffic.py gpurun_out/r04/pmc.json` anywhere and copy the files named in profiles/README.md.
R=gpurun_out/r04
mkdir -p $R
# the driver's own command line first (fresh process, 20 steps after 5 warm-up), then the default one
python bench.py --gpus 1 --steps 20 --warmup 5 > $R/bench_driver_cmd.json 2> $R/bench.err && tail -c 300 $R/bench_driver_cmd.json && echo &&
python bench.py > $R/bench.json 2>> $R/bench.err &&
python bench.py --arm exact --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs --no-configs > $R/bench_exact_arm.json 2>/dev/null &&
python bench.py --codes-only --no-cpu-baseline --no-extra-legs > $R/bench_codes_only.json 2>/dev/null &&
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --no-parity > $GRAFT_REPO_ROOT/$R/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$R/rocprof.err &&
cd $GRAFT_REPO_ROOT &&
python3 tools/pmc.py --pass-timeout 150 --out $R/pmc.json --kernels "gram_" --skip-launches 2 -- python3 bench.py --no-cpu-baseline --no-extra-legs --no-parity --no-configs --spinup-ms 0 --steps 4 --warmup 2 > $R/pmc.log 2>&1 &&
python3 tools/pmc.py --pass-timeout 200 --out $R/pmc_configs.json --kernels "gram_|deep_|sign_" -- python3 bench.py --no-cpu-baseline --no-extra-legs --no-parity --spinup-ms 0 --steps 2 --warmup 1 > $R/pmc_configs.log 2>&1
tail -3 $R/pmc.log $R/pmc_configs.log; ls $R $R/stats
