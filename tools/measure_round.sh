#!/bin/bash
# The measurement pass behind profiles/r02_*: run on a GPU box from the repo root (gpurun -- bash tools/measure_round.sh), then
# `python3 tools/make_traffic.py gpurun_out/r02/pmc.json` anywhere and copy the files named in profiles/README.md.
mkdir -p gpurun_out/r02
python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err; tail -c 600 gpurun_out/r02/bench.json; echo
python bench.py --arm exact --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_exact_arm.json 2>/dev/null
python bench.py --codes-only --no-cpu-baseline > gpurun_out/r02/bench_codes_only.json 2>/dev/null
python bench.py --tokens 1048576 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_1M_tokens.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02/rocprof.err
cd $GRAFT_REPO_ROOT
python3 tools/pmc.py --pass-timeout 150 --out gpurun_out/r02/pmc.json --kernels "gram_" --skip-launches 2 -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r02/pmc.log 2>&1
tail -3 gpurun_out/r02/pmc.log; ls gpurun_out/r02 gpurun_out/r02/stats
