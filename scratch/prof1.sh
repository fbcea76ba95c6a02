set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof1
python scratch/time_variants.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof1/ablate.txt
cat gpurun_out/prof1/ablate.txt
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/prof1/counters.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1/kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof1/bench_kt.json 2> gpurun_out/prof1/kt.err
ls -R gpurun_out/prof1/kt | head -20
