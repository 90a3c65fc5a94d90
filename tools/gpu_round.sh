set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_observed.jsonl
timeout 1200 python -m pytest tests -m gpu -x -q -rP 2>&1 | grep -v "^PARITY\|^---\|^$" | tail -40 > gpurun_out/pytest_gpu.log; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_r02_v0.json 2> gpurun_out/bench_r02_v0.err; tail -c 3000 gpurun_out/bench_r02_v0.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_v0 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_v0.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_v0 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_v0_bench_kernel_stats.csv
find gpurun_out/prof_v0 -name "*.csv" -size +4M -delete
timeout 600 bash tools/pmc_traffic.sh r02 2>&1 | tail -40
nproc
