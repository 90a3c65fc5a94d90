set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
MIS_SAMPLER_WIDE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_sampler_wide.log 2>&1; tail -1 gpurun_out/bench_sampler_wide.log | cut -c1-700
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_sampler_narrow.log 2>&1; tail -1 gpurun_out/bench_sampler_narrow.log | cut -c1-700
timeout 300 python tools/bench_whisper.py 8 > gpurun_out/whisper_v3.log 2>&1; tail -1 gpurun_out/whisper_v3.log
