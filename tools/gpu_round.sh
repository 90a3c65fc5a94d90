set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^PARITY" | tail -25
MIS_ATTN_V1=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_attn_v1.log 2>&1; tail -1 gpurun_out/bench_attn_v1.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['phases_ms'], j['roofline']['step']['ms'], j['roofline']['kernels']['attn_decode_ctx368'])"
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_attn_w0.log 2>&1; tail -1 gpurun_out/bench_attn_w0.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['phases_ms'], j['roofline']['step']['ms'], j['roofline']['kernels']['attn_decode_ctx368'])"
