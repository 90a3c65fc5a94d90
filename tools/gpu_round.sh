set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^PARITY" | tail -8
export TMPDIR=/tmp
run() {  # name, workload, start kernel, env...
  name=$1; w=$2; k=$3; shift; shift; shift
  rm -rf /tmp/tr_$name
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python $GRAFT_REPO_ROOT/tools/pmc_codec_probe.py $w > /tmp/tr_$name.log 2>&1)
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  python tools/codec_dispatch_trace.py $f 400 > gpurun_out/dispatch_$name.txt 2>&1
  python tools/dispatch_sum.py gpurun_out/dispatch_$name.txt $k
}
run q3_fs q3b32 k_q3_rvq A=1
run snac_fs snac32 k_snac_embed A=1
timeout 300 python tools/bench_qwen3tts.py 32 100 16 > gpurun_out/q3_fs.log 2>&1; tail -1 gpurun_out/q3_fs.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_fs.log 2>&1; tail -1 gpurun_out/bench_fs.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['phases_ms'])"
