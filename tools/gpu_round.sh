set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_snac.py tests/test_gpu_dac.py tests/test_gpu_encodec.py tests/test_gpu_soprano.py tests/test_gpu_qwen3tts.py tests/test_gpu_fullwidth.py tests/test_gpu_whisper.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
export TMPDIR=/tmp
run() {  # name, workload, start kernel, env...
  name=$1; w=$2; k=$3; shift; shift; shift
  rm -rf /tmp/tr_$name
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python $GRAFT_REPO_ROOT/tools/pmc_codec_probe.py $w > /tmp/tr_$name.log 2>&1)
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  python tools/codec_dispatch_trace.py $f 400 > gpurun_out/dispatch_$name.txt 2>&1
  python tools/dispatch_sum.py gpurun_out/dispatch_$name.txt $k
}
run snac_rs snac32 k_snac_embed A=1
run q3_rs q3b32 k_q3_rvq A=1
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_rs.log 2>&1; tail -1 gpurun_out/bench_rs.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['phases_ms'])"
