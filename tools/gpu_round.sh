set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_snac.py tests/test_gpu_dac.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
export TMPDIR=/tmp
run() {  # name, workload, start kernel, env...
  name=$1; w=$2; k=$3; shift; shift; shift
  rm -rf /tmp/tr_$name
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python $GRAFT_REPO_ROOT/tools/pmc_codec_probe.py $w > /tmp/tr_$name.log 2>&1)
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  python tools/codec_dispatch_trace.py $f 400 > gpurun_out/dispatch_$name.txt 2>&1
  python tools/dispatch_sum.py gpurun_out/dispatch_$name.txt $k
}
run snac_pwf2 snac32 k_snac_embed A=1
run q3_pwf2 q3b32 k_q3_rvq A=1
