# round 4, GPU call 1: land the four-tile gate+up (parity subset, A/B), run both laboratories, measure inter-kernel gaps
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_depth.py tests/test_gpu_generate.py -m gpu -x -q --durations=8 ) > $O/c1_pytest.txt 2>&1
tail -15 $O/c1_pytest.txt
bash tools/gemm_lab/run_labs.sh > $O/c1_labs.txt 2>&1
cp gpurun_out/gemm_lab_rows32.jsonl gpurun_out/gemm_lab_rows16.jsonl gpurun_out/big_lab.jsonl gpurun_out/big_lab.err $O/ 2>/dev/null
cat $O/c1_labs.txt
timeout 400 python tools/ab_decode.py $O/c1_ab.json r_gu2:MIS_R_GU=2 > $O/c1_ab.log 2>&1
cat $O/c1_ab.log
rm -rf /tmp/kt; R=$(pwd); (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1)
python tools/trace_gaps.py /tmp/kt $O/c1_gaps.json
