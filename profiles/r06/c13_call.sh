# round 6, GPU call 13: one-wave attention for tiny caches (Qwen3-TTS code predictor) - parity (Qwen3 files + full depth), A/B of the frame loop, kernel stats
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl $O/c13_*
( timeout 900 python -m pytest tests/test_gpu_qwen3tts.py tests/test_gpu_q3_reference.py tests/test_gpu_fulldepth.py tests/test_gpu_fullwidth.py -k "qwen3 or q3 or predictor or frame" -m gpu -q -x --durations=5 ) 2>&1 | grep -vE "^$|warnings|amdgpu.ids" | tail -14 | tee $O/c13_pytest.txt
cp gpurun_out/parity_observed.jsonl $O/c13_parity_observed.jsonl 2>/dev/null
for i in 1 2 3; do for v in 0 1; do echo "ATTN_SMALL=$v $(MIS_ATTN_SMALL=$v timeout 300 python tools/bench_qwen3tts.py 2>&1 | tail -1 | cut -c1-400)" | tee -a $O/c13_qwen3tts_ab.txt; done; done
rm -rf /tmp/ks; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $OLDPWD/tools/bench_qwen3tts.py > /tmp/ks.log 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/c13_qwen3tts_kernel_stats.csv
head -12 $O/c13_qwen3tts_kernel_stats.csv | cut -c1-150
