# round 6, GPU call 14: shape-only split-K of Soprano's launch-shaped ConvNeXt GEMMs - parity (Soprano, full-depth decoder, SNAC / codec files), configs[1] A/B, kernel stats
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl $O/c14_*
( timeout 900 python -m pytest tests/test_gpu_soprano.py tests/test_gpu_fulldepth.py tests/test_gpu_snac.py tests/test_gpu_dac.py tests/test_gpu_encodec.py -k "not whisper and not qwen3 and not token_engine_at" -m gpu -q -x --durations=4 ) 2>&1 | grep -vE "^$|warnings|amdgpu.ids" | tail -10 | tee $O/c14_pytest.txt
cp gpurun_out/parity_observed.jsonl $O/c14_parity_observed.jsonl 2>/dev/null
for i in 1 2 3; do
  echo "NO_SPLITK $(MIS_BF3_NO_SPLITK=1 timeout 200 python tools/bench_soprano.py 1 2>&1 | tail -1)" | tee -a $O/c14_bench_soprano.txt
  echo "default $(timeout 200 python tools/bench_soprano.py 1 2>&1 | tail -1)" | tee -a $O/c14_bench_soprano.txt
done
rm -rf /tmp/ks2; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python $OLDPWD/tools/bench_soprano.py 1 > /tmp/ks2.log 2>&1)
cp $(find /tmp/ks2 -name "*kernel_stats.csv" | head -1) $O/c14_soprano_engine_kernel_stats.csv
head -8 $O/c14_soprano_engine_kernel_stats.csv | cut -c1-150
