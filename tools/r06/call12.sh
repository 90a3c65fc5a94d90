# round 6, GPU call 12: short prompts on the decode-step kernels - parity (LM file, Soprano, token engine), configs[1] bench + kernel stats
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl $O/c12_*
( timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_soprano.py tests/test_gpu_token_engine.py tests/test_gpu_generate.py -m gpu -q -x --durations=5 ) 2>&1 | grep -vE "^$|warnings|amdgpu.ids" | tail -14 | tee $O/c12_pytest.txt
cp gpurun_out/parity_observed.jsonl $O/c12_parity_observed.jsonl 2>/dev/null
for i in 1 2; do
  echo "PREFILL_SMALL=0 $(MIS_PREFILL_SMALL=0 timeout 200 python tools/bench_soprano.py 1 2>&1 | tail -1)" | tee -a $O/c12_bench_soprano.txt
  echo "default $(timeout 200 python tools/bench_soprano.py 1 2>&1 | tail -1)" | tee -a $O/c12_bench_soprano.txt
done
rm -rf /tmp/ks2; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python $OLDPWD/tools/bench_soprano.py 1 > /tmp/ks2.log 2>&1)
cp $(find /tmp/ks2 -name "*kernel_stats.csv" | head -1) $O/c12_soprano_engine_kernel_stats.csv
head -12 $O/c12_soprano_engine_kernel_stats.csv | cut -c1-150
