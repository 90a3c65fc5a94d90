# round 6, GPU call 7: Soprano decoder LayerNorm in one pass (parity + configs[1] bench with kernel stats); then the whole -m gpu suite (default set)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
( timeout 400 python -m pytest tests/test_gpu_soprano.py tests/test_gpu_fulldepth.py -k "soprano or decoder" -m gpu -q -x ) 2>&1 | grep -vE "^$|warnings|amdgpu.ids" | tail -8 | tee $O/c7_pytest_soprano.txt
for i in 1 2; do timeout 200 python tools/bench_soprano.py 1 2>&1 | tail -1 | tee -a $O/c7_bench_soprano.jsonl; done
rm -rf /tmp/ks2; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python $OLDPWD/tools/bench_soprano.py 1 > /tmp/ks2.log 2>&1)
cp $(find /tmp/ks2 -name "*kernel_stats.csv" | head -1) $O/c7_soprano_engine_kernel_stats.csv
head -8 $O/c7_soprano_engine_kernel_stats.csv | cut -c1-160
( timeout 1500 python -m pytest tests -m gpu -q -rs --durations=10 ) 2>&1 | grep -vE "^$|amdgpu.ids" | tail -30 | tee $O/c7_pytest_gpu_default_set.txt
cp gpurun_out/parity_observed.jsonl $O/c7_parity_observed.jsonl 2>/dev/null
