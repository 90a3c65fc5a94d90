# round 6, GPU call 8: token engine with the host-visible store issued by a matrix wave (parity + bench + kernel stats); Orpheus depth fast variant timing
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
( timeout 600 python -m pytest tests/test_gpu_token_engine.py tests/test_gpu_soprano.py tests/test_gpu_fulldepth.py -k "token_engine or soprano or engine" -m gpu -q -x ) 2>&1 | grep -vE "^$|warnings|amdgpu.ids" | tail -8 | tee $O/c8_pytest_engine.txt
for i in 1 2 3; do timeout 200 python tools/bench_soprano.py 1 2>&1 | tail -1 | tee -a $O/c8_bench_soprano.jsonl; done
rm -rf /tmp/ks2; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python $OLDPWD/tools/bench_soprano.py 1 > /tmp/ks2.log 2>&1)
cp $(find /tmp/ks2 -name "*kernel_stats.csv" | head -1) $O/c8_soprano_engine_kernel_stats.csv
head -4 $O/c8_soprano_engine_kernel_stats.csv | cut -c1-160
( timeout 600 python -m pytest tests/test_gpu_depth.py -m gpu -q --durations=3 ) 2>&1 | grep -vE "^$|amdgpu.ids" | tail -8 | tee $O/c8_pytest_depth.txt
