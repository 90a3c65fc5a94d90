# round 6, GPU call 1: token engine at its real shape (17 layers, V 8192, 1 and 4 XCDs), generateStream on the engine, forced fall-back
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
rm -f gpurun_out/parity_observed.jsonl
( timeout 900 python -m pytest tests/test_gpu_token_engine.py tests/test_gpu_soprano.py tests/test_gpu_fulldepth.py -k "token_engine or soprano or engine" -m gpu -q -x --durations=6 ) 2>&1 | grep -vE "^$|warnings" | tail -40 | tee $O/c1_pytest.txt
cp gpurun_out/parity_observed.jsonl $O/c1_parity_observed.jsonl 2>/dev/null
timeout 200 python tools/bench_soprano.py 1 2>&1 | tail -3 | tee $O/c1_bench_soprano.json
