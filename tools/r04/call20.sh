# round 4, GPU call 20: the driver's entry points on the final tree: smoke() and the new / changed GPU tests
cd ${GRAFT_REPO_ROOT:-.}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( timeout 600 python -m pytest tests/test_gpu_switches.py tests/test_gpu_sampler.py tests/test_gpu_whisper.py tests/test_gpu_generate.py -m gpu -x -q ) 2>&1 | grep -E "passed|failed|error" | tail -2
