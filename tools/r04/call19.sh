# round 4, GPU call 19: the sampler's failure path under its real cause (CUs held by another stream)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_sampler.py -m gpu -x -q ) > $O/c19_pytest.txt 2>&1; tail -8 $O/c19_pytest.txt | cut -c1-300
