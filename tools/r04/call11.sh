cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 200 python tools/samp_phases.py 32 2> $O/c11_samp_phases.txt; grep SAMP_DBG $O/c11_samp_phases.txt | cut -c1-170
