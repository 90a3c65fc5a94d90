cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 120 python tools/samp_phases.py 32 2> $O/c8_samp_phases.txt; grep SAMP_DBG $O/c8_samp_phases.txt
