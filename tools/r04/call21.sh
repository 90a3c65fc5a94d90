cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 200 python tools/samp_phases.py 32 2> $O/c21_samp_phases.txt; grep "^SAMP_DBG\|all_ids_live SAMP" $O/c21_samp_phases.txt | cut -c150-900
