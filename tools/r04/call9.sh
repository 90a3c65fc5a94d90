# round 4, GPU call 9: attention with the first two tiles requested up front (MIS_ATTN_PAIR=1): parity + A/B; sampler after the scratch fix
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( MIS_ATTN_PAIR=1 timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_fullwidth.py -k "not whisper and not qwen3 and not snac and not dac" -m gpu -x -q ) > $O/c9_pytest_pair.txt 2>&1; grep -E "passed|failed|error" $O/c9_pytest_pair.txt | tail -2
timeout 120 python tools/samp_phases.py 32 2> $O/c9_samp_phases.txt; grep SAMP_DBG $O/c9_samp_phases.txt | cut -c1-160
timeout 400 python tools/ab_decode.py $O/c9_ab.json pair:MIS_ATTN_PAIR=1 pair_again:MIS_ATTN_PAIR=1 base_again: > $O/c9_ab.log 2>&1
cat $O/c9_ab.log
