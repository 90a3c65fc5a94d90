# round 4, GPU call 7: who waits for whom in the one-launch sampler; Whisper encoder after the epilogue changes
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_whisper.py tests/test_gpu_sampler.py -m gpu -x -q ) > $O/c7_pytest.txt 2>&1; grep -E "passed|failed|error" $O/c7_pytest.txt | tail -2
timeout 120 python tools/samp_phases.py 32 2> $O/c7_samp_phases.txt; grep SAMP_DBG $O/c7_samp_phases.txt
rm -rf /tmp/kw; R=$(pwd); (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kw -- python $R/tools/bench_whisper.py > /tmp/kw.log 2>&1); tail -1 /tmp/kw.log | cut -c1-300
cp $(find /tmp/kw -name "*kernel_stats.csv" | head -1) $O/c7_whisper_kernel_stats.csv; head -12 $O/c7_whisper_kernel_stats.csv | cut -c1-130
( timeout 600 python -m pytest tests/test_gpu_fullwidth.py -k whisper -m gpu -x -q ) 2>&1 | grep -E "passed|failed|error" | tail -2
