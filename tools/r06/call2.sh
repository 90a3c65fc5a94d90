# round 6, GPU call 2: Whisper decoder with the glue folded into the consuming GEMM - parity, then A/B (alternating runs) + kernel stats
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
( timeout 600 python -m pytest tests/test_gpu_whisper.py -m gpu -q -x --durations=4 ) 2>&1 | grep -vE "^$|warnings" | tail -25 | tee $O/c2_pytest.txt
cp gpurun_out/parity_observed.jsonl $O/c2_parity_observed.jsonl 2>/dev/null
for i in 1 2; do
  for f in 0 1; do
    echo "MIS_WHISPER_FOLD=$f" | tee -a $O/c2_whisper_ab.txt
    MIS_WHISPER_FOLD=$f timeout 300 python tools/bench_whisper.py 2>&1 | tail -1 | tee -a $O/c2_whisper_ab.txt
  done
done
rm -rf /tmp/ks; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $OLDPWD/tools/bench_whisper.py > /tmp/ks.log 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/c2_whisper_fold_kernel_stats.csv
head -25 $O/c2_whisper_fold_kernel_stats.csv
