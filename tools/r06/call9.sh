# round 6, GPU call 9: the one-slab arrangement of the Whisper decoder step (eight-wave producers, LayerNorm 1 in q|k|v) - parity, A/B, kernel stats
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl $O/c9_whisper_fold_variants.txt
( timeout 600 python -m pytest tests/test_gpu_whisper.py -m gpu -q -x --durations=4 ) 2>&1 | grep -vE "^$|warnings" | tail -25 | tee $O/c9_pytest.txt
cp gpurun_out/parity_observed.jsonl $O/c9_parity_observed.jsonl 2>/dev/null
for i in 1 2 3; do
  for v in 0 20 84; do
    export MIS_WHISPER_FOLD=$v
    echo "FOLD=$v $(timeout 300 python tools/bench_whisper.py 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("transcribe_ms %.2f encode_ms %.2f crc %d" % (d["transcribe_ms"], d["encode_ms"], d["token_crc32"]))')" | tee -a $O/c9_whisper_fold_variants.txt
  done
done
unset MIS_WHISPER_FOLD
rm -rf /tmp/ks; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $OLDPWD/tools/bench_whisper.py > /tmp/ks.log 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/c9_whisper_kernel_stats.csv
head -12 $O/c9_whisper_kernel_stats.csv | cut -c1-260
