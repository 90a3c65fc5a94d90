# round 6, GPU call 6: cleaned decoder folds (parity), the tiled encoder attention (bit identity, encode time, kernel stats, MfmaUtil)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl $O/c6_*.txt
( timeout 600 python -m pytest tests/test_gpu_whisper.py -m gpu -q -x --durations=4 ) 2>&1 | grep -vE "^$|warnings" | tail -25 | tee $O/c6_pytest.txt
cp gpurun_out/parity_observed.jsonl $O/c6_parity_observed.jsonl 2>/dev/null
for i in 1 2; do
  for v in 1 0; do
    echo "PREFILL_V1=$v $(MIS_ATTN_PREFILL_V1=$v timeout 300 python tools/bench_whisper.py 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("transcribe_ms %.2f encode_ms %.2f enc_TF %.1f crc %d" % (d["transcribe_ms"], d["encode_ms"], d["encoder_TFLOPs"], d["token_crc32"]))')" | tee -a $O/c6_whisper_encoder_attention_ab.txt
  done
done
rm -rf /tmp/ks; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $OLDPWD/tools/bench_whisper.py > /tmp/ks.log 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/c6_whisper_kernel_stats.csv
head -14 $O/c6_whisper_kernel_stats.csv | cut -c1-200
rm -rf /tmp/pm; (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/pm -- python $OLDPWD/tools/pmc_codec_probe.py whisper > /tmp/pm.log 2>&1)
python tools/pmc_mfma_reduce.py /tmp/pm $O/c6_whisper_mfma_util.json | tail -12
