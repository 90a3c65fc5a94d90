# round 6, GPU call 10: k_gemm_big3 pipelining variants (bit identity against the 128 x 128 kernel, encoder time, kernel stats of the best)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -f $O/c10_*.txt
for v in 0 1 2; do
  export MIS_GEMM_BIG3_VAR=$v
  echo "VAR=$v $( (timeout 300 python -m pytest tests/test_gpu_whisper.py -m gpu -q -x -k '256x256 or encoder_and_teacher' 2>&1 | grep -E 'passed|failed' | tail -1) )" | tee -a $O/c10_big3_variants.txt
done
for i in 1 2 3; do
  for v in 0 1 2; do
    echo "VAR=$v $(MIS_GEMM_BIG3_VAR=$v timeout 300 python tools/bench_whisper.py 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("transcribe_ms %.2f encode_ms %.2f enc_TF %.1f crc %d" % (d["transcribe_ms"], d["encode_ms"], d["encoder_TFLOPs"], d["token_crc32"]))')" | tee -a $O/c10_big3_variants.txt
  done
done
