# round 4, GPU call 14: single-wave / few-wave glue for the small models (d = 1024, 1280, 768, 512): parity + A/B on the secondary workloads
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_fullwidth.py tests/test_gpu_soprano.py -m gpu -x -q ) > $O/c14_pytest.txt 2>&1; grep -E "passed|failed|error" $O/c14_pytest.txt | tail -2
for rep in 1 2; do for v in "" 1; do
  MIS_GLUE_CPT=$v timeout 200 python tools/bench_whisper.py 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('whisper cpt_env=$v', round(j['transcribe_ms'],1), round(j['encode_ms'],1))"
  MIS_GLUE_CPT=$v timeout 200 python tools/bench_qwen3tts.py 32 100 16 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('qwen3 bf16 cpt_env=$v ms_per_frame', round(j['ms_per_frame'],3), 'audio_s_per_s', round(j['audio_s_per_s'],1))"
done; done > $O/c14_secondary_ab.txt 2>&1; cat $O/c14_secondary_ab.txt
