# A/B of the split-K factors of the hidden-1024 models (Qwen3-TTS): fewer slabs = a cheaper glue launch, a longer K range per wave
cd ${GRAFT_REPO_ROOT:-.}
for v in "" "MIS_S_DOWN=4" "MIS_S_DOWN=4 MIS_S_O=2" "MIS_S_DOWN=2 MIS_S_O=2 MIS_S_QKV=2" "MIS_S_DOWN=4 MIS_S_O=2 MIS_S_QKV=2" "MIS_S_DOWN=8 MIS_S_O=8 MIS_S_QKV=8"; do
  r=$(env $v python tools/bench_qwen3tts.py 32 60 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_frame'],4))")
  echo "{\"env\": \"$v\", \"ms_per_frame\": $r}" | tee -a gpurun_out/ab_q3_split.jsonl
done
