# A/B of the one-shot quantised GEMM arrangement (csrc/lm_qgemm.hip, k_gemm_skinny_q1; MIS_QGEMM_V2=1) against the streaming kernel:
# parity tests with the switch on, the Qwen3-TTS frame loop (8 bit with and without, bf16 beside it), per-role GEMM times at
# Orpheus-3B and Qwen3-TTS widths.  Run through gpurun from the repo root; results land in gpurun_out/oneshot/.
set -x
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/oneshot
rm -rf $O; mkdir -p $O
rm -f gpurun_out/qgemm_probe.jsonl
MIS_QGEMM_V2=1 timeout 150 python -m pytest tests/test_gpu_loader.py tests/test_gpu_qwen3tts.py -q -x -k "native_quantised or code_streamed or override or teacher_forcing or quantised" 2>&1 | tail -15 > $O/parity_v2.txt
cat $O/parity_v2.txt
MIS_QGEMM_V2=1 timeout 60 python tools/bench_qwen3tts.py 32 100 8 > $O/q3_8bit_v2.json 2>/dev/null
timeout 60 python tools/bench_qwen3tts.py 32 100 8 > $O/q3_8bit_v1.json 2>/dev/null
timeout 60 python tools/bench_qwen3tts.py 32 100 16 > $O/q3_bf16.json 2>/dev/null
MIS_PROBE_BITS=8,4 MIS_QGEMM_V2=1 timeout 40 python tools/qgemm_probe.py qwen3 32 > /dev/null 2>&1
MIS_PROBE_BITS=8,4 timeout 40 python tools/qgemm_probe.py qwen3 32 > /dev/null 2>&1
MIS_PROBE_BITS=8 MIS_QGEMM_V2=1 timeout 40 python tools/qgemm_probe.py orpheus 32 > /dev/null 2>&1
MIS_PROBE_BITS=8 MIS_QGEMM_V2=1 timeout 40 python tools/qgemm_probe.py orpheus 16 > /dev/null 2>&1
MIS_PROBE_BITS=8 timeout 40 python tools/qgemm_probe.py orpheus 16 > /dev/null 2>&1
cp gpurun_out/qgemm_probe.jsonl $O/qgemm_probe.jsonl
cat $O/qgemm_probe.jsonl
grep -h ms_per_frame $O/q3_*.json | python -c "import sys,json; [print(json.loads(l)['workload'][:60], json.loads(l)['ms_per_frame'], json.loads(l)['audio_s_per_s']) for l in sys.stdin]"
