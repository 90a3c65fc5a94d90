# round 4, GPU call 26: Whisper cross-attention with two pairs of key tiles in flight (k_attn_decode<64, 2, XS>) against the pair-at-a-time
# loop (MIS_ATTN_XS=0), alternating, tokens compared by checksum; then the evidence bank on these sources; then the Whisper parity tests
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
for v in 0 1 0 1; do MIS_ATTN_XS=$v timeout 120 python tools/bench_whisper.py 2>&1 | tail -1; done > $O/c26_whisper_xs_ab.txt
cat $O/c26_whisper_xs_ab.txt | cut -c1-400
SKIP_TESTS=1 bash tools/gpu_round.sh > $O/c26_gpu_round.log 2>&1
python3 -c "
import json
j = json.loads(open('gpurun_out/final/bench.json').read())
print('value', j['value'], 'step', j['roofline']['step']['ms'], j['roofline']['traffic_source'])
print({k: (round(v['audio_s_per_s'], 1), v.get('ms')) for k, v in j['secondary'].items() if isinstance(v, dict)})"
( timeout 150 python -m pytest tests/test_gpu_whisper.py -m gpu -x -q ) 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/c26_pytest_whisper.txt
