# round 4, GPU call 24: non-temporal K/V loads landed: attention parity at every depth / context, then the evidence bank on the final sources
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_fullwidth.py tests/test_gpu_switches.py tests/test_gpu_generate.py -k "not whisper and not qwen3 and not snac and not dac" -m gpu -x -q ) 2>&1 | grep -E "passed|failed|error" | tail -2
SKIP_TESTS=1 bash tools/gpu_round.sh > $O/c24_gpu_round.log 2>&1
python3 -c "
import json
j = json.loads(open('gpurun_out/final/bench.json').read())
print('value', j['value'], 'step', j['roofline']['step']['ms'], 'gate_up', j['roofline']['launch_us'], j['roofline']['frac'], j['roofline']['kernels']['attn_decode_ctx368'])
t = json.load(open('gpurun_out/final/traffic.json')); print(t['kernel_source_sha1'], {k: t[k].get('ratio') for k in ('gate_up', 'attn_decode_ctx368') if k in t})"
