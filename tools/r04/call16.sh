# round 4, GPU call 16: the A/B switches held to the default's results (tests/test_gpu_switches.py), then the PMC passes on the final sources
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O gpurun_out/final
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_switches.py -m gpu -x -q ) > $O/c16_pytest.txt 2>&1; tail -5 $O/c16_pytest.txt | cut -c1-300
timeout 900 bash tools/pmc_traffic.sh final > gpurun_out/final/pmc.log 2>&1; cp gpurun_out/final_pmc/traffic.json gpurun_out/final/traffic.json
rm -rf gpurun_out/final_pmc/fetch gpurun_out/final_pmc/write
python3 -c "import json; t=json.load(open('gpurun_out/final/traffic.json')); print(t['kernel_source_sha1'], {k: t[k].get('ratio') for k in ('gate_up','lm_head','qkv','o_proj','down','attn_decode_ctx368') if k in t})"
