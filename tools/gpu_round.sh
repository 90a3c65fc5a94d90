set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
export TMPDIR=/tmp
# 1. full GPU suite
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^PARITY" | tail -4 > gpurun_out/final/pytest_gpu.txt; cat gpurun_out/final/pytest_gpu.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/final/ 2>/dev/null
# 2. the driver's bench command
timeout 900 python bench.py > gpurun_out/final/bench.log 2>&1; tail -1 gpurun_out/final/bench.log > gpurun_out/final/bench.json; python -c "import json; j=json.load(open('gpurun_out/final/bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline'])"
# 3. rocprofv3 kernel stats of the same command (fewer steps)
rm -rf /tmp/ks; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/ks.log 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) gpurun_out/final/bench_kernel_stats.csv; head -8 gpurun_out/final/bench_kernel_stats.csv | cut -c1-150
# 4. PMC traffic on the final sources
timeout 900 bash tools/pmc_traffic.sh final > gpurun_out/final/pmc.log 2>&1; cp gpurun_out/final_pmc/traffic.json gpurun_out/final/traffic.json; python -c "import json; j=json.load(open('gpurun_out/final/traffic.json')); print(j.get('kernel_source_sha1'), {k: v for k, v in j.items() if k not in ('kernels','_how')})" | cut -c1-600
rm -rf gpurun_out/final_pmc/fetch gpurun_out/final_pmc/write
# 5. secondary benches
timeout 300 python tools/bench_whisper.py > gpurun_out/final/whisper.json 2>/dev/null; tail -1 gpurun_out/final/whisper.json
timeout 300 python tools/bench_soprano.py 32 > gpurun_out/final/soprano_b32.json 2>/dev/null; tail -1 gpurun_out/final/soprano_b32.json
timeout 300 python tools/bench_soprano.py 1 > gpurun_out/final/soprano_b1.json 2>/dev/null; tail -1 gpurun_out/final/soprano_b1.json
timeout 300 python tools/bench_qwen3tts.py 32 100 16 > gpurun_out/final/q3_bf16.json 2>/dev/null; tail -1 gpurun_out/final/q3_bf16.json
timeout 300 python tools/bench_qwen3tts.py 32 100 8 > gpurun_out/final/q3_8bit.json 2>/dev/null; tail -1 gpurun_out/final/q3_8bit.json
# 6. MFMA utilisation of the MFMA-bound kernels (Whisper encoder, codecs)
rm -rf /tmp/pm; (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/pm -- python $GRAFT_REPO_ROOT/tools/pmc_codec_probe.py all > /tmp/pm.log 2>&1)
python tools/pmc_mfma_reduce.py /tmp/pm gpurun_out/final/mfma_util.json > /dev/null; python -c "
import json; j=json.load(open('gpurun_out/final/mfma_util.json'))
for k,v in list(j['kernels'].items())[:12]: print(k[:44], v)"
