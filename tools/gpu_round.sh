# One GPU call that regenerates the evidence under profiles/ (run through gpurun from the repo root):
#   full -m gpu suite INCLUDING the slow full-depth variants (--runslow), the driver's bench command (with its secondary block), rocprofv3 kernel stats of the bench, PMC traffic passes,
#   sampler phase stamps.  Outputs land in gpurun_out/final/; copy what is to be judged into profiles/.
set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/final
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
[ -n "$SKIP_TESTS" ] || { timeout 2400 python -m pytest tests -m gpu --runslow -q -rs --durations=8 2>&1 | grep -E "passed|failed|error|SKIPPED|s call|s setup" | tail -14 > gpurun_out/final/pytest_gpu.txt; cat gpurun_out/final/pytest_gpu.txt; }
cp gpurun_out/parity_observed.jsonl gpurun_out/final/ 2>/dev/null
# PMC traffic first, and installed where bench.py looks for it (on this box's copy of the repo): the bench line below then names the traffic
# measured on the very sources it runs (`traffic_source.matches_current_build`)
timeout 900 bash tools/pmc_traffic.sh final > gpurun_out/final/pmc.log 2>&1; cp gpurun_out/final_pmc/traffic.json gpurun_out/final/traffic.json
rm -rf gpurun_out/final_pmc/fetch gpurun_out/final_pmc/write
mkdir -p profiles/r06_pmc; [ -s gpurun_out/final/traffic.json ] && cp gpurun_out/final/traffic.json profiles/r06_pmc/traffic.json
timeout 900 python bench.py > gpurun_out/final/bench.log 2>&1; tail -1 gpurun_out/final/bench.log > gpurun_out/final/bench.json
rm -rf /tmp/ks; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/ks.log 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) gpurun_out/final/bench_kernel_stats.csv
timeout 120 python tools/samp_phases.py 32 2> gpurun_out/final/samp_phases.txt
rm -rf /tmp/pm; (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/pm -- python $OLDPWD/tools/pmc_codec_probe.py whisper > /tmp/pm.log 2>&1)
python tools/pmc_mfma_reduce.py /tmp/pm gpurun_out/final/whisper_mfma_util.json > /dev/null
# configs[1] on the batch-1 token engine: kernel stats of the generate call (k_token_engine = the LM loop as one launch)
rm -rf /tmp/ks2; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python $OLDPWD/tools/bench_soprano.py 1 > /tmp/ks2.log 2>&1)
cp $(find /tmp/ks2 -name "*kernel_stats.csv" | head -1) gpurun_out/final/soprano_engine_kernel_stats.csv
timeout 200 python tools/bench_token_engine.py > gpurun_out/final/token_engine_bench.jsonl 2>/dev/null
# identity of the sources this evidence belongs to (the box has no .git): sha1 over every source file of the product, the tests and the tools
find mlx-audio-swift_amd include tests tools oracle bench.py __graft_entry__.py -type f \( -name "*.hip" -o -name "*.h" -o -name "*.cpp" -o -name "*.py" -o -name "*.sh" -o -name Makefile \) | sort | xargs sha1sum | sha1sum | cut -c1-16 > gpurun_out/final/source_tree_sha1.txt
