# round 4, GPU call 4: sampler v3 (LDS staging, DPP scans) parity + phases, arrangement A/B, in-chain kernel stats
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_generate.py tests/test_gpu_whisper.py -m gpu -x -q ) > $O/c4_pytest.txt 2>&1
tail -6 $O/c4_pytest.txt
timeout 120 python tools/samp_phases.py 32 2> $O/c4_samp_phases.txt; grep SAMP_DBG $O/c4_samp_phases.txt
timeout 500 python tools/ab_decode.py $O/c4_ab.json o_r2k4s2:MIS_ARR_O=2,4,4,2 head_r2k1:MIS_ARR_HEAD=2,1,4 qkv_r2k4:MIS_ARR_QKV=2,4,4,3 qkv_r4k2:MIS_ARR_QKV=4,2,2,3 > $O/c4_ab.log 2>&1
cat $O/c4_ab.log
rm -rf /tmp/ks; R=$(pwd); (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/ks.log 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/c4_bench_kernel_stats.csv
head -14 $O/c4_bench_kernel_stats.csv | cut -c1-200
