# round 4, GPU call 10: row maximum from the output projection's per-item maxima (no first barrier in the one-launch sampler): parity + A/B
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_generate.py tests/test_gpu_sampler.py tests/test_gpu_lm.py tests/test_gpu_soprano.py -m gpu -x -q ) > $O/c10_pytest.txt 2>&1; grep -E "passed|failed|error" $O/c10_pytest.txt | tail -3
timeout 500 python tools/ab_decode.py $O/c10_ab.json no_tmax:MIS_SAMPLER_TMAX=0 tmax_again: no_tmax_again:MIS_SAMPLER_TMAX=0 > $O/c10_ab.log 2>&1
cat $O/c10_ab.log
rm -rf /tmp/ks; R=$(pwd); (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/ks.log 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/c10_bench_kernel_stats.csv
grep -E "k_samp_cluster|2, 4, 1, 4, 3|2, 4, 5, 4, 3" $O/c10_bench_kernel_stats.csv | cut -c1-60,150-260
