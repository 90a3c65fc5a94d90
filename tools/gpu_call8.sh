set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
R=$PWD
for mode in 1 0; do
  rm -rf /tmp/ks$mode; (cd /tmp && MIS_FUSE_NORM=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$mode -- python $R/tools/bench_qwen3tts.py 32 40 16 > /tmp/ks$mode.log 2>&1)
  cp $(find /tmp/ks$mode -name "*kernel_stats.csv" | head -1) gpurun_out/c8/q3_kernel_stats_fuse$mode.csv
  tail -1 /tmp/ks$mode.log | cut -c1-400
done
head -25 gpurun_out/c8/q3_kernel_stats_fuse1.csv | cut -c1-200
head -25 gpurun_out/c8/q3_kernel_stats_fuse0.csv | cut -c1-200
