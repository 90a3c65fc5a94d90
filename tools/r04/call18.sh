# round 4, GPU call 18: per-kernel table of the decode step on the 8-bit Orpheus-3B checkpoint
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/kq; R=$(pwd); (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kq -- python $R/tools/bench_orpheus_q8.py 8 > /tmp/kq.log 2>&1); tail -1 /tmp/kq.log | cut -c1-400
cp $(find /tmp/kq -name "*kernel_stats.csv" | head -1) $O/c18_orpheus_q8_kernel_stats.csv; head -12 $O/c18_orpheus_q8_kernel_stats.csv | cut -c1-110,200-330
