# round 4, GPU call 6: sampler prologue order, Whisper encoder kernel stats (is k_gemm_big3 running?), quantised wide-role arrangements
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_sampler.py -m gpu -x -q ) > $O/c6_pytest.txt 2>&1; grep -E "passed|failed|error" $O/c6_pytest.txt | tail -2
timeout 120 python tools/samp_phases.py 32 2> $O/c6_samp_phases.txt; grep SAMP_DBG $O/c6_samp_phases.txt
rm -rf /tmp/kw; R=$(pwd); (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kw -- python $R/tools/bench_whisper.py > /tmp/kw.log 2>&1); tail -1 /tmp/kw.log | cut -c1-300
cp $(find /tmp/kw -name "*kernel_stats.csv" | head -1) $O/c6_whisper_kernel_stats.csv; head -12 $O/c6_whisper_kernel_stats.csv | cut -c1-160
rm -f gpurun_out/qgemm_probe.jsonl
for v in "2,4:2,1" "4,4:4,4" "4,8:4,8" "2,8:2,4"; do
  gu=${v%%:*}; hd=${v##*:}
  MIS_PROBE_BITS=8 MIS_QARR_GU=$gu MIS_QARR_HEAD=$hd timeout 200 python tools/qgemm_probe.py orpheus 32 2>/dev/null | tail -1 | sed "s/^/GU=$gu HEAD=$hd /"
done > $O/c6_qgemm_arr.txt 2>&1
python3 - <<'PY'
import json
for l in open("gpurun_out/r04/c6_qgemm_arr.txt"):
    i = l.find("{")
    if i < 0:
        print(l.strip()[:200]); continue
    j = json.loads(l[i:])
    print(l[:i], {k: (j[k]["us"], j[k]["GBps"]) for k in ("qkv", "o_proj", "gate_up", "down", "lm_head")})
PY
