# round 4, GPU call 5: sampler prologue, Whisper 256x256 GEMM (parity + A/B), quantised wide-role arrangements, bench secondary block
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_whisper.py tests/test_gpu_loader.py tests/test_gpu_generate.py -m gpu -x -q ) > $O/c5_pytest.txt 2>&1
tail -6 $O/c5_pytest.txt
timeout 120 python tools/samp_phases.py 32 2> $O/c5_samp_phases.txt; grep SAMP_DBG $O/c5_samp_phases.txt
for v in "" "0"; do MIS_GEMM_BIG3=$v timeout 200 python tools/bench_whisper.py 2>/dev/null | tail -1 | sed "s/^/big3=$v /"; done > $O/c5_whisper_big3_ab.txt; cat $O/c5_whisper_big3_ab.txt
rm -f gpurun_out/qgemm_probe.jsonl
for v in "GU=2,4 HEAD=2,1" "GU=4,4 HEAD=4,4" "GU=4,8 HEAD=4,8" "GU=2,8 HEAD=2,4"; do
  set -- $v; MIS_PROBE_BITS=8 MIS_QARR_${1} MIS_QARR_${2} timeout 200 env MIS_QARR_GU=${1#GU=} MIS_QARR_HEAD=${2#HEAD=} MIS_PROBE_BITS=8 python tools/qgemm_probe.py orpheus 32 2>/dev/null | tail -1 | sed "s/^/$v /"
done > $O/c5_qgemm_arr.txt 2>&1; cut -c1-400 $O/c5_qgemm_arr.txt
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/c5_bench_secondary.log 2>&1; tail -1 $O/c5_bench_secondary.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['step']['ms']); print(json.dumps(j.get('secondary'), indent=1)[:3000])"
