# What bounds the streaming quantised GEMM (k_gemm_skinny_q) at Orpheus-3B width: the same launches timed in two diagnostics builds
# (make -C mlx-audio-swift_amd/csrc ablate) - loads only (every loaded register consumed by one xor, no conversion / MFMA / scale math)
# and math only (no loads in the K loop) - next to the product build.  gate_up and lm_head are the roles that stream; the others run
# the one-shot kernel, which the diagnostics builds leave alone.  Results: gpurun_out/qgemm_probe.jsonl ("lib" names the build).
set -x
cd ${GRAFT_REPO_ROOT:-.}
rm -f gpurun_out/qgemm_probe.jsonl
P=$PWD/mlx-audio-swift_amd
for lib in libmi_speech.so libmi_speech_qgemm_loads_only.so libmi_speech_qgemm_math_only.so; do
  MIS_LIB_PATH=$P/$lib MIS_PROBE_BITS=8,4 timeout 60 python tools/qgemm_probe.py orpheus 32 > /dev/null 2>&1
done
MIS_LIB_PATH=$P/libmi_speech_qgemm_loads_only.so MIS_PROBE_BITS=8 timeout 40 python tools/qgemm_probe.py orpheus 32 > /dev/null 2>&1
MIS_LIB_PATH=$P/libmi_speech.so MIS_PROBE_BITS=8 timeout 40 python tools/qgemm_probe.py orpheus 32 > /dev/null 2>&1
cat gpurun_out/qgemm_probe.jsonl
