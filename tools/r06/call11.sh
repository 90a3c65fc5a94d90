# round 6, GPU call 11: wave-state and LDS counters of the Whisper encoder's kernels (why k_gemm_big3 sits at 0.3 of the MFMA peak)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/p1 -- python $OLDPWD/tools/pmc_codec_probe.py whisper > /tmp/p1.log 2>&1; tail -3 /tmp/p1.log)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/p2 -- python $OLDPWD/tools/pmc_codec_probe.py whisper > /tmp/p2.log 2>&1; tail -3 /tmp/p2.log)
python tools/pmc_wave_states.py /tmp/p1 /tmp/p2 $O/c11_whisper_encoder_wave_states.json
