# round 6, GPU call 3: which of the three folds pays (wall clock of the transcribe call, alternating runs), and with which split factors
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
rm -f $O/c3_whisper_fold_variants.txt
for i in 1 2; do
  for v in "0 0 0" "2 0 0" "6 0 0" "7 0 0" "7 4 0" "7 4 2" "3 4 0" "0 4 0" "2 0 2"; do
    set -- $v
    export MIS_WHISPER_FOLD=$1; unset MIS_WS_FC2 MIS_WS_O
    [ "$2" != "0" ] && export MIS_WS_FC2=$2
    [ "$3" != "0" ] && export MIS_WS_O=$3
    echo "FOLD=$1 WS_FC2=$2 WS_O=$3 $(timeout 300 python tools/bench_whisper.py 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("transcribe_ms %.2f encode_ms %.2f crc %d" % (d["transcribe_ms"], d["encode_ms"], d["token_crc32"]))')" | tee -a $O/c3_whisper_fold_variants.txt
  done
done
