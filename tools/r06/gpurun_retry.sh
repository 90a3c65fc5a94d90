# gpurun with retries while the pod's GPU slots are busy (exit 3 = nothing charged): usage gpurun_retry.sh <timeout> <command>
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout $1 -- "$2"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
