#!/bin/bash
# HBM traffic of the decode-step kernels from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 passes (they do not fit one pass), --kernel-trace only (no sys/hip/hsa trace domains together with --pmc).
#   bash tools/pmc_traffic.sh r02      -> gpurun_out/r02_pmc/{fetch,write}/..., gpurun_out/r02_pmc/traffic.json
set -e
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_pmc
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/fetch -- python tools/pmc_probe.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/write -- python tools/pmc_probe.py > $OUT/write.log 2>&1
python tools/pmc_reduce.py $OUT/fetch $OUT/write $OUT/traffic.json
# keep the merged-back directory small: per-dispatch CSVs of the step kernels only
find $OUT -name "*.csv" -size +8M -delete
