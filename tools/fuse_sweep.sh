#!/bin/bash
# sweep of the fused-producer wait parameters on the bench (experiment)
for ps in 0 1 2; do
  MIS_FUSED_GLUE=1 MIS_GLUE_PRESLEEP=$ps MIS_GLUE_SPIN=200000 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null > /tmp/fb.json
  python -c "
import json; d=json.load(open('/tmp/fb.json')); print('presleep', $ps, round(d['value'],1), round(d['roofline']['step']['ms'],3))"
done
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null > /tmp/fb.json
python -c "
import json; d=json.load(open('/tmp/fb.json')); print('unfused', round(d['value'],1), round(d['roofline']['step']['ms'],3))"
