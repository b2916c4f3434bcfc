#!/bin/bash
# tools/shift_sweep.sh workload steps shift... : kernel time per frame for each pixel/sample lane split
wl=$1; steps=$2; shift 2
for sh in "$@"; do
  VRT_SAMPLE_SHIFT=$sh timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 2 --min-time 0 --no-cpu-baseline --frames-8k 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl shift $sh', round(d['ms_per_frame_kernel'],4), 'ms', round(d['value'],1), 'Mrays/s')" || exit 1
done
