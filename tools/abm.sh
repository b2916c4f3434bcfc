#!/bin/bash
# like ab.sh with a mode argument: tools/abm.sh workload steps mode lib...
wl=$1; steps=$2; mode=$3; shift 3
for round in 1 2; do
  for lib in "$@"; do
    VRT_LIB=$lib timeout -k 10 120 python bench.py --workload $wl --steps $steps --warmup 2 --mode $mode --min-time 0 --no-cpu-baseline --frames-8k 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$mode', round(d['ms_per_frame_kernel'],4), 'ms', round(d['value'],1), 'Mrays/s')" || exit 1
  done
done
