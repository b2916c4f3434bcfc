#!/bin/bash
# A/B of library options on the GPU box: tools/abopt.sh <workload> <steps> "<opts A>" "<opts B>" ... (3 interleaved rounds)
# each opts string is a list of bench.py flags, e.g. "--opt packet=0"
wl=$1; steps=$2; shift 2
for round in 1 2 3; do
  for o in "$@"; do
    timeout -k 10 200 python3 bench.py --workload $wl --steps $steps --warmup 2 --min-time 0 --no-cpu-baseline --frames-8k 0 $o | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$o]', round(d['ms_per_frame_kernel'],4), 'ms', round(d['value'],1), 'Mrays/s')" || exit 1
  done
done
