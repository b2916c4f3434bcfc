#!/bin/bash
# A/B of libvrt builds on the GPU box: tools/ab.sh [workload] [steps] lib1 lib2 ... (3 interleaved rounds)
wl=$1; steps=$2; shift 2
for round in 1 2 3; do
  for lib in "$@"; do
    VRT_LIB=$lib timeout -k 10 120 python bench.py --workload $wl --steps $steps --warmup 5 --min-time 0 --no-cpu-baseline --frames-8k 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_frame_kernel'],4), 'ms', round(d['value'],1), 'Mrays/s')" || exit 1
  done
done
