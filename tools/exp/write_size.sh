#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the frame kernel for one workload and several libraries (GPU box):
# tools/exp/write_size.sh <workload> lib1 lib2 ...
wl=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  for c in WRITE_SIZE FETCH_SIZE; do
    out=$GRAFT_REPO_ROOT/gpurun_out/ws_$(basename $lib .so)_$c
    rm -rf $out
    VRT_LIB=$GRAFT_REPO_ROOT/$lib timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out -- python3 bench.py --workload $wl --steps 3 --warmup 1 --min-time 0 --no-cpu-baseline --no-host-path --frames-8k 0 > $out.log 2>&1 || { tail -3 $out.log; exit 1; }
    python3 - "$out" "$lib" $c <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_mega" in r["Kernel_Name"]]
print("%s %s: %.3f GiB per launch (%d launches)" % (sys.argv[2], sys.argv[3], sum(v) / len(v) * 1024 / 2**30 * (2 if sys.argv[3] == "FETCH_SIZE" else 1), len(v)))
PY
  done
done
