#!/bin/bash
# Instruction-cache counters of the frame kernel: tools/exp/icache.sh <workload> <tag> "<opts A>" ... (GPU box)
wl=$1; tag=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
k=0
for o in "$@"; do
  k=$((k+1))
  timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/p$k -- python3 bench.py --workload $wl --steps 2 --warmup 1 --min-time 0 --no-cpu-baseline --frames-8k 0 $o > $out/p$k.log 2>&1 || { tail -3 $out/p$k.log; exit 1; }
  python3 - "$out/p$k" "$o" <<'PY'
import csv, glob, collections, sys
f=glob.glob(sys.argv[1]+"/*/*counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_mega" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in agg.items()}
print("[%s]" % sys.argv[2], " ".join("%s=%.4g" % kv for kv in sorted(m.items())))
PY
done
