#!/bin/bash
# Instruction-cache and scalar-data-cache counters of the frame kernel (GPU box): tools/icache.sh <workload> <tag> [lib]
wl=$1; tag=$2; lib=${3:-raytracing_amd/libvrt.so}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export VRT_LIB=$GRAFT_REPO_ROOT/$lib
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d $out/ic -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 2 --warmup 1 --min-time 0 --no-cpu-baseline --no-host-path --frames-8k 0 > $out/ic.log 2>&1 || { tail -3 $out/ic.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - "$out/ic" <<'PY'
import csv, glob, collections, sys
f=glob.glob(sys.argv[1]+"/*/*counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_mega" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("%-32s %.4g" % (k, sum(v)/len(v)))
PY
