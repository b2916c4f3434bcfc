#!/bin/bash
# VALU instruction count and lane activity of the frame kernel for several option sets (GPU box):
# tools/valu.sh <workload> <tag> "<opts A>" "<opts B>" ...
wl=$1; tag=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
k=0
for o in "$@"; do
  k=$((k+1))
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/p$k -- python3 bench.py --workload $wl --steps 2 --warmup 1 --min-time 0 --no-cpu-baseline --frames-8k 0 $o > $out/p$k.log 2>&1 || { tail -3 $out/p$k.log; exit 1; }
  python3 - "$out/p$k" "$o" <<'PY'
import csv, glob, collections, sys
f=glob.glob(sys.argv[1]+"/*/*counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_mega" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
n=len(agg["SQ_INSTS_VALU"]); m={k:sum(v)/max(n,1) for k,v in agg.items()}
print("[%s] VALU %.3fG SALU %.3fG lanes %.3f active_valu_cycles/busy %.3f" % (sys.argv[2], m["SQ_INSTS_VALU"]/1e9, m["SQ_INSTS_SALU"]/1e9, m["SQ_THREAD_CYCLES_VALU"]/(64*m["SQ_INSTS_VALU"]), m["SQ_ACTIVE_INST_VALU"]/max(m["SQ_BUSY_CYCLES"],1)))
PY
done
