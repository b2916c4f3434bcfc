#!/bin/bash
# Per-depth cost study of one workload (on the GPU box): frame time, nodes and primitives visited and
# lane activity with the path cut at depth 0, 1, 2, 3 and uncut.  tools/depth_study.sh <workload> <tag>
wl=$1; tag=$2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 8; do
  VRT_BENCH_DEPTH=$d VRT_LIB=$GRAFT_REPO_ROOT/raytracing_amd/libvrt_cnt.so timeout -k 10 100 python3 bench.py --workload $wl --steps 2 --warmup 1 --min-time 0 --no-cpu-baseline --frames-8k 0 "${@:3}" > $out/cnt_d$d.json || exit 1
  VRT_BENCH_DEPTH=$d timeout -k 10 100 python3 bench.py --workload $wl --steps 5 --warmup 2 --min-time 0 --no-cpu-baseline --frames-8k 0 "${@:3}" > $out/time_d$d.json || exit 1
  VRT_BENCH_DEPTH=$d timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/pmc_d$d -- python3 bench.py --workload $wl --steps 2 --warmup 1 --min-time 0 --no-cpu-baseline --frames-8k 0 "${@:3}" > $out/pmc_d$d.log 2>&1 || exit 1
done
python3 - <<PY
import json, csv, glob, collections
for d in (0,1,2,3,8):
    j=json.load(open("$out/cnt_d%d.json"%d))
    ms=json.load(open("$out/time_d%d.json"%d))["ms_per_frame_kernel"]
    f=glob.glob("$out/pmc_d%d/*/*counter_collection.csv"%d)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_mega" in r["Kernel_Name"] or "k_wf" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    n=len(agg["SQ_INSTS_VALU"]); m={k:sum(v)/max(n,1) for k,v in agg.items()}
    print("depth",d,"ms",round(ms,3),"rays",j["config"]["rays_per_frame"],j.get("traversal"),"VALU",m.get("SQ_INSTS_VALU"),"lanes",m.get("SQ_THREAD_CYCLES_VALU",0)/max(64*m.get("SQ_INSTS_VALU",1),1), "wait_any/wave_cycles", m.get("SQ_WAIT_INST_ANY",0)/max(m.get("SQ_WAVE_CYCLES",1),1))
PY
