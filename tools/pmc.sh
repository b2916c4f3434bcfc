#!/bin/bash
# Counter passes for one workload/library (on the GPU box): tools/pmc.sh <tag> <workload> [lib]
# Each pass is its own short run; the per-kernel means are printed by tools/pmc_summary.py.
tag=$1; wl=$2; lib=${3:-raytracing_amd/libvrt.so}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VRT_LIB=$GRAFT_REPO_ROOT/$lib
run() { # name counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$n -- python3 bench.py --workload $wl --steps 3 --warmup 1 --min-time 0 --no-cpu-baseline --no-host-path --frames-8k 0 > $out/$n.log 2>&1 || { echo "pass $n failed"; tail -5 $out/$n.log; exit 1; }
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run ta TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
python3 tools/pmc_summary.py $out
