#!/bin/bash
# Runs the benchmark and the rocprofv3 passes whose summaries are kept under profiles/.
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 bench.py --steps 300 --warmup 30 > $out/bench_c2.json 2> $out/bench_c2.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c2 -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $out/trace_c2.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/pmc_write.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/pmc_sq.log 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --workload c3 --steps 5 --warmup 1 > $out/bench_c3.json 2> $out/bench_c3.err || exit 1
timeout -k 10 300 python3 bench.py --workload c4 --steps 10 --warmup 2 > $out/bench_c4.json 2> $out/bench_c4.err || exit 1
echo done
