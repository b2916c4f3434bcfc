#!/bin/bash
# Runs the benchmark and the rocprofv3 passes whose summaries are kept under profiles/.
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag> [workloads...]
# Afterwards, here: python tools/summarize_profiles.py gpurun_out/<tag> profiles r02
# (build raytracing_amd/libvrt_cnt.so first: tools/build_variant.sh cnt -DVRT_COUNT_NODES=1)
tag=$1; shift
wls=${@:-c2 c3 c4}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in $wls; do
  case $wl in c2) steps=300; warm=30;; c3) steps=5; warm=1;; c4) steps=20; warm=2;; *) steps=2; warm=1;; esac
  echo "== $wl bench"
  timeout -k 10 400 python3 bench.py --workload $wl --steps $steps --warmup $warm > $out/bench_$wl.json 2> $out/bench_$wl.err || { tail -5 $out/bench_$wl.err; exit 1; }
  if [ -f raytracing_amd/libvrt_cnt.so ]; then
    echo "== $wl nodes and primitives per ray (counting build)"
    VRT_LIB=$GRAFT_REPO_ROOT/raytracing_amd/libvrt_cnt.so timeout -k 10 300 python3 bench.py --workload $wl --steps 1 --warmup 1 --min-time 0 --no-cpu-baseline --no-host-path --frames-8k 0 > $out/cnt_$wl.json 2> $out/cnt_$wl.err || { tail -5 $out/cnt_$wl.err; exit 1; }
  fi
  echo "== $wl kernel trace"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$wl -- python3 bench.py --workload $wl --steps $steps --warmup $warm --min-time 0 --no-cpu-baseline --no-host-path --frames-8k 0 > $out/trace_$wl.log 2>&1 || { tail -5 $out/trace_$wl.log; exit 1; }
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== $wl $c"
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${wl}_$c -- python3 bench.py --workload $wl --steps 3 --warmup 1 --min-time 0 --no-cpu-baseline --no-host-path --frames-8k 0 > $out/pmc_${wl}_$c.log 2>&1 || { tail -5 $out/pmc_${wl}_$c.log; exit 1; }
  done
  echo "== $wl SQ passes"
  tools/pmc.sh $tag/sq_$wl $wl > $out/sq_$wl.txt 2>&1 || { tail -5 $out/sq_$wl.txt; exit 1; }
done
echo done
