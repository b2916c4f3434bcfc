for wl in c3 c5; do
 st=3; [ $wl = c5 ] && st=1
 for d in 0 8; do
  for lib in raytracing_amd/libvrt.so raytracing_amd/libvrt_noany.so; do
   VRT_BENCH_DEPTH=$d VRT_LIB=$lib timeout -k 10 200 python3 bench.py --workload $wl --steps $st --warmup 1 --min-time 0 --no-cpu-baseline --frames-8k 0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl depth $d $lib', round(d['ms_per_frame_kernel'],3), 'ms rays', d['config']['rays_per_frame'])" || exit 1
  done
 done
done
