for d in 0 1 8; do
 for lib in raytracing_amd/libvrt.so raytracing_amd/libvrt_nobeam.so; do
  VRT_BENCH_DEPTH=$d VRT_LIB=$lib timeout -k 10 200 python3 bench.py --workload c3 --steps 3 --warmup 1 --min-time 0 --no-cpu-baseline --frames-8k 0 --opt beam=0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('depth $d $lib', round(d['ms_per_frame_kernel'],3))" || exit 1
 done
done
