#!/usr/bin/env python3
"""Benchmark of the per-pixel tracing loop on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4] [--mode auto|mega|wavefront]

A step = one frame of the workload rendered by libvrt with camera/scene resident on the GPU
and the framebuffer left in HBM.  At N = 1 the default workload is the teaching scene at
1920x1080, 1 spp, reflection depth 5.  With N > 1 (one process per GPU under
torch.distributed.run) the same 1920x1080 frame is rendered with N samples per pixel, cut into
8-row strips dealt round-robin to the ranks (a rank renders all samples of its pixels), and
the strips are gathered to rank 0 with RCCL and de-interleaved there: the rays per GPU are
those of the one-GPU frame, i.e. weak scaling.

Rank 0 prints one JSON line.  `value` is Mrays/s over all GPUs, a ray being one intersection
query (primary + reflection + transmission + shadow + ambient-occlusion; the reference's
redundant primary pre-check is not counted).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import raytracing_amd as ra  # noqa: E402
from raytracing_amd import abi, host, tiles  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X vector FP32 (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBPS = 8000.0


def build_workload(name, n_gpus):
    if name == "c2":
        # The frame stays 1920x1080 at every world size; N GPUs render N samples per pixel, so
        # the rays per GPU stay those of the one-GPU frame (weak scaling) while a rank's share
        # of the frame gather shrinks to 1/N of the 33 MB frame.  (Growing the pixel count with
        # N instead leaves every rank a 33 MB message for its one xGMI link to rank 0 per
        # 0.15 ms of rendering: the link, not the GPUs, would be what is measured.)
        w, h = (1920, 1080)
        scene = ra.teaching_scene()
        cam = ra.make_camera((1.45, 2.0, 2.5), (3.0, 1.55, 2.5 + (16 / 9) / 2), (0.0, 0.0, -16 / 9),
                             (0.0, 1.0, 0.0), w, h)
        spp = max(n_gpus, 1)
        kw = dict(spp=spp, max_depth=5, rng=30.0)
        label = f"teaching scene {w}x{h}, {spp} spp{' jittered' if spp > 1 else ''}, depth 5"
        offsets = None
        if spp > 1:
            offsets = host.sampler_table(abi.VRT_SAMPLER_JITTERED, spp, 83, (16 / 9) / w, 1.0 / h, seed=1)
            kw["seed"] = 1
    elif name == "c3":
        w, h = (3840, 2160)
        scene = ra.random_spheres_scene(10000)
        cam = ra.random_spheres_camera(w, h)
        offsets = host.sampler_table(abi.VRT_SAMPLER_JITTERED, 16, 83, 2.4 * (w / h) / w, 2.4 / h, seed=1)
        kw = dict(spp=16, max_depth=8, rng=200.0, seed=1)
        label = f"10k random spheres {w}x{h}, 16 spp jittered, depth 8"
    elif name == "c4":
        w, h = (1920, 1080)
        scene = ra.tessellated_mesh_scene(708)
        cam = ra.tessellated_mesh_camera(w, h)
        offsets = host.sampler_table(abi.VRT_SAMPLER_JITTERED, 4, 83, 2.0 * (w / h) / w, 2.0 / h, seed=1)
        kw = dict(spp=4, max_depth=5, rng=100.0, seed=1)
        label = f"1M-triangle mesh {w}x{h}, 4 spp jittered, depth 5"
    elif name == "c5":
        w, h = (7680, 4320)
        scene = ra.random_spheres_scene(10000)
        cam = ra.random_spheres_camera(w, h)
        offsets = host.sampler_table(abi.VRT_SAMPLER_JITTERED, 64, 83, 2.4 * (w / h) / w, 2.4 / h, seed=1)
        kw = dict(spp=64, max_depth=8, rng=200.0, seed=1)
        label = f"10k random spheres {w}x{h}, 64 spp jittered, depth 8"
    else:
        raise SystemExit(f"unknown workload {name}")
    if os.environ.get("VRT_BENCH_DEPTH"):  # study only: the same workload cut at another depth
        kw["max_depth"] = int(os.environ["VRT_BENCH_DEPTH"])
        label += f" [depth overridden to {kw['max_depth']}]"
    return scene, cam, offsets, kw, label


def algorithmic_flops(scene, rays, pixels):
    """FP32 operations the brute-force path needs for these ray counts: every query walks all
    primitives (17 per sphere, 30 per box, 40 per triangle), each shaded hit costs
    25 + 30 per light, each primary ray 12 for ray generation and each secondary miss 8 for the
    sky colour.  Shaded hits are bounded by primary + reflection + transmission rays."""
    per_query = 17 * len(scene.spheres) + 30 * len(scene.boxes) + 40 * len(scene.indices)
    queries = sum(int(r) for r in rays[1:])
    shaded = int(rays[abi.VRT_RAY_PRIMARY] + rays[abi.VRT_RAY_REFLECT] + rays[abi.VRT_RAY_TRANSMIT])
    return queries * per_query + shaded * (25 + 30 * len(scene.lights)) + pixels * 12 + shaded * 8


# Clocks (of the nominal 2.4 GHz) a SIMD needs per wave64 vector instruction: the hardware model's
# figure, and what tools/exp/valu_issue.hip sustained on an MI355X with 2-8 waves per SIMD
# (profiles/r03/valu_issue_probe.txt: 2.34-2.94 depending on the instruction mix).
VALU_CLOCKS_MODEL = 2.0
VALU_CLOCKS_MEASURED = 2.45
# Clocks a SIMD needs per scalar instruction (s_add_u32 chains in the same probe: one per 4.2 clocks
# with two or more waves resident, however many; scalar and vector instructions of different waves
# overlap).  Branches and scalar loads are issued by the same unit.
SALU_CLOCKS_MEASURED = 4.2


def profiled_counters(workload, prefix="r03"):
    """Issue-side view of the dominant kernel from the rocprofv3 summaries committed under
    profiles/ (not measured live): share of the chip's vector-ALU issue rate the kernel used and the
    mean fraction of the 64 lanes active per vector instruction.  The issue rate is a measured one
    (tools/exp/valu_issue.hip, profiles/r03/valu_issue_probe.txt): with two or more waves per SIMD a
    SIMD issues one wave64 v_fma_f32 per 2.6-2.9 clocks of the nominal 2.4 GHz (2.3-2.5 for a mix of
    FP and integer instructions; one wave alone: 5.9; v_pk_fma_f32: 4.4), i.e. the 2 clocks per
    instruction of a SIMD-32 at the clock the chip holds under that load.  `valu_issue_frac` prices
    an instruction at the 2 clocks of the hardware model, `valu_issue_frac_of_measured_peak` at the
    2.45 clocks the probe sustained.  `scalar_issue_frac_of_measured_peak` is the same for the scalar
    side (scalar ALU, branch and scalar-memory instructions at the 4.2 clocks per instruction and SIMD
    the probe measured): in these kernels it is the busier of the two units."""
    import csv
    for older in ("r02", "r01"):
        if os.path.exists(os.path.join(ROOT, "profiles", f"{prefix}_{workload}_pmc.csv")):
            break
        prefix = older
    try:
        pmc = {r["counter"]: float(r["mean_per_dispatch"])
               for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{prefix}_{workload}_pmc.csv")))}
        stats = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"{prefix}_{workload}_kernel_stats.csv"))))
        avg_ns = next(float(r["AverageNs"]) for r in stats if "k_mega" in r["Name"])
        insts, lanes = pmc["SQ_INSTS_VALU"], pmc["SQ_THREAD_CYCLES_VALU"]
        return {"source": f"profiles/{prefix}_{workload}_pmc.csv, profiles/{prefix}_{workload}_kernel_stats.csv",
                "kernel_avg_us": avg_ns / 1e3,
                "valu_instructions_per_launch": insts,
                "valu_issue_frac": insts * VALU_CLOCKS_MODEL / (1024 * 2.4e9 * avg_ns * 1e-9),
                "valu_issue_frac_of_measured_peak": insts * VALU_CLOCKS_MEASURED / (1024 * 2.4e9 * avg_ns * 1e-9),
                "lane_utilisation": lanes / (insts * 64.0),
                "scalar_instructions_per_launch": (pmc.get("SQ_INSTS_SALU", 0.0) + pmc.get("SQ_INSTS_BRANCH", 0.0)
                                                   + pmc.get("SQ_INSTS_SMEM", 0.0)),
                "scalar_issue_frac_of_measured_peak":
                    ((pmc.get("SQ_INSTS_SALU", 0.0) + pmc.get("SQ_INSTS_BRANCH", 0.0) + pmc.get("SQ_INSTS_SMEM", 0.0))
                     * SALU_CLOCKS_MEASURED / (1024 * 2.4e9 * avg_ns * 1e-9))}
    except Exception:
        return None


def tree_roofline(workload, scene, st, rays, queries, kernel_ms, fb_bytes):
    """What bounds the frame kernel of a scene traced through the 4-wide tree, from what the kernel
    itself incurs: per ray it enters `nodes_per_ray` wide nodes (128 B each, seven 16-byte loads per
    lane, four slab tests; a frontier node of a wave's beam counts once) and tests `prims_per_ray`
    primitives (measured with a -DVRT_COUNT_NODES=1 build, kept in profiles/traversal.json).  Those loads hit the vector L1 (98 %), so they are
    priced against the L1 rate, the measured HBM bytes against the HBM rate, and the arithmetic
    against the FP32 vector peak: the vector ALUs are the roof that is closest."""
    trav = None
    if st.nodes_visited:
        trav = {"nodes_per_ray": st.nodes_visited / max(queries, 1), "prims_per_ray": st.prims_tested / max(queries, 1),
                "source": "this run (counting build)"}
    else:
        try:
            trav = dict(json.load(open(os.path.join(ROOT, "profiles", "traversal.json")))[workload])
            trav["source"] = "profiles/traversal.json"
        except Exception:
            pass
    secs = kernel_ms * 1e-3
    n_sph, n_box, n_tri = len(scene.spheres), len(scene.boxes), len(scene.indices)
    n_prims = max(n_sph + n_box + n_tri, 1)
    prim_bytes = (16 * n_sph + 64 * n_box + 48 * n_tri) / n_prims
    prim_flops = (17 * n_sph + 30 * n_box + 40 * n_tri) / n_prims
    shaded = int(rays[abi.VRT_RAY_PRIMARY] + rays[abi.VRT_RAY_REFLECT] + rays[abi.VRT_RAY_TRANSMIT])
    roof = {"bound": "valu", "unit": "TFLOP/s", "peak": PEAK_FP32_TFLOPS, "traffic": None,
            "kernel": "k_mega<bvh>" if st.mode_used == abi.VRT_MODE_MEGAKERNEL else "k_wf_*", "kernel_ms": kernel_ms,
            "framebuffer_bytes_per_launch": fb_bytes}
    if trav:
        # a slab test: 6 subtractions, 6 multiplications, 10 min/max, 2 for the comparison = 24 per box
        flops = queries * (trav["nodes_per_ray"] * 4 * 24 + trav["prims_per_ray"] * prim_flops) \
            + shaded * (25 + 30 * len(scene.lights) + 8) + st.pixels * 12
        l1_bytes = queries * (trav["nodes_per_ray"] * 128 + trav["prims_per_ray"] * prim_bytes)
        ach = flops / secs / 1e12
        l1_peak = 256 * 64 * 2.4   # GB/s: 256 CUs x 64 B per clock x 2.4 GHz
        roof.update({"achieved": ach, "frac": ach / PEAK_FP32_TFLOPS, "algorithmic_flops_per_launch": flops,
                     "traversal": trav,
                     "l1": {"bytes_per_launch": l1_bytes, "achieved_GBps": l1_bytes / secs / 1e9, "peak_GBps": l1_peak,
                            "frac": l1_bytes / secs / 1e9 / l1_peak},
                     "note": "arithmetic of the box and primitive tests the kernel executes for this frame (bookkeeping "
                             "excluded) against the FP32 vector peak; node and primitive loads against the vector-L1 "
                             "rate; measured HBM bytes in `traffic`.  Camera rays and the occlusion rays of their hits "
                             "walk the tree once per wave (a frontier node counts once, not once per ray), so fewer "
                             "box tests are executed per ray than a per-ray traversal needs (2.8 instead of 16.5 nodes "
                             "on the 10k-sphere frame) and `frac` falls while the frame gets faster; `profiled` holds "
                             "the issue-slot share and lane activity of the committed counter passes, which is the "
                             "figure that says how busy the vector ALUs are"})
    else:
        roof.update({"achieved": None, "frac": None, "note": "profiles/traversal.json missing: no per-ray node counts"})
    return roof


def cpu_baseline(scene, cam, prm):
    """Times the CPU oracle (a restatement of the reference algorithm, not the Unity player) on
    this host: a bounded sub-frame once on one core, then once on every core this job may use."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    oracle.build()
    # bounded sample: the same camera and scene at a pixel count that keeps pixels x spp near
    # the 1080p single-sample frame (the whole frame for the default workload)
    scale = max(1.0, (cam.width * cam.height * prm.c.spp / 2.1e6) ** 0.5)
    sw, sh = max(int(cam.width / scale), 1), max(int(cam.height / scale), 1)
    scam = ra.make_camera(cam.origin[:], cam.p0[:], cam.x_axis[:], cam.y_axis[:], sw, sh)
    t0 = time.perf_counter()
    _, _, rays = oracle.render(scene, scam, prm.c, n_threads=1)
    t1 = time.perf_counter()
    # every core this job may use: the scheduler's affinity mask, capped by OMP_NUM_THREADS when the
    # box sets it (os.cpu_count() shows the whole machine there)
    nproc = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except Exception:
        allowed = nproc
    nthr = allowed
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        nthr = max(1, min(nthr, int(os.environ["OMP_NUM_THREADS"])))
    _, _, rays2 = oracle.render(scene, scam, prm.c, n_threads=nthr)
    t2 = time.perf_counter()
    q = float(sum(int(r) for r in rays[1:]))
    return {"value": q / (t1 - t0) / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"{sw}x{sh} sub-frame of the same camera and scene ({q:.0f} rays), oracle/vrt_oracle.c serial",
            "ms_per_sample_frame": (t1 - t0) * 1e3,
            "omp_value": q / (t2 - t1) / 1e6, "omp_cores": nthr,
            "nproc": nproc, "cores_allowed": allowed, "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS")}


def in_library_multi(n_gpus, workload, steps):
    """The same split behind the C ABI (vrt_create_multi: one host thread per GPU; every GPU stores
    the rows of its strips into the caller's page-locked frame buffer itself, over its own PCIe
    link), measured in a child process of its own after the ranks of this job have gone idle: the
    call a one-process host such as Unity makes.  Also timed with the strips gathered on the first
    GPU over xGMI and read back from there (option multi_gather).  Returns a dict or a short reason."""
    import subprocess
    code = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, %r)
import bench
import raytracing_amd as ra
from raytracing_amd import abi, host
n, wl, steps = %d, %r, %d
scene, cam, offsets, kw, label = bench.build_workload(wl, n)
one = host.Context(0)
one.set_scene(scene); one.set_camera(cam)
prm = ra.Params(offsets=offsets, **kw)
want, want_hit, st1 = one.render(prm)
one.close()
m = host.Context(devices=list(range(n)))
m.set_scene(scene); m.set_camera(cam)
rgba, hit = m.host_alloc(want.shape, np.float32), m.host_alloc(want_hit.shape, np.int32)
res = {"workload": label, "gpus": n}
for gather in (0, 1):
    m.set_option("multi_gather", gather)
    rgba[:] = 0; hit[:] = -9
    m.render(prm, rgba=rgba, hit=hit)
    m.render(prm, rgba=rgba, hit=hit)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter(); _, _, st = m.render(prm, rgba=rgba, hit=hit); ts.append((time.perf_counter() - t0) * 1e3)
    q = float(sum(st.rays[1:]))
    r = {"ms_per_frame_host": float(np.median(ts)), "gpu_ms": st.gpu_ms, "Mrays_per_s": q / float(np.median(ts)) / 1e3,
         "frame_equals_one_gpu": bool(np.array_equal(rgba, want, equal_nan=True) and np.array_equal(hit, want_hit)),
         "rays_equal_one_gpu": list(st.rays) == list(st1.rays)}
    if gather == 0:
        res.update(r)
        res["delivery"] = "every GPU stores its strips into the page-locked frame buffer over its own PCIe link"
    else:
        res["gathered_on_first_gpu"] = r
print(json.dumps(res))
""" % (ROOT, n_gpus, workload, steps)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
        if out.returncode != 0:
            return {"error": (out.stderr or out.stdout)[-400:]}
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:   # the headline measurement must not depend on this
        return {"error": repr(e)[-400:]}


def scaling_8k(local_rank, world, rank, dev, cdev, rehearsal, frames, tile_rows, mode):
    """The configuration the scaling target is stated on: 10k random spheres at 7680x4320,
    64 samples per pixel, depth 8, ONE frame of fixed size split into 8-row strips over the
    ranks (strong scaling) and gathered to rank 0 with RCCL.  Timed like the main loop: barrier
    and device synchronisation on both sides, maximum over ranks.  Returns a dict on rank 0."""
    scene, cam, offsets, kw, label = build_workload("c5", world)
    if world > 1:
        scene = tiles.broadcast_scene(scene if rank == 0 else ra.Scene(), cdev)
    W, H = cam.width, cam.height
    ctx = host.Context(local_rank)
    ctx.set_scene(scene)
    ctx.set_camera(cam)
    prm = ra.Params(offsets=offsets, mode=mode, tile_index=rank, tile_count=world, tile_rows=tile_rows,
                    compact_rows=True, **kw)
    fg = tiles.FrameGather(H, W, 4, world, rank, tile_rows, cdev)
    local = fg.local if not rehearsal else torch.empty_like(fg.local, device=dev)
    stream = torch.cuda.current_stream()

    def frame():
        ctx.render_device(prm, local.data_ptr(), None, stream.cuda_stream)
        if rehearsal:
            fg.local.copy_(local)
        return fg.gather()

    # warm-up at 1 sample per pixel (same kernels, same gather), then the timed full-cost frames
    wprm = ra.Params(offsets=np.ascontiguousarray(offsets[:, :1]), mode=mode, tile_index=rank, tile_count=world, tile_rows=tile_rows,
                     compact_rows=True, **dict(kw, spp=1))
    ctx.render_device(wprm, local.data_ptr(), None, stream.cuda_stream)
    fg.gather()
    # one untimed frame at full cost: the first frame of a layout measures the tile costs the later
    # ones are dispatched by, and the timed frames should be the steady state of a viewer
    frame()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        final = frame()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    my_q = float(sum(int(r) for r in ctx.fetch_stats().rays[1:]))
    # render-only time of this rank's strips (events on the launch stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    ctx.render_device(prm, local.data_ptr(), None, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    render_ms = e0.elapsed_time(e1)
    tot = torch.tensor([elapsed, render_ms, my_q], dtype=torch.float64, device=cdev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, render_ms, queries = float(mx[0]), float(mx[1]), float(sm[2])
    else:
        queries = my_q
    out = None
    if rank == 0:
        out = {"workload": label, "scaling": "strong", "frames": frames, "frame": [W, H],
               "rays_per_frame": queries, "ms_per_frame": elapsed * 1e3 / frames,
               "value": queries * frames / elapsed / 1e6, "unit": "Mrays/s",
               "slowest_rank_render_ms": render_ms,
               "gather": "rccl gather of the packed strips to rank 0 + de-interleave, inside the timed region"
                         if world > 1 else "none",
               "note": "parallel efficiency at N GPUs = value(N) / (N * value(1)) of this object"}
        if world > 1:
            # the gathered frame must equal the frame one GPU renders alone, bit for bit
            whole = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
            wp = ra.Params(offsets=offsets, mode=mode, **kw)
            ctx.render_device(wp, whole.data_ptr(), None, stream.cuda_stream)
            torch.cuda.synchronize()
            out["gathered_frame_equals_single_gpu_frame"] = bool(
                torch.equal(torch.nan_to_num(final.to(dev)), torch.nan_to_num(whole)))
            del whole
    if world > 1:
        dist.barrier()
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--mode", default="auto", choices=["auto", "mega", "wavefront"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="on one GPU: time each rank's strips of an N-way split of the workload's own "
                         "frame one after the other (strong-scaling load balance without N GPUs)")
    ap.add_argument("--tile-rows", type=int, default=8, help="rows per strip of the multi-GPU split")
    ap.add_argument("--record", type=int, default=0, help="also record the ray tree of this many pixels")
    ap.add_argument("--frames-8k", type=int, default=2,
                    help="frames of the 8K x 64 spp strong-scaling measurement reported as `scaling_8k` "
                         "(0 = skip it)")
    ap.add_argument("--no-in-library-multi", action="store_true",
                    help="with N > 1: skip the measurement of vrt_create_multi (one process driving all N GPUs)")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the vrt_render measurements into host memory (their frame kernels store over PCIe and "
                         "would mix into a profiler's per-kernel averages)")
    ap.add_argument("--min-time", type=float, default=0.5,
                    help="the timed region of --steps steps is repeated until this many seconds are timed")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="study option of the library (vrt_set_option), e.g. --opt packet=0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("VRT_BENCH_ONE_GPU"):  # rehearsal of the multi-rank code path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rehearsal = bool(os.environ.get("VRT_BENCH_ONE_GPU")) and world > 1
    if world > 1:
        if rehearsal:  # RCCL refuses two ranks on one GPU: gloo over host copies, control flow only
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if rehearsal else dev

    scene, cam, offsets, kw, label = build_workload(args.workload, world)
    if world > 1:
        # the scene description lives on rank 0 (the host that marshalled it) and is replicated
        # to the other ranks before each uploads it to its own GPU
        scene = tiles.broadcast_scene(scene if rank == 0 else ra.Scene(), cdev)
    mode = {"auto": abi.VRT_MODE_AUTO, "mega": abi.VRT_MODE_MEGAKERNEL, "wavefront": abi.VRT_MODE_WAVEFRONT}[args.mode]
    # With more than one GPU the strips travel as 3 floats per pixel, the averaged radiance before
    # tone mapping (12 instead of 16 bytes per pixel over a rank's one xGMI link to rank 0), and
    # rank 0 tone-maps the gathered frame: the same arithmetic on the same values, one step later.
    linear = world > 1
    prm_kw = dict(kw, flags=kw.get("flags", 0) | abi.VRT_OUT_LINEAR_RGB) if linear else kw
    prm = ra.Params(offsets=offsets, mode=mode, tile_index=rank, tile_count=world, tile_rows=args.tile_rows,
                    compact_rows=True, **prm_kw)
    ctx = host.Context(local_rank)
    for o in args.opt:
        name, _, value = o.partition("=")
        ctx.set_option(name, int(value))
    ctx.set_scene(scene)
    ctx.set_camera(cam)

    W, H = cam.width, cam.height
    # two frame buffers: while frame k is being gathered (RCCL, its own stream) frame k+1 renders
    fgs = [tiles.FrameGather(H, W, 3 if linear else 4, world, rank, args.tile_rows, cdev)
           for _ in range(2 if world > 1 else 1)]
    frames_rgba = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) if (linear and rank == 0) else None
                   for _ in fgs]
    staged = [None] * len(fgs)
    fg = fgs[0]
    local = fg.local if not rehearsal else torch.empty_like(fg.local, device=dev)
    local_hit = torch.empty((fg.rows_max, W), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    pending = [None, None]
    frame_no = [0]

    # Assembling a gathered frame (de-interleave + tone mapping, ~60 MB of traffic on rank 0) runs on
    # a stream of its own so that it overlaps rank 0's rendering instead of queueing behind it; the
    # render stream only waits for the gather itself before it reuses a strip buffer.
    side = torch.cuda.Stream(device=dev) if world > 1 else None
    gathered_ev = [torch.cuda.Event() for _ in fgs]
    assembled_ev = [torch.cuda.Event() for _ in fgs]

    def finish(k):
        # wait for the gather of the frame in buffer k, de-interleave and tone-map it on rank 0
        with torch.cuda.stream(side):
            handle, pending[k] = pending[k], None
            handle.wait()
            gathered_ev[k].record(side)
            if rank == 0:
                fgs[k].deinterleave()
            if frames_rgba[k] is not None:
                staged[k] = fgs[k].final if not rehearsal else fgs[k].final.to(dev)
                ctx.tonemap_device(staged[k].data_ptr(), frames_rgba[k].data_ptr(), H * W, side.cuda_stream)
            assembled_ev[k].record(side)
        stream.wait_event(gathered_ev[k])   # the strip buffer `local` of slot k may be rendered into again

    def step():
        k = frame_no[0] % len(fgs)
        frame_no[0] += 1
        if world > 1 and pending[k] is not None:
            finish(k)  # the frame rendered two steps ago
        target = local if rehearsal else fgs[k].local
        ctx.render_device(prm, target.data_ptr(), local_hit.data_ptr(), stream.cuda_stream)
        if rehearsal:
            fgs[k].local.copy_(local)
        if world > 1:
            # the next gather into slot k's `gathered` buffer must not start while the side stream
            # still de-interleaves the previous frame out of it
            stream.wait_event(assembled_ev[k])
            pending[k] = fgs[k].gather_async()  # RCCL gather of the packed strips to rank 0

    def drain():
        for k in range(len(fgs)):
            if pending[k] is not None:
                finish(k)

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    frames_match = None
    if world > 1:
        # once, outside the timed region: the gathered frame must equal the frame one GPU renders alone
        step()
        drain()
        torch.cuda.synchronize()
        if rank == 0:
            whole = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
            wp = ra.Params(offsets=offsets, mode=mode, **kw)
            ctx.render_device(wp, whole.data_ptr(), None, stream.cuda_stream)
            torch.cuda.synchronize()
            got = frames_rgba[(frame_no[0] - 1) % len(fgs)]
            frames_match = bool(torch.equal(torch.nan_to_num(got), torch.nan_to_num(whole)))
            del whole, got
        ctx.render_device(prm, (local if rehearsal else fgs[0].local).data_ptr(), local_hit.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
    st = ctx.fetch_stats()
    rays = np.array(st.rays[:], np.int64)
    my_queries = int(rays[1:].sum())

    # kernel-only time of the render launches (events on the launch stream, no gather)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    for _ in range(args.steps):
        ctx.render_device(prm, local.data_ptr(), local_hit.data_ptr(), stream.cuda_stream)
    ev1.record(stream)
    torch.cuda.synchronize()
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)

    def timed_region():
        """EXACTLY args.steps steps between barrier + device synchronisation on both sides."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        r0.record(stream)
        for _ in range(args.steps):
            step()
        r1.record(stream)
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        region_kernel_ms.append(r0.elapsed_time(r1) / max(args.steps, 1))
        t = torch.tensor([el], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)   # the slowest rank's clock, the same on every rank
        return float(t[0])

    # The region is repeated until at least --min-time seconds have been timed (a 1080p frame takes
    # 0.15 ms: a few dozen steps are over before the clocks have settled); a step stays one frame,
    # and the figure reported is the median region.
    region_kernel_ms = []   # HIP events on the launch stream around the same steps, per region
    regions = [timed_region()]
    n_rep = int(min(max(1, np.ceil(args.min_time / max(regions[0], 1e-9))), 200))
    for _ in range(n_rep - 1):
        regions.append(timed_region())
    elapsed = float(np.median(regions))
    if world == 1:
        # one GPU: a region's steps are the render launches and nothing else, so the kernel time that
        # goes with `value` is the one the events measured over the very same launches
        kernel_ms = float(np.median(region_kernel_ms))

    tot = torch.tensor([float(my_queries)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    queries = float(tot[0])

    s8k = None
    if args.frames_8k > 0 and args.workload == "c2":
        s8k = scaling_8k(local_rank, world, rank, dev, cdev, rehearsal, args.frames_8k, args.tile_rows, mode)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        flops = algorithmic_flops(scene, rays, st.pixels)
        fb_bytes = st.pixels * 20
        if ctx.fetch_stats().bvh_nodes == 0:
            ach = flops / (kernel_ms * 1e-3) / 1e12
            roof = {"bound": "valu", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_FP32_TFLOPS, "traffic": None,
                    "kernel": "k_mega<brute>" if st.mode_used == abi.VRT_MODE_MEGAKERNEL else "k_wf_*",
                    "kernel_ms": kernel_ms, "algorithmic_flops_per_launch": flops,
                    "framebuffer_bytes_per_launch": fb_bytes,
                    "hbm_achieved_GBps": fb_bytes / (kernel_ms * 1e-3) / 1e9,
                    "note": "instruction-issue bound, neither HBM nor MFMA (no dense contraction, primitives in "
                            "scalar registers): achieved / peak prices the arithmetic of the intersection tests and "
                            "of shading against the MI355X FP32 vector rate; `profiled` holds the measured shares of "
                            "the vector and of the scalar issue rate of a SIMD (the scalar unit is the busier one)"}
        else:
            roof = tree_roofline(args.workload, scene, st, rays, my_queries, kernel_ms, fb_bytes)
        prof = profiled_counters(args.workload)
        if prof:
            roof["profiled"] = prof
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                roof["traffic"] = json.load(open(tpath)).get(args.workload)
            except Exception:
                pass
        if roof.get("traffic"):
            roof["hbm_measured_GBps"] = roof["traffic"] / (kernel_ms * 1e-3) / 1e9
            roof["hbm_measured_frac"] = roof["hbm_measured_GBps"] / PEAK_HBM_GBPS
        out = {
            # BASELINE.json's metric is quoted on the default workload; the other workloads name their own frame
            "metric": "Mrays/s (primary+secondary) at 1080p depth 5" if args.workload == "c2"
                      else f"Mrays/s (primary+secondary), {label}",
            "value": queries * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "timing": {"regions": len(regions), "steps_per_region": args.steps,
                       "ms_per_step_min": min(regions) * 1e3 / args.steps,
                       "ms_per_step_median": ms_per_step,
                       "ms_per_step_max": max(regions) * 1e3 / args.steps},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": label, "frame": [W, H], "rays_per_frame": queries,
                       "mode": "wavefront" if st.mode_used == abi.VRT_MODE_WAVEFRONT else "megakernel",
                       "gather": "rccl gather of 3-float radiance strips to rank 0, tone mapping after the gather, "
                                 "overlapped with the next frame" if world > 1 else "none",
                       "scene": "broadcast from rank 0" if world > 1 else "local",
                       "rays_by_class": {n: int(r) for n, r in zip(abi.RAY_CLASS_NAMES, rays)}},
            "ms_per_frame_kernel": kernel_ms,
            "roofline": roof,
        }
        if st.bvh_nodes:
            out["config"]["bvh_build_ms"] = st.bvh_build_ms   # first build of the process (module load, first allocations)
            ctx.set_scene(scene)                              # the same scene again: what a host that edits its scene pays
            out["config"]["bvh_rebuild_ms"] = ctx.fetch_stats().bvh_build_ms
        if st.nodes_visited:  # only a library built with -DVRT_COUNT_NODES=1 counts these
            out["traversal"] = {"nodes_per_ray": st.nodes_visited / max(my_queries, 1),
                                "prims_per_ray": st.prims_tested / max(my_queries, 1)}
        if frames_match is not None:
            out["gathered_frame_equals_single_gpu_frame"] = frames_match
        if s8k is not None:
            out["scaling_8k"] = s8k
        if world == 1 and not args.no_host_path:
            # The boundary call itself: vrt_render with the frame delivered to host memory, i.e. the
            # `ms/frame` a Unity host sees.  Page-locked buffers of the library (vrt_host_alloc, what
            # the adapter gives PixelColorData) and, for reference, pageable numpy arrays.  Never `value`.
            hp = ra.Params(offsets=offsets, mode=mode, **kw)

            def host_ms(rgba, hit, n, prm=None):
                prm = prm or hp
                ctx.render(prm, rgba=rgba, hit=hit, want_hit=hit is not None)
                ts = []
                for _ in range(n):
                    t0 = time.perf_counter()
                    ctx.render(prm, rgba=rgba, hit=hit, want_hit=hit is not None)
                    ts.append((time.perf_counter() - t0) * 1e3)
                return float(np.median(ts))

            n_host = int(min(50, max(3, 200.0 / max(kernel_ms, 0.05))))
            p_rgba, p_hit = ctx.host_alloc((H, W, 4), np.float32), ctx.host_alloc((H, W), np.int32)
            out["ms_per_frame_host"] = host_ms(p_rgba, p_hit, n_host)
            out["ms_per_frame_host_no_hit_ids"] = host_ms(p_rgba, None, n_host)
            # the frame as the Color32 per pixel the player's RGBA32 texture holds anyway (VRT_OUT_RGBA8)
            hp8 = ra.Params(offsets=offsets, mode=mode, **dict(kw, flags=kw.get("flags", 0) | abi.VRT_OUT_RGBA8))
            p_rgba8 = ctx.host_alloc((H, W, 4), np.uint8)
            rgba8_ms = host_ms(p_rgba8, None, n_host, hp8)
            rgba8_ids_ms = host_ms(p_rgba8, p_hit, n_host, hp8)
            rgba8_pageable_ms = host_ms(np.zeros((H, W, 4), np.uint8), None, min(n_host, 15), hp8)
            ctx.host_free(p_rgba8)
            ctx.host_free(p_hit)
            ctx.host_free(p_rgba)
            h_rgba = np.zeros((H, W, 4), np.float32)
            h_hit = np.zeros((H, W), np.int32)
            pageable = host_ms(h_rgba, h_hit, min(n_host, 15))
            out["host_buffer_path"] = {
                "what": "vrt_render incl. device-to-host copy of the frame (16 B/pixel) and the hit ids (4 B/pixel)",
                "ms_per_frame_page_locked": out["ms_per_frame_host"],
                "ms_per_frame_page_locked_no_hit_ids": out["ms_per_frame_host_no_hit_ids"],
                "ms_per_frame_pageable": pageable,
                "Mrays_per_s_page_locked": queries / out["ms_per_frame_host"] / 1e3,
                "rgba8": {"what": "the same call with VRT_OUT_RGBA8: 4 B/pixel, the Color32 of the player's RGBA32 texture",
                          "ms_per_frame_page_locked_no_hit_ids": rgba8_ms, "ms_per_frame_page_locked": rgba8_ids_ms,
                          "ms_per_frame_pageable_no_hit_ids": rgba8_pageable_ms,
                          "Mrays_per_s_page_locked_no_hit_ids": queries / rgba8_ms / 1e3}}
        if world == 1 and args.emulate_ranks > 1:
            n = args.emulate_ranks
            per_rank, per_rank_rays = [], []
            for k in range(n):
                rp = ra.Params(offsets=offsets, mode=mode, tile_index=k, tile_count=n, tile_rows=args.tile_rows,
                               compact_rows=True, **kw)
                ctx.render_device(rp, local.data_ptr(), local_hit.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(max(args.steps, 1)):
                    ctx.render_device(rp, local.data_ptr(), local_hit.data_ptr(), stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize()
                per_rank.append(e0.elapsed_time(e1) / max(args.steps, 1))
                per_rank_rays.append(int(sum(ctx.fetch_stats().rays[1:])))
            out["emulated_strong_scaling"] = {
                "ranks": n, "ms_per_rank": per_rank, "rays_per_rank": per_rank_rays,
                "one_gpu_full_frame_ms": kernel_ms,
                "predicted_parallel_efficiency_render_only": kernel_ms / (n * max(per_rank)),
                "note": "each rank's strips rendered alone on this GPU; excludes the RCCL gather"}
        if world == 1 and args.record > 0:
            # Record mode (the consumer is the polyline of RayTracerUnity.cs:620-654): the same frame call
            # with the ray trees of a bounded set of pixels recorded.  Timed against the same call without
            # records, both into buffers that have been written before, so that the difference is the
            # recording pass itself (kernel + copies of the segments), not first-touch page faults.
            rng = np.random.default_rng(0)
            px = rng.integers(0, W * H, args.record).astype(np.int32)
            cap = 256
            rp = ra.Params(offsets=offsets, mode=mode, **kw)
            segs, counts = rp.record(px, cap=cap)
            plain = ra.Params(offsets=offsets, mode=mode, **kw)
            r_rgba, r_hit = ctx.host_alloc((H, W, 4), np.float32), ctx.host_alloc((H, W), np.int32)
            ctx.render(rp, rgba=r_rgba, hit=r_hit)
            t_plain, t_rec = [], []
            for _ in range(3):
                t0 = time.perf_counter(); ctx.render(plain, rgba=r_rgba, hit=r_hit); t_plain.append((time.perf_counter() - t0) * 1e3)
                t0 = time.perf_counter(); ctx.render(rp, rgba=r_rgba, hit=r_hit); t_rec.append((time.perf_counter() - t0) * 1e3)
            ctx.host_free(r_hit); ctx.host_free(r_rgba)
            out["record_mode"] = {"pixels": int(args.record), "cap": cap, "segments": int(counts.sum()),
                                  "overflowed_pixels": int((counts > cap).sum()),
                                  "frame_ms": float(np.median(t_plain)), "frame_plus_records_ms": float(np.median(t_rec)),
                                  "records_ms": float(np.median(t_rec) - np.median(t_plain))}
        if world > 1 and not rehearsal and not args.no_in_library_multi:
            # every rank is idle here (rank 0 is past the last barrier of the timed work and the others
            # wait in the barrier below), so the GPUs are free for the one-process form
            out["in_library_multi"] = in_library_multi(world, args.workload, 20 if args.workload == "c2" else 3)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, cam, prm)
            out["gpu_over_cpu_serial"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.barrier()   # the other ranks stay idle until rank 0 has finished its extra measurements
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
