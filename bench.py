#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

Workload at N=1 (BASELINE.json configs[1]): the reference's default scene at 2560x1369, 64 spp
as 64 progressively accumulated frames of 1 ray/pixel, shipped bounce count. One "step" is
that whole 64-frame render: clear the accumulation image, 64 dispatches.

N>1 (sample split, weak scaling): every rank renders 64 frames of its own (rank r takes frames
r, r+N, r+2N, ...), so the image gets 64*N spp; the accumulation images are then summed with
one RCCL all-reduce and resolved. Scene and images are resident in HBM before timing starts.

metric: Mrays/s, where a ray is one GetHit call of the reference shader (primary, bounce or
shadow ray); the ray count of every frame is taken by an untimed counting pass.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak (spec)
VALU_PEAK_TFLOPS = 157.3        # MI355X FP32 vector peak (spec)


def cpu_baseline(arrays, pc_bytes_fn, W, H, seconds, threads):
    """Times the CPU oracle (a port of the reference shader, oracle/rt_oracle.c) on row bands of
    frame 0 of the same workload until `seconds` have elapsed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    band = 32
    rays = 0
    pixels = 0
    bands = 0
    frame = 0
    pc = np.frombuffer(pc_bytes_fn().tobytes(), dtype=O.PC_DTYPE).copy()
    t0 = time.time()
    done = False
    while not done:                       # whole frames 0, 1, 2, ... in row bands until the budget is spent
        pc["frame"] = frame
        for y0 in range(0, H, band):
            r = O.render(arrays, O.pc_bytes(pc), W, H, window=(0, y0, W, min(H, y0 + band)), counters=True, nthreads=threads)
            rays += int(r["rays"].sum())
            pixels += W * (min(H, y0 + band) - y0)
            bands += 1
            if time.time() - t0 >= seconds:
                done = True
                break
        frame += 1
    dt = time.time() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "frames 0..%d of the same workload in row bands of %d rows: %d bands, %d pixel samples, %d rays in %.1f s" % (frame - 1, band, bands, pixels, rays, dt),
            "msamples_per_s": pixels / dt / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=2560)
    ap.add_argument("--height", type=int, default=1369)
    ap.add_argument("--spp", type=int, default=64, help="frames per step per GPU")
    ap.add_argument("--fast", type=int, default=0, help="1 = FMA-contracted flavour (not bit-exact with the oracle)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 disables it")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--share-gpu", type=int, default=0, help="1 = every rank uses cuda:0 (rehearsal of the N > 1 flow on a one-GPU box, with --backend gloo; two processes on one GPU "
                         "hang now and then inside gloo's all_reduce of device tensors: run it under a timeout)")
    ap.add_argument("--pipeline", type=int, default=-1, help="RT_OPT_PIPELINE to measure (-1 = the library's default)")
    ap.add_argument("--trace", type=int, default=0, help="1 = stage markers on stderr")
    ap.add_argument("--single-dispatch", type=int, default=0,
                    help="1 = also time 64 spp as ONE dispatch with raysPerPixel = 64 (extra launches of the same kernel: off by "
                         "default so that a profile of the default run holds only the timed launches)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import raytracer_amd as R

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    def mark(what):
        if args.trace:
            print("[rank %d] %s" % (rank, what), file=sys.stderr, flush=True)

    if args.trace:
        import faulthandler
        faulthandler.dump_traceback_later(35, exit=False)
    W, H, spp = args.width, args.height, args.spp
    mark("process group ready")
    scene = R.Scene()
    scene.init_default()
    sb = scene.buffers()
    arrays = sb.to_arrays()
    r = R.Renderer(local_rank)
    r.upload_scene(sb)
    r.set_option(R.RT_OPT_FAST_MATH, args.fast)
    if args.pipeline >= 0:
        r.set_option(R.RT_OPT_PIPELINE, args.pipeline)
    pc = R.default_push_constants()
    R.camera_to_push_constants(R.default_camera(), W, H, pc)
    pc.parentBVHCount = 8
    from raytracer_amd import distributed as D
    target = D.GpuTarget(r, pc, W, H, dev)      # images are torch tensors, kernels run on a torch stream the target owns
    frames = D.frames_of_rank(rank, world, spp)

    # untimed counting pass: GetHit calls and the shader's node/triangle test counters per frame
    rays = node = tri = 0
    for f in frames:
        p = pc.copy()
        p.frame = f
        st = r.frame_stats(p)
        rays += st.rays; node += st.bvhNodeTests; tri += st.triangleTests
    mark("counting pass done")
    counts = torch.tensor([rays, node, tri], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(counts)
    total_rays = float(counts[0].item())

    def step():
        # clear, `spp` dispatches (frames rank, rank+N, ...), one all-reduce of the accumulation
        # image when N > 1, resolve
        D.render_sample_split(target, spp)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    mark("counts reduced")
    for _ in range(args.warmup):
        step()
    mark("warm-up enqueued")
    fence()
    mark("warm-up done")
    kernel_ms = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
        mark("step %d enqueued" % k)
        # HIP events recorded by the library around the 64 dispatches, on the stream they run on;
        # reading them back waits only for work that is already enqueued
        kernel_ms.append(r.last_render_ms()[0])
        mark("step %d kernels timed" % k)
    fence()
    mark("timed steps done")
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        # one launch of the kernel (default: rt_pairkernel, two samples per lane) renders `group` consecutive frames
        # (RT_OPT_FRAMES_PER_LAUNCH)
        group = r.get_option(R.RT_OPT_FRAMES_PER_LAUNCH)
        launches = (spp + group - 1) // group
        launch_ms = float(np.mean(kernel_ms)) / launches      # average duration of one launch of the megakernel
        bytes_per_launch = (40.0 * node + 48.0 * tri + 48.0 * W * H * spp) / launches     # rank 0's frames
        flops_per_launch = (21.0 * node + 45.0 * tri) / launches
        achieved_gbps = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s", "value": total_rays * args.steps / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "reference default scene (hard-coded InitializeScene: 8 objects, 8794 triangles; dragon_lowres.obj stands in for the undistributed dragon.obj)",
            "config": {"workload": "configs[1]: default scene %dx%d, %d spp per GPU as %d accumulated frames of 1 ray/pixel, maxBounces 3" % (W, H, spp, spp),
                       "width": W, "height": H, "frames_per_gpu": spp, "total_spp": spp * world,
                       "parallelism": "sample-split x%d (RCCL all-reduce of the accumulation image)" % world if distributed else "single GPU",
                       "flavour": "fast" if args.fast else "strict"},
            "ms_per_frame": ms_step / spp, "msamples_per_s": W * H * spp * world * args.steps / dt / 1e6,
            "rays_per_step": total_rays,
            "roofline": {"bound": "hbm", "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved_gbps / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": {0: "rt_megakernel", 3: "rt_pairkernel"}.get(r.get_option(R.RT_OPT_PIPELINE), "pipeline %d" % r.get_option(R.RT_OPT_PIPELINE)), "launch_ms": launch_ms, "frames_per_launch": group,
                         "launches_per_step": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "valu": {"achieved_tflops": flops_per_launch / (launch_ms * 1e-3) / 1e12, "peak_tflops": VALU_PEAK_TFLOPS,
                                  "frac": flops_per_launch / (launch_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS}},
        }
        if world == 1 and args.single_dispatch:
            # the other reading of "64 spp": ONE dispatch with raysPerPixel = 64 (samples seeded by the
            # sample index instead of the frame index, raytrace.comp:779-783), after the timed region
            p64 = pc.copy()
            p64.raysPerPixel = spp
            p64.frame = 0
            target.clear()
            r.render_frames(p64, 0, 1, 1)
            r.sync()
            r.render_frames(p64, 0, 1, 1)
            r.sync()
            out["ms_single_dispatch_%dspp" % spp] = r.last_render_ms()[0]
        if world == 1 and args.cpu_seconds > 0:
            threads = min(16, os.cpu_count() or 1)
            p0 = pc.copy()
            p0.frame = 0
            out["cpu_baseline"] = cpu_baseline(arrays, p0.as_bytes, W, H, args.cpu_seconds, threads)
        print(json.dumps(out), flush=True)
    target.close()
    r.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
