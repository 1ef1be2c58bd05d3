#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

Default workload (BASELINE.json configs[1]): the reference's default scene at 2560x1369, 64 spp as 64
progressively accumulated frames of 1 ray/pixel, shipped bounce count. One "step" is that whole render:
clear the accumulation image, the dispatches, (N > 1: one merge), resolve. Scene and images are resident
in HBM before timing starts.

  --config 1   default scene 2560x1369, 64 frames             (headline; N > 1: sample split)
  --config 2   generated 100 352-triangle height field, 1920x1080, 16 frames
  --config 3   default scene 3840x2160, 1024 frames           (N > 1: sample split; use --scaling strong)
  --config 4s  generated 999 424-triangle mixed-material scene, 1920x1080, 16 frames (N > 1: tile split)
  --scaling weak    every rank renders the config's frame count (sample split) - N GPUs give N times the spp
  --scaling strong  the config's frames are divided among the ranks (frames / N each)
  (tile split always divides the image, i.e. is strong scaling)

metric: Mrays/s, where a ray is one GetHit call of the reference shader (primary, bounce or shadow ray),
counted by an untimed pass of the counting kernel, which walks the reference's trees in the reference's
order - so the number is the reference's work per second, whatever the timed kernel skips.

Two traversal flavours are timed in the same run (N = 1): "near-first" (RT_OPT_TRAVERSAL 1: rebuilt SAH
trees, nearer child first; within the stated tolerance of the oracle, > 99.99 % of pixels bit-identical)
is the headline, "reference-order" (RT_OPT_TRAVERSAL 0: bit-exact with the oracle) is reported beside it.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak (spec)
VALU_PEAK_TFLOPS = 157.3        # MI355X FP32 vector peak (spec)


class Watchdog:
    """Ends the process with a non-zero status when a stage does not finish in time, instead of sitting
    in a collective for ever. (No re-exec: the process exits; a driver starts a fresh one.)"""

    def __init__(self, rank):
        self.rank = rank
        self.deadline = None
        self.what = ""
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, seconds, what):
        self.what = what
        self.deadline = time.monotonic() + seconds

    def disarm(self):
        self.deadline = None

    def _run(self):
        while True:
            time.sleep(1.0)
            d = self.deadline
            if d is not None and time.monotonic() > d:
                import faulthandler
                sys.stderr.write("[rank %d] bench.py watchdog: '%s' did not finish in time; giving up\n" % (self.rank, self.what))
                faulthandler.dump_traceback(file=sys.stderr)
                sys.stderr.flush()
                os._exit(3)


def cpu_baseline(arrays, pc_bytes, W, H, seconds, threads, host_cpus):
    """Times the CPU oracle (a port of the reference shader, oracle/rt_oracle.c) on row bands of
    frames 0, 1, ... of the same workload until `seconds` have elapsed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    band = 64
    rays = pixels = bands = frame = 0
    pc = np.frombuffer(pc_bytes.tobytes(), dtype=O.PC_DTYPE).copy()
    t0 = time.time()
    done = False
    while not done:
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
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "host_cpus": host_cpus,
            "sample": "frames 0..%d of the same workload in row bands of %d rows: %d bands, %d pixel samples, %d rays in %.1f s on %d threads "
                      "(the box reports %d CPUs; a one-GPU job is given a share of 16)" % (frame - 1, band, bands, pixels, rays, dt, threads, host_cpus),
            "msamples_per_s": pixels / dt / 1e6}


KERNEL_SOURCES = ("rt_kernels.hip", "rt_team.inl", "rt_device.h", "rt_bvh.cpp", "rt_host.cpp")


def kernel_source_hash():
    """sha256 over the sources that decide what the dominant kernel does and touches: the kernels, the device
    layout, the tree builder (its constants shape the trees) and the host code that lays the scene out."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "raytracer_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def measured_profile(kernel, config, traversal, frames_per_launch):
    """-> (HBM bytes per launch, PMC summary dict) of `kernel` from profiles/hbm_traffic.json: rocprofv3
    FETCH_SIZE x 2 + WRITE_SIZE (separate --pmc passes, scripts/traffic.sh) and, when the same entry carries
    them, the issue statistics of the SQ passes (scripts/pmc_cfg.sh). An entry only counts when it was taken
    from the sources this run uses (kernel_source_hash) in the same configuration; otherwise (None, None)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        sha = kernel_source_hash()
        for e in json.load(open(path)).get("entries", []):
            if (e.get("kernel"), str(e.get("config")), e.get("traversal"), e.get("frames_per_launch"), e.get("sources_sha256")) == \
                    (kernel, str(config), traversal, frames_per_launch, sha):
                return e.get("bytes_per_launch"), e.get("pmc")
    except Exception:
        pass
    return None, None


def lavapipe_probe():
    """The reference's own shader on a CPU Vulkan driver (lavapipe) would be the other CPU baseline. This only
    looks whether a Vulkan loader and a software ICD exist on this machine and says what it found; it runs
    nothing of the reference's."""
    import ctypes, glob
    found = {}
    try:
        ctypes.CDLL("libvulkan.so.1")
        found["loader"] = "libvulkan.so.1"
    except OSError as exc:
        found["loader"] = None
        found["loader_error"] = str(exc)
    icds = []
    for d in ("/usr/share/vulkan/icd.d", "/etc/vulkan/icd.d", "/usr/local/share/vulkan/icd.d"):
        icds += sorted(glob.glob(os.path.join(d, "*.json")))
    found["icd_files"] = icds
    lvp = [f for f in icds if "lvp" in os.path.basename(f)]
    spv = os.environ.get("RT_REFERENCE_SPV", "")
    if found["loader"] and lvp and spv and os.path.exists(spv):
        return {"status": "not run", "reason": "loader, lavapipe ICD and a SPIR-V file are present, but this build carries no Vulkan runner", **found}
    missing = [what for what, ok in (("Vulkan loader (libvulkan.so.1)", found["loader"]), ("lavapipe ICD (lvp_icd*.json)", lvp),
                                     ("the reference's SPIR-V (RT_REFERENCE_SPV; not shipped with this repository)", spv and os.path.exists(spv))) if not ok]
    return {"status": "unavailable", "reason": "missing on this machine: " + "; ".join(missing), **found}


def count_timed_work(config, traversal, frames):
    """Box tests, triangle tests, GetHit calls and record fetches of the timed kernel's own traversal, counted by
    the counter build of the same kernels (raytracer_amd/librt_amd_count.so) in a child process, untimed."""
    import subprocess
    lib = os.path.join(ROOT, "raytracer_amd", "librt_amd_count.so")
    if not os.path.exists(lib):
        return None
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "count_work.py"), str(config), str(traversal), str(frames)],
                             capture_output=True, text=True, timeout=240)
        if out.returncode != 0:
            sys.stderr.write("bench.py: count_work.py failed: %s\n" % out.stderr[-400:])
            return None
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as exc:
        sys.stderr.write("bench.py: count_work.py: %s\n" % exc)
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="1", choices=["1", "2", "3", "4s"], help="benchmark configuration (see the module text)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--frames", type=int, default=0, help="frames per step (per GPU when weak); 0 = the configuration's")
    ap.add_argument("--width", type=int, default=0, help="override the image size (both or none)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--traversal", type=int, default=1, choices=[0, 1], help="headline flavour: 1 = near-first on rebuilt trees, 0 = reference order")
    ap.add_argument("--compare", type=int, default=1, help="1 = at N = 1 also time the other traversal flavour in the same run")
    ap.add_argument("--fast", type=int, default=0, help="1 = FMA-contracted arithmetic (not bit-exact with the oracle)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 disables it")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--comm", default="auto", choices=["auto", "native", "torch"],
                    help="who merges the images for N > 1: native = rt_merge_accum / rt_gather_tiles inside librt_amd.so (RCCL), "
                         "torch = torch.distributed; auto = native on the nccl backend")
    ap.add_argument("--share-gpu", type=int, default=0, help="1 = every rank uses cuda:0 (rehearsal of the N > 1 flow on a one-GPU box, with --backend gloo)")
    ap.add_argument("--pipeline", type=int, default=-1, help="RT_OPT_PIPELINE to measure (-1 = the library's default)")
    ap.add_argument("--trace", type=int, default=0, help="1 = stage markers on stderr")
    ap.add_argument("--stage-timeout", type=float, default=600.0, help="seconds a stage (set-up, warm-up, timed steps) may take before the watchdog ends the process")
    ap.add_argument("--count-work", type=int, default=1, help="1 = at N = 1 count the timed kernel's own box / triangle tests with the counter build, in a child process "
                                                              "after the timed region (0 for profiling runs, which should hold one process)")
    ap.add_argument("--single-dispatch", type=int, default=0,
                    help="1 = also time the frames as ONE dispatch with raysPerPixel = frames (extra launches of the same kernel: off by "
                         "default so that a profile of the default run holds only the timed launches)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import raytracer_amd as R
    from raytracer_amd import distributed as D
    from raytracer_amd.workloads import build_workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    distributed = world > 1
    dog = Watchdog(rank)
    dog.arm(args.stage_timeout, "set-up")

    def mark(what):
        if args.trace:
            print("[rank %d] %.3f %s" % (rank, time.time(), what), file=sys.stderr, flush=True)

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    mark("process group ready")

    wl = build_workload(args.config)
    W, H = (args.width, args.height) if args.width > 0 and args.height > 0 else (wl["width"], wl["height"])
    scene, pc = wl["scene"], wl["pc"]
    if (W, H) != (wl["width"], wl["height"]):
        raise SystemExit("--width/--height override is only meant for quick checks; rebuild the workload's camera for another size")
    sb = scene.buffers()
    arrays = sb.to_arrays()
    r = R.Renderer(local_rank)
    r.upload_scene(sb)
    r.set_option(R.RT_OPT_FAST_MATH, args.fast)
    if args.pipeline >= 0:
        r.set_option(R.RT_OPT_PIPELINE, args.pipeline)
    r.set_option(R.RT_OPT_TRAVERSAL, args.traversal)
    tile_split = args.config == "4s" and distributed
    total_frames = args.frames if args.frames > 0 else wl["frames"]
    if tile_split:
        frames_rank = total_frames                       # every rank renders every frame of its tiles
        scaling = "strong"
    elif args.scaling == "strong" and distributed:
        if total_frames % world:
            raise SystemExit("--scaling strong: %d frames do not divide among %d ranks" % (total_frames, world))
        frames_rank = total_frames // world
        scaling = "strong"
    else:
        frames_rank = total_frames
        scaling = "weak"
    target = D.GpuTarget(r, pc, W, H, dev)      # images are torch tensors, kernels run on a torch stream the target owns
    native = distributed and (args.comm == "native" or (args.comm == "auto" and args.backend == "nccl"))
    if native:
        # the library's own communicator; should it fail to come up on any rank, every rank falls back to
        # torch.distributed's collectives on the same backend (RCCL either way), and the bench line says which
        try:
            target.init_comm()
            up = 1.0
        except R.RtError as exc:
            sys.stderr.write("[rank %d] rt_comm_init failed (%s): using torch.distributed's collectives\n" % (rank, exc))
            up = 0.0
        flag = torch.tensor([up], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) < 1.0:
            if target.native_comm:
                r.comm_destroy()
                target.native_comm = False
            native = False
    mark("scene uploaded, communicator %s" % ("native" if native else "torch" if distributed else "none"))

    # untimed counting pass: GetHit calls and the shader's node/triangle test counters of this rank's
    # share (its frames; in a tile split its tiles). Exact for up to 64 frames; longer frame sets are
    # counted on an evenly spaced sample of 64 frames and scaled (frames differ by < 0.1 %).
    if tile_split:
        target.set_tiles(rank, world, 64)
        my_frames = list(range(total_frames))
    else:
        my_frames = D.frames_of_rank(rank, world, frames_rank) if distributed else list(range(frames_rank))
    counted = my_frames if len(my_frames) <= 64 else [my_frames[(k * len(my_frames)) // 64] for k in range(64)]
    rays = node = tri = 0
    for f in counted:
        p = pc.copy()
        p.frame = f
        st = r.frame_stats(p)
        rays += st.rays; node += st.bvhNodeTests; tri += st.triangleTests
    scale = len(my_frames) / float(len(counted))
    rays, node, tri = rays * scale, node * scale, tri * scale
    if tile_split:
        target.set_tiles(0, 1, 64)
    counts = torch.tensor([rays, node, tri], dtype=torch.float64, device=dev)
    if distributed:
        if args.backend == "nccl":
            dist.all_reduce(counts)
        else:
            host = counts.cpu(); dist.all_reduce(host); counts = host
    total_rays = float(counts[0].item())
    mark("counting pass done")

    def step():
        if tile_split:
            D.render_tile_split(target, frames_rank, 64)
        else:
            D.render_sample_split(target, frames_rank)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(traversal):
        """-> (seconds for args.steps steps, max over ranks; kernel ms per step of this rank)"""
        r.set_option(R.RT_OPT_TRAVERSAL, traversal)
        dog.arm(args.stage_timeout, "warm-up (traversal %d)" % traversal)
        for _ in range(max(args.warmup, 1 if traversal != args.traversal else 0)):
            step()
        fence()
        mark("warm-up done")
        dog.arm(args.stage_timeout, "timed steps (traversal %d)" % traversal)
        kernel_ms = []
        t0 = time.perf_counter()
        for k in range(args.steps):
            step()
            # HIP events recorded by the library around the dispatches, on the stream they run on;
            # reading them back waits only for work that is already enqueued
            kernel_ms.append(r.last_render_ms()[0])
            mark("step %d enqueued and its kernels timed" % k)
        fence()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if distributed:
            if args.backend == "nccl":
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            else:
                host = tmax.cpu(); dist.all_reduce(host, op=dist.ReduceOp.MAX); tmax = host
        mark("timed steps done")
        return float(tmax.item()), float(np.mean(kernel_ms))

    dt, kernel_ms = timed(args.traversal)
    other = None
    if world == 1 and args.compare:
        other = timed(1 - args.traversal)
        r.set_option(R.RT_OPT_TRAVERSAL, args.traversal)

    exchange = None
    if distributed:
        # the exchange step alone, after the timed region (it is inside `value` as part of every step): best of three
        dog.arm(args.stage_timeout, "exchange timing")
        image_bytes = W * H * 16
        best = float("inf")
        if tile_split:
            target.set_tiles(rank, world, 64)
        for _ in range(3):
            fence()
            t0 = time.perf_counter()
            with target.stream_context():
                if tile_split:
                    D.exchange_tile_split(target, 64)
                else:
                    D.exchange_sample_split(target)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        if tile_split:
            target.set_tiles(0, 1, 64)
        tex = torch.tensor([best], dtype=torch.float64)
        if args.backend == "nccl":
            tex = tex.to(dev); dist.all_reduce(tex, op=dist.ReduceOp.MAX)
        else:
            dist.all_reduce(tex, op=dist.ReduceOp.MAX)
        ex_s = float(tex.item())
        # bytes a GPU sends in a ring schedule: all-reduce 2(N-1)/N of the image, all-gather (N-1)/N of its share of two images x N
        sent = (2.0 * (world - 1) / world * image_bytes) if not tile_split else ((world - 1) / world * 2.0 * image_bytes)
        exchange = {"kind": "all-gather of owned tiles (accumulation + draw image)" if tile_split else "all-reduce (sum) of the accumulation image",
                    "ms": ex_s * 1e3, "bytes_sent_per_gpu": sent, "gbps_per_gpu": sent / ex_s / 1e9,
                    "note": "ring schedule assumed for the byte count; one xGMI link carries about 153 GB/s"}
        mark("exchange timed")
    dog.arm(args.stage_timeout, "reporting")

    if rank == 0:
        names = {0: "reference-order", 1: "near-first"}
        ms_step = dt / args.steps * 1e3
        pipeline = r.get_option(R.RT_OPT_PIPELINE)
        group = r.get_option(R.RT_OPT_FRAMES_PER_LAUNCH) if pipeline in (0, 3) else 1
        launches = (frames_rank + group - 1) // group
        kernel = {0: "rt_megakernel", 3: "rt_pairkernel" if frames_rank >= 2 else "rt_megakernel"}.get(pipeline, "pipeline %d" % pipeline)
        kernel += "_fast" if args.fast else "_strict"
        launch_ms = kernel_ms / launches                                  # average duration of one launch of the dominant kernel
        pixels_rank = W * H if not tile_split else W * H / world
        bytes_per_launch = (40.0 * node + 48.0 * tri + 48.0 * pixels_rank * frames_rank) / launches     # rank 0's share
        flops_per_launch = (21.0 * node + 45.0 * tri) / launches
        alg_gbps = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        alg_tflops = flops_per_launch / (launch_ms * 1e-3) / 1e12
        traffic, pmc = measured_profile(kernel, args.config, args.traversal, group)
        scene_bytes = sum(a.nbytes for a in arrays)
        # HBM is the bound only when the measured traffic says so (half the peak or more); without a measurement,
        # when the scene cannot sit in the caches (aggregate L2 8 x 4 MiB, 256 MiB memory-side cache behind it).
        # Measured: every configuration moves 0.3-0.8 TB/s - every record fetch is a dependent access served by
        # a cache, so all of them are bound by issue and latency.
        traffic_rate = None if traffic is None else traffic / (launch_ms * 1e-3) / 1e9
        cache_resident = (traffic_rate < 0.5 * HBM_PEAK_GBPS) if traffic_rate is not None else (scene_bytes < 256 * 2 ** 20)
        spp_total = frames_rank * (world if not tile_split else 1)
        # the work the TIMED kernel does (its own trees and order), counted by the counter build of the same kernels
        # over the frames of one step (up to 64, scaled beyond); reference order: the counting kernel's figures are it
        done = None
        if world == 1 and pipeline == 3 and args.count_work:
            cf = min(frames_rank, 64)
            cw = count_timed_work(args.config, args.traversal, cf)
            if cw is not None:
                k = frames_rank / float(cf) / launches
                done = {"box_tests_per_launch": cw["box_tests"] * k, "triangle_tests_per_launch": cw["triangle_tests"] * k,
                        "record_fetches_per_launch": cw["record_fetches"] * k, "rays_per_launch": cw["rays"] * k,
                        "flops_per_launch": (21.0 * cw["box_tests"] + 45.0 * cw["triangle_tests"]) * k,
                        "bytes_per_launch": (40.0 * cw["box_tests"] + 48.0 * cw["triangle_tests"]) * k + 48.0 * pixels_rank * frames_rank / launches,
                        "counted_by": "raytracer_amd/librt_amd_count.so (the timed kernels with work counters), %d frames, untimed" % cf}
                if cw.get("lane_share"):
                    # share of a wave's 64 lanes that work where the wave passes (walks of four-child records only)
                    done["lane_share_in_walks"] = cw["lane_share"]
                done["tflops"] = done["flops_per_launch"] / (launch_ms * 1e-3) / 1e12
                done["gbps"] = done["bytes_per_launch"] / (launch_ms * 1e-3) / 1e9
        ref_equiv = {"flops_per_launch": flops_per_launch, "bytes_per_launch": bytes_per_launch, "tflops": alg_tflops, "gbps": alg_gbps,
                     "frac_of_valu_peak": alg_tflops / VALU_PEAK_TFLOPS,
                     "counted_by": "counting build of kernel 0: the reference's trees in the reference's order, i.e. the work the reference "
                                   "shader would do for these rays (21 flops per box test, 45 per triangle test; 40 / 48 bytes)"}
        ach_tflops = done["tflops"] if done else alg_tflops
        ach_gbps = done["gbps"] if done else alg_gbps
        out = {
            "metric": "Mrays/s", "value": total_rays * args.steps / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": wl["data"],
            "config": {"workload": wl["workload"] + (" (this run: %d frames per GPU)" % frames_rank if frames_rank != wl["frames"] else ""),
                       "width": W, "height": H, "frames_per_gpu": frames_rank, "total_spp": spp_total,
                       "parallelism": ("tile-split x%d, 64x64 tiles along a serpentine path, one all-gather of the owned tiles" % world if tile_split else
                                       "sample-split x%d (one all-reduce of the accumulation image)" % world) +
                                      (" inside librt_amd.so over RCCL" if native else " through torch.distributed/%s" % args.backend) if distributed else "single GPU",
                       "flavour": "%s traversal, %s arithmetic" % (names[args.traversal], "FMA-contracted" if args.fast else "strict binary32"),
                       "library_default": "RT_OPT_TRAVERSAL 0 (reference order, bit-exact with the oracle): the other_flavour figures; near-first is opt-in and held to the tolerance bars",
                       "pipeline": pipeline},
            "ms_per_frame": ms_step / frames_rank, "msamples_per_s": W * H * spp_total * args.steps / dt / 1e6,
            "rays_per_step": total_rays,
            "rays_counted": "exact (counting kernel over every frame of the step)" if len(counted) == len(my_frames) else
                            "counting kernel over %d evenly spaced frames of %d, scaled" % (len(counted), len(my_frames)),
            # Served by the caches: the bound is vector-ALU issue and the latency of dependent record fetches, not
            # HBM. `achieved` / `frac`: algorithmic flops per second of the work the TIMED kernel does (21 per box
            # test, 45 per triangle test, its own trees and order) against the FP32 vector peak; the work the
            # reference's order would need for the same rays is given beside it, and so are the algorithmic byte
            # rate (a work rate served by caches: no fraction of the HBM peak is formed from it) and the measured
            # HBM traffic.
            "roofline": {"bound": "valu+latency" if cache_resident else "hbm",
                         "achieved": ach_tflops if cache_resident else ach_gbps,
                         "peak": VALU_PEAK_TFLOPS if cache_resident else HBM_PEAK_GBPS,
                         "unit": "TFLOP/s" if cache_resident else "GB/s",
                         "frac": ach_tflops / VALU_PEAK_TFLOPS if cache_resident else ach_gbps / HBM_PEAK_GBPS,
                         "counters": "timed kernel's own work" if done else "reference-order work (no count of the timed kernel's own work in this run)",
                         "traffic": traffic,
                         "traffic_gbps": traffic_rate,
                         "traffic_frac_of_hbm_peak": None if traffic_rate is None else traffic_rate / HBM_PEAK_GBPS,
                         "kernel": kernel, "launch_ms": launch_ms, "frames_per_launch": group, "launches_per_step": launches,
                         "work_done": done, "reference_work_equivalent": ref_equiv,
                         "valu_busy": None if not pmc else pmc.get("valu_busy"), "lane_util": None if not pmc else pmc.get("lane_util"),
                         "pmc": pmc,
                         "scene_bytes": scene_bytes,
                         "hbm": {"algorithmic_gbps": ach_gbps, "peak_gbps": HBM_PEAK_GBPS,
                                 "note": "algorithmic bytes (40 per box test, 48 per triangle test, 48 per pixel and frame) per second; "
                                         "a work rate that the caches serve, not bytes that cross the HBM interface - see traffic"}},
            "lavapipe_baseline": lavapipe_probe() if world == 1 else None,
        }
        if exchange is not None:
            out["exchange"] = exchange
        if other is not None:
            odt, okms = other
            out["other_flavour"] = {"flavour": "%s traversal, %s arithmetic" % (names[1 - args.traversal], "FMA-contracted" if args.fast else "strict binary32"),
                                    "value": total_rays * args.steps / odt / 1e6, "unit": "Mrays/s", "ms_per_step": odt / args.steps * 1e3,
                                    "ms_per_frame": odt / args.steps * 1e3 / frames_rank, "launch_ms": okms / launches}
        if world == 1:
            # the reference's interactive use: ONE dispatch per frame (rt_render_frame, 1 ray per pixel), after the
            # timed region; 16 dispatches of frames 0..15 into a cleared accumulation image
            target.clear()
            pF = pc.copy()
            for f in range(2):
                pF.frame = f
                r.render_frame(pF)
            r.sync()
            t0 = time.perf_counter()
            for f in range(2, 18):
                pF.frame = f
                r.render_frame(pF)
            r.sync()
            out["ms_per_single_frame_dispatch"] = (time.perf_counter() - t0) / 16 * 1e3
        if world == 1 and args.single_dispatch:
            # the other reading of "64 spp": ONE dispatch with raysPerPixel = frames (samples seeded by the
            # sample index instead of the frame index, raytrace.comp:779-783), after the timed region
            pN = pc.copy()
            pN.raysPerPixel = frames_rank
            pN.frame = 0
            target.clear()
            r.render_frames(pN, 0, 1, 1)
            r.sync()
            r.render_frames(pN, 0, 1, 1)
            r.sync()
            out["ms_single_dispatch_%dspp" % frames_rank] = r.last_render_ms()[0]
        if world == 1 and args.cpu_seconds > 0:
            host_cpus = os.cpu_count() or 1
            share = int(os.environ.get("OMP_NUM_THREADS", "16") or 16)      # the box sets it to a one-GPU job's CPU share
            threads = max(1, min(host_cpus, share))
            p0 = pc.copy()
            p0.frame = 0
            dog.arm(args.stage_timeout + args.cpu_seconds * 4, "CPU baseline")
            out["cpu_baseline"] = cpu_baseline(arrays, p0.as_bytes(), W, H, args.cpu_seconds, threads, host_cpus)
        print(json.dumps(out), flush=True)
    dog.arm(args.stage_timeout, "shutdown")
    target.close()
    r.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    dog.disarm()


if __name__ == "__main__":
    main()
