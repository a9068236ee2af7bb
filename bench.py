#!/usr/bin/env python3
"""bench.py -- Mrays/s of the brickmap ray-tracing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one launch over a batch of 16 views (the benchmark cameras A-D in turn, every frame with its own
FrameNumber), each a 1920x1080 frame of BASELINE.json configs[2]: primary ray + shadow ray for every primary hit +
1 bounce sample (reference gate `lDot == 0`), 8192x512x8192 procedurally generated brickmap (factor 32).
`--views-per-step 1` gives the reference's one view per launch (also reported in the line, `one_view_per_launch`).
The batch is what keeps a GPU busy when it renders only 1/N of every frame: a launch's low-occupancy tail is paid
once per launch, not once per frame.  With N > 1 every view is sharded by interleaved 16-row
strips, the brickmap is replicated per GPU and the packed strips of the step are gathered to rank 0 over RCCL in one
collective (strong scaling: the frames and the batch are the same at every N).
`value` = rays actually traced by all ranks / wall time of the K timed steps (inputs resident in HBM, gather
included).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, Chip-level parameters)

WORKLOADS = {
    # name: (world X, Y, Z, factor, generator, width, height, shadow, bounce_samples)
    "c3_8k_1080p_shadow_bounce": (8192, 512, 8192, 32, 1, 1920, 1080, 1, 1),
    "c2_1k_1080p_primary": (1024, 256, 1024, 32, 1, 1920, 1080, 0, 0),
    "c4_8k_4k_shadow_bounce": (8192, 512, 8192, 32, 1, 3840, 2160, 1, 1),
    "dev_small": (1024, 256, 1024, 32, 2, 1920, 1080, 1, 1),
    # configs[4]'s world and frame with the reference's ray set (its second bounce is an extension, not built)
    "c5_16k_4k_shadow_bounce": (16384, 1024, 16384, 32, 1, 3840, 2160, 1, 1),
    # configs[2] with the 2-D tester's brick edge ("f=8 reported as a variant")
    "c3_f8_variant": (8192, 512, 8192, 8, 1, 1920, 1080, 1, 1),
    # ablations of configs[2] for kernel work (not bench lines)
    "c3_primary_only": (8192, 512, 8192, 32, 1, 1920, 1080, 0, 0),
    "c3_primary_shadow": (8192, 512, 8192, 32, 1, 1920, 1080, 1, 0),
    # the reference's SHIPPED configuration: 1024^3 world (VoxelApp/main.cu:24), 1280x720 (:15-16), DEBUG_VIEW + checkerboard
    # (Renderer.cu:4-5): half the pixels per frame, primary rays only
    "ref_shipped_720p_checkerboard_debug": (1024, 1024, 1024, 32, 1, 1280, 720, 0, 0),
}
# render mode / checkerboard per workload (default: shaded, off)
WORKLOAD_FLAGS = {"ref_shipped_720p_checkerboard_debug": dict(mode=1, checkerboard=True)}

# Fixed cameras (position as a fraction of the world extent, euler angles for GetDirections); see DESIGN.md.
CAMERAS = [
    ("A", (0.50, 0.90, 0.50), (-0.45, 0.70, 0.0)),
    ("B", (0.10, 1.20, 0.10), (-0.60, 3.90, 0.0)),
    ("C", (0.50, 1.50, 0.50), (-1.5707, 0.0, 0.0)),
    ("D", (0.02, 0.55, 0.50), (-0.05, 1.5707, 0.0)),
]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="c3_8k_1080p_shadow_bounce", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--cpu-frames", type=int, default=48, help="frames of the workload timed on the host cores")
    ap.add_argument("--views-per-step", type=int, default=16,
                    help="views rendered by one launch (vxrt_render_views); 1 = one RenderScreen-style launch per frame")
    ap.add_argument("--kernel-variant", type=int, default=4, choices=[1, 4, 7],
                    help="render kernel (vxrt_set_kernel_variant): 4 = default (= 7, the persistent kernel on the wave-level "
                         "tracer); 1 = the straightforward per-lane loops (cross-check)")
    ap.add_argument("--bounce-all-hits", type=int, default=0)
    ap.add_argument("--bounce-depth", type=int, default=1, help="2 = second bounce (BASELINE config 5; extension beyond the reference)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: all ranks share cuda:0 and the gather goes through gloo/host memory, to "
                         "exercise the N>1 code path on a one-GPU box (numbers are meaningless)")
    ap.add_argument("--force-gather", action="store_true",
                    help="development only: with ONE rank, run the N>1 code path anyway (packed strips, two-deep "
                         "pipeline, gather on a one-rank NCCL communicator, de-interleave) and check the frames")
    return ap.parse_args()


def main():
    args = parse()
    from voxelengine_amd import launcher  # imports neither torch nor HIP

    if args.gpus > 1 and not launcher.under_launcher():
        # started as a plain program: become the launcher.  N fresh children of this script, one per GPU, are started
        # BEFORE anything here touches the GPU; rank 0's one JSON line is relayed, a failing rank fails the job.
        # The library is (re)built HERE, once, before any rank exists: the cross-compile needs no GPU, and N ranks that all
        # find a stale tree would otherwise queue up behind build_lib's lock.
        from voxelengine_amd import build as _build
        _build.build_lib()
        rc, out = launcher.launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        lines = [ln for ln in out.splitlines() if ln.strip()]
        if lines:
            print(lines[-1], flush=True)
        sys.exit(rc if rc else (0 if lines else 1))
    # stdout carries exactly ONE line, the result JSON.  Native libraries write there too (RCCL prints a version banner
    # to fd 1 when its communicator is created), so fd 1 points at stderr until the line is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        result = run(args)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if result is not None:
        print(json.dumps(result), flush=True)


def run(args):
    import torch
    import torch.distributed as dist

    import voxelengine_amd as vx
    from voxelengine_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start bench.py directly (it launches its own ranks) or under "
                         "torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    rehearse = args.rehearse_on_one_gpu and world > 1
    if rehearse:
        local_rank = 0  # every rank on the one GPU; collectives through gloo and host memory
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if rehearse else dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    elif args.force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sharded = world > 1 or args.force_gather

    X, Y, Z, F, gen, W, H, shadow, bounce = WORKLOADS[args.workload]
    wflags = WORKLOAD_FLAGS.get(args.workload, {})
    ctx = vx.Context(local_rank)
    if args.kernel_variant != 4:
        ctx.set_kernel_variant(args.kernel_variant)
    t_build0 = time.time()
    info = ctx.build_world(gen, X, Y, Z, F)  # every rank builds its own replica in its HBM
    ctx.synchronize()
    t_build = time.time() - t_build0

    light = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))  # VoxelApp/main.cu:59-63
    ctx.SetEnvironment((light, light, light), (2, 2, 2), (0.5, 0.5, 0.5))
    ctx.SetFOV(90.0)
    cams = []
    for name, frac, euler in CAMERAS:
        f, u, r = vx.GetDirections(euler)
        cams.append((name, (frac[0] * X, frac[1] * Y, frac[2] * Z), f, u, r))

    # One step = one launch over V views (the benchmark cameras in turn; frame g of the run uses camera g mod 4 and
    # FrameNumber g + 1).  V = 1 is the reference's one-view-per-RenderScreen behaviour.
    V = max(1, min(16, args.views_per_step))
    plan = sharding.ShardPlan(W, H, sharding.STRIP_ROWS, world, rank)
    step_bytes = V * plan.shard_bytes  # what a rank contributes per step when sharded: V packed shards back to back
    local = torch.zeros(step_bytes, dtype=torch.uint8, device=dev)
    frames = torch.zeros((V, H, W, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
    shards = torch.zeros((world, step_bytes), dtype=torch.uint8, device=dev) if (rank == 0 and sharded) else None

    def opts(stats=False):
        return vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, bounce_all_hits=bool(args.bounce_all_hits),
                                bounce_depth=args.bounce_depth, strip_rows=plan.strip_rows, strip_count=world,
                                strip_index=rank, compact=sharded, collect_stats=stats, **wflags)

    def views_of(i, target, hits=None):
        """the V views of step i, rendered into `target` ((V,H,W,4) frames, or V packed shards back to back)"""
        out = []
        for j in range(V):
            g = i * V + j
            name, pos, f, u, r = cams[g % len(cams)]
            fb = target[j] if target.dim() == 4 else target[j * plan.shard_bytes:(j + 1) * plan.shard_bytes]
            v = dict(fb=fb, origin=pos, fwd=f, up=u, right=r, frame_number=g + 1)
            if hits is not None:
                v["hit_aov"] = hits[j]
            out.append(v)
        return out

    def deinterleave(sh, fr):
        # one launch for the V views: the shards of view j sit at byte offset j * shard_bytes of every rank's contribution
        ctx.deinterleave_views(W, H, plan.strip_rows, world, sh, step_bytes, plan.shard_bytes, V, fr, W * H * 4)

    # N > 1: two-deep pipeline, the RCCL gather of step k overlaps the render of step k+1
    # ... and consecutive steps alternate between two streams: a strip shard is a small launch, its tail a fifth of it; step
    # k+1's first waves fill the SIMD slots step k's last ones leave (tools/views_probe.py: 1/8 shards 6.7 -> 7.9 Grays/s per GPU)
    pipe = None
    frames2 = None
    if sharded and not rehearse:
        frames2 = [frames, torch.zeros_like(frames)] if rank == 0 else [None, None]
        pipe = sharding.GatherPipeline(plan, lambda n: torch.zeros(n, dtype=torch.uint8, device=dev), frames2, deinterleave,
                                       nbytes=step_bytes, streams=[torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)], depth=3)

    def step(i, ev=None, one_view_launches=False):
        if pipe:
            with pipe.stream(i):  # (everything below is issued on the step's stream: torch's current stream inside)
                return step_on_current_stream(i, ev, one_view_launches)
        return step_on_current_stream(i, ev, one_view_launches)

    def step_on_current_stream(i, ev, one_view_launches):
        target = frames if not sharded else (pipe.local(i) if pipe else local)
        if ev is not None:
            ev[0].record()
        if one_view_launches:  # the reference's call pattern: one RenderScreen-style launch per view
            o = opts()
            for v in views_of(i, target):
                o.frame_number = v["frame_number"]
                ctx.RenderScreen(W, H, v["fb"], v["origin"], v["fwd"], v["up"], v["right"], o)
        else:
            ctx.RenderViews(W, H, views_of(i, target), opts())
        if ev is not None:
            ev[1].record()
        if pipe:
            pipe.submit(i)
        elif world > 1:
            host = torch.zeros((world, step_bytes), dtype=torch.uint8) if rank == 0 else None
            sharding.gather_frame(plan, local.cpu(), host, frames,
                                  lambda sh, fr: (shards.copy_(sh), deinterleave(shards, fr)))

    def fence():
        if pipe:
            pipe.flush()
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    ctx.frame_stats()  # drop the warm-up's ray counters
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, events[k])
    fence()
    dt = time.perf_counter() - t0
    st = ctx.frame_stats()  # rays traced by this rank in the K timed steps (always counted by the kernel)
    rays_local = st.total_rays()
    kernel_ms = [a.elapsed_time(b) for a, b in events]

    # algorithmic bytes of the same K launches, from the probe-counting kernel variant, untimed
    for k in range(args.steps):
        ctx.RenderViews(W, H, views_of(args.warmup + k, frames if not sharded else local), opts(stats=True))  # `local`: scratch
    sp = ctx.frame_stats()
    assert sp.total_rays() == rays_local, "ray counts differ between the timed and the counting pass"
    if os.environ.get("BENCH_DIAG") and rank == 0:  # wave-loop diagnostics of the counting launches (development)
        g = [int(v) for v in sp.dbg]
        it = max(g[0], 1)
        print("diag: iterations/launch/wave %.0f  walking lanes/iteration %.1f  probes/iteration %.1f | per 100 iterations "
              "(lanes per run): next/pass %.1f (%.1f)  end %.1f (%.1f)  box %.1f (%.1f) | time share next/pass+retire %.1f%%  "
              "box+end (first vote of a round) %.1f%%" % (
                  g[0] / args.steps / 4096.0, g[1] / it, (sp.coarse_probes + sp.fine_probes) / it, 100.0 * g[4] / it, g[7] / max(g[4], 1),
                  100.0 * g[2] / it, g[5] / max(g[2], 1), 100.0 * g[3] / it, g[6] / max(g[3], 1), 100.0 * g[10] / max(g[8], 1),
                  100.0 * g[11] / max(g[8], 1)), file=sys.stderr, flush=True)
    bytes_local = sp.algorithmic_bytes()
    probes_local = (int(sp.coarse_probes), int(sp.brick_entries), int(sp.fine_probes), int(sp.primary_rays))

    tot = torch.tensor([float(rays_local), float(bytes_local), float(sum(kernel_ms))], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    per_rank = [tot.clone()]
    if world > 1:
        per_rank = [torch.zeros_like(tot) for _ in range(world)]
        dist.all_gather(per_rank, tot)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    per_rank = [[float(v) for v in t.tolist()] for t in per_rank]

    # the same frames with one view per launch (the reference's RenderScreen call pattern), every rank, same fences
    dt1 = None
    if V > 1 and not args.force_gather:
        ctx.frame_stats()
        fence()
        t1 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + k, one_view_launches=True)
        fence()
        dt1 = time.perf_counter() - t1
        s1 = ctx.frame_stats()
        assert s1.total_rays() == rays_local, "ray counts differ between multi-view and single-view launches"
        t1max = torch.tensor([dt1], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t1max, op=dist.ReduceOp.MAX)
        dt1 = float(t1max.item())
    rays_total, bytes_total, kernel_ms_total = [float(v) for v in tot.tolist()]
    dt = float(tmax.item())

    # ... and with the interactive caller's remedy: one launch per frame, two frames in flight on two streams, the host
    # never more than two frames ahead (what Graphics::RenderScreenAsync / WaitFrame of the C++ facade does)
    dt2 = None
    if V > 1 and world == 1 and not args.force_gather:
        # two non-blocking streams from torch itself (its pool streams are created hipStreamNonBlocking): the handles go to
        # libvxrt through the C ABI, and torch and the library share the one HIP runtime torch loaded
        ext = [torch.cuda.Stream(device=dev) for _ in range(2)]
        raw = [st_.cuda_stream for st_ in ext]
        if raw:
            done = [torch.cuda.Event() for _ in range(3)]
            ring = torch.zeros((3, H, W, 4), dtype=torch.uint8, device=dev)
            o = opts()
            torch.cuda.synchronize()
            ctx.frame_stats()
            t2 = time.perf_counter()
            n2 = args.steps * V
            for g in range(n2 + 2):
                if g < n2:
                    v = views_of(args.warmup + g // V, frames)[g % V]  # the timed steps' views, frame numbers included
                    o.frame_number = v["frame_number"]
                    ctx.RenderScreen(W, H, ring[g % 3], v["origin"], v["fwd"], v["up"], v["right"], o, stream=raw[g % 2])
                    done[g % 3].record(ext[g % 2])
                if g >= 2:
                    done[(g - 2) % 3].synchronize()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            s2 = ctx.frame_stats()
            assert s2.total_rays() == rays_local, "ray counts differ between multi-view and two-in-flight launches"

    # ... and the bench step itself (V views per launch) with two steps in flight on two streams: what the N > 1 path does on
    # every rank, measured here on whole frames (an extra field; `value` stays one step at a time, so that the HIP events
    # around a launch time that launch alone)
    dt3 = None
    if V > 1 and world == 1 and not args.force_gather:
        ext3 = [torch.cuda.Stream(device=dev) for _ in range(2)]
        tgt3 = [frames, torch.zeros_like(frames)]
        for k in range(2):  # warm both streams
            ctx.RenderViews(W, H, views_of(args.warmup + k, tgt3[k & 1]), opts(), stream=ext3[k & 1].cuda_stream)
        torch.cuda.synchronize()
        ctx.frame_stats()
        t3 = time.perf_counter()
        for k in range(args.steps):
            ctx.RenderViews(W, H, views_of(args.warmup + k, tgt3[k & 1]), opts(), stream=ext3[k & 1].cuda_stream)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t3
        s3 = ctx.frame_stats()
        assert s3.total_rays() == rays_local, "ray counts differ between one step at a time and two steps in flight"

    # SURVEY 8(d)'s side measurements, rank 0 of a one-GPU run only (untimed region): a measured stream-copy ceiling beside the
    # spec peak -- a device-to-device copy of 1 GiB, bytes read + bytes written per second -- and the per-camera split of the
    # workload from single-view launches (cameras A-D differ several-fold in rays per pixel chain)
    copy_gbs = None
    per_camera = None
    if rank == 0 and world == 1 and not args.force_gather:
        nbytes = 1 << 30
        src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        src.zero_()
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * nbytes * 5 / (e0.elapsed_time(e1) / 1e3) / 1e9
        del src, dst
        per_camera = []
        o = opts()
        reps = 4
        for ci, (name, pos, f, u, r) in enumerate(cams):
            o.frame_number = ci + 1
            ctx.RenderScreen(W, H, frames[0], pos, f, u, r, o)  # warm
            torch.cuda.synchronize()
            ctx.frame_stats()
            e0.record()
            for _ in range(reps):
                ctx.RenderScreen(W, H, frames[0], pos, f, u, r, o)
            e1.record()
            torch.cuda.synchronize()
            sc = ctx.frame_stats()
            ms = e0.elapsed_time(e1) / reps
            per_camera.append({"camera": name, "rays": int(sc.total_rays() // reps), "primary_hits": int(sc.primary_hits // reps),
                               "ms": round(ms, 4), "mrays_s": round(sc.total_rays() / reps / ms / 1e3, 1)})

    result = None
    if rank == 0:
        mrays = rays_total / dt / 1e6
        # dominant kernel = k_render: algorithmic bytes per launch / average launch duration (HIP events on the
        # launch stream), averaged over all ranks' launches
        n_launch = args.steps * world
        avg_kernel_s = kernel_ms_total / 1e3 / n_launch
        achieved = (bytes_total / n_launch) / avg_kernel_s / 1e9
        launch_overlap = None
        if pipe:
            # N > 1: consecutive steps run on two streams and overlap, so the events around one launch also cover part of its
            # neighbour: the rate a GPU sustains is its bytes over the timed region, not bytes over an (inflated) launch duration
            achieved = (bytes_total / world) / dt / 1e9
            launch_overlap = ("consecutive steps alternate between two streams (the next step's first waves fill the SIMD slots "
                              "the previous step's tail leaves); achieved = one GPU's algorithmic bytes / the timed region; "
                              "avg_launch_ms (HIP events around each launch) includes the overlap")
        # PMC-derived fields come from a profile kept in profiles/traffic.json (tools/prof.sh + tools/make_traffic_entry.py).
        # They are reported only when that profile was taken with THIS library (content hash of its sources) and THIS
        # kernel; anything else is stale: null, and "stale_profile": true says why.
        traffic = None
        issue = None
        stale_profile = None
        kernel_id = ctx.kernel_for_launch(W, H, opts(), V if V > 1 else 0)
        # template arguments: <probe counting, second bounce, multi-view, wide grid (beyond the tracer's packed step counters)>
        wide = vx.grid_is_wide(info.cdims)
        kernel_symbol = "%s<false,%s,%s,%s>" % (ctx.KERNEL_NAMES[kernel_id], "true" if args.bounce_depth == 2 else "false",
                                                "true" if V > 1 else "false", "true" if wide else "false")
        if kernel_id == 1:
            kernel_symbol = "k_render<false>"
        # the counting pass above ran the STATS instantiation of the same kernel (it derives the probe counts from the
        # tracer's own step counters; no other kernel stands in for it)
        count_symbol = kernel_symbol.replace("<false", "<true", 1)
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj) and world == 1:
            try:
                # profiled launch shapes: the default batch (entry `workload`, which names its views_per_launch) and V = 1
                tab = json.load(open(tj))
                ent = tab.get(args.workload + "_single_view_launch", {}) if V == 1 else tab.get(args.workload, {})
                if ent.get("views_per_launch", 1 if V == 1 else None) == V:
                    hash_file = vx.lib_path() + ".srchash"
                    lib_hash = open(hash_file).read().strip() if os.path.exists(hash_file) else None
                    if lib_hash and ent.get("lib_srchash") == lib_hash and ent.get("kernel_symbol") == kernel_symbol.replace(" ", ""):
                        traffic = ent.get("hbm_bytes_per_launch")
                        issue = ent.get("issue_utilisation")
                        stale_profile = False
                    else:
                        stale_profile = True
            except Exception:
                traffic, issue, stale_profile = None, None, None
        result = {
            "metric": "Mrays/s primary+1-bounce @1080p, 8k×512×8k brickmap; % HBM roofline",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": args.workload, "world": [X, Y, Z], "factor": F,
                "generator": ["hash_heightfield", "perlin_ref", "int_terrain"][gen], "resolution": [W, H],
                "rays": "primary + shadow per hit + %d bounce sample(s), gate=%s%s" % (
                    bounce, "all-hits" if args.bounce_all_hits else "reference lDot==0",
                    ", second bounce (extension beyond the reference)" if args.bounce_depth >= 2 else ""),
                "step": "one launch over %d view(s) of %dx%d: cameras %s in turn" % (
                    V, W, H, ",".join(c[0] for c in CAMERAS)),
                "views_per_step": V,
                "cameras": [c[0] for c in CAMERAS], "sharding": "interleaved %d-row strips, gather to rank 0" % plan.strip_rows
                if world > 1 else "none", "rays_per_step": round(rays_total / args.steps, 1),
                "world_build_s": round(t_build, 2), "bricks": int(info.nslots), "world_hbm_gib": round(info.hbm_bytes / 2**30, 3),
            },
            "roofline": {"bound": "hbm (algorithmic-bytes convention)", "kernel": kernel_symbol,  # the library's choice
                         "probe_count_kernel": count_symbol, "fallback": None,  # (no launch shape or world takes another kernel)
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(bytes_total / n_launch, 1),
                         "avg_launch_ms": round(avg_kernel_s * 1e3, 4), "launch_overlap": launch_overlap,
                         "bytes_per_ray": round(bytes_total / max(rays_total, 1.0), 1),
                         # SURVEY 8(d)'s per-ray counters behind the bytes (rank 0's launches): Nc x 28 B + Nb x 24 B + Nf x 4 B + 4 B per pixel
                         "probes_per_ray": {"coarse_Nc": round(probes_local[0] / max(rays_local, 1), 3), "brick_entries_Nb": round(probes_local[1] / max(rays_local, 1), 3),
                                            "fine_Nf": round(probes_local[2] / max(rays_local, 1), 3), "pixels": round(probes_local[3] / max(rays_local, 1), 3)},
                         # what the fraction above is and is not: the contract's yardstick is ALGORITHMIC bytes on the
                         # reference's data layout; the bytes that really cross the HBM interface (PMC, profiles/) are a
                         # fraction of them, and the kernel's physical limiter is instruction issue (vector ALU pipe ~93 % busy)
                         "convention": "achieved = algorithmic bytes (SURVEY.md 8d: 28 B per coarse probe, 24 B per brick entry, "
                                       "4 B per brick probe, 4 B per pixel) / launch time; not measured HBM traffic",
                         "traffic_frac": None if traffic is None else round(traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 5),
                         "physical_bound": "instruction issue (divergent traversal): a SIMD retires a wave64 vector instruction in ~2.3 "
                                           "cycles (add/mul/logic/shift on vector or constant operands) or ~4.15 (compares, selects, "
                                           "min/max, conversions, three-operand forms, scalar operands); priced that way the vector ALU "
                                           "pipe is ~93 % busy, and the launch runs at ~2.4 cycles per issued instruction per SIMD, "
                                           "vector or scalar (profiles/r03_instr_cost.md, profiles/r03_variant7.md)",
                         "issue_utilisation": issue, "stale_profile": stale_profile},
        }
        if rehearse or args.force_gather:  # the gathered frames of the last step must equal single-GPU, single-view renders of the same frames
            full = torch.zeros_like(frames)
            for v in views_of(args.warmup + args.steps - 1, full):
                ctx.RenderScreen(W, H, v["fb"], v["origin"], v["fwd"], v["up"], v["right"],
                                 vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce,
                                                  bounce_all_hits=bool(args.bounce_all_hits), bounce_depth=args.bounce_depth,
                                                  frame_number=v["frame_number"]))
            torch.cuda.synchronize()
            got = pipe.frame_of(args.warmup + args.steps - 1) if pipe else frames  # (two render streams: two frame buffers)
            result["rehearsal"] = {"gathered_frame_equals_single_gpu_frame": bool(torch.equal(full, got)),
                                   "note": "all ranks on one GPU over gloo: value is not a measurement" if rehearse
                                   else "one-rank NCCL communicator: exercises the N>1 code path, not a scaling number"}
            ctx.frame_stats()
        if copy_gbs is not None:
            result["roofline"]["stream_copy_gbs"] = round(copy_gbs, 1)  # measured: 1 GiB device-to-device copy, read + write
            result["roofline"]["frac_of_stream_copy"] = round(achieved / copy_gbs, 5)
            result["per_camera_single_view"] = per_camera
        if dt1 is not None:
            result["one_view_per_launch"] = {
                "value": round(rays_total / dt1 / 1e6, 2), "unit": "Mrays/s",
                "ms_per_frame": round(dt1 / (args.steps * V) * 1e3, 4),
                "roofline_frac": round(bytes_total / dt1 / 1e9 / HBM_PEAK_GBS, 5),
                "note": "same frames, one vxrt_render launch per frame (the reference's RenderScreen call pattern)"
                        + ("; every rank launches its strip shard of each frame, the step's shards are gathered as before" if world > 1 else "")}
        if dt2 is not None:
            result["one_view_two_in_flight"] = {
                "value": round(rays_total / dt2 / 1e6, 2), "unit": "Mrays/s",
                "ms_per_frame": round(dt2 / (args.steps * V) * 1e3, 4),
                "roofline_frac": round(bytes_total / dt2 / 1e9 / HBM_PEAK_GBS, 5),
                "note": "same frames, one vxrt_render launch per frame on two alternating streams, the host at most two frames "
                        "ahead: the call pattern of the facade's RenderScreenAsync/WaitFrame for interactive callers"}
        if dt3 is not None:
            result["two_steps_in_flight"] = {
                "value": round(rays_total / dt3 / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt3 / args.steps * 1e3, 4),
                "roofline_frac": round(bytes_total / dt3 / 1e9 / HBM_PEAK_GBS, 5),
                "note": "the same steps alternating between two streams (the next launch's first waves fill the SIMD slots the "
                        "previous launch's tail leaves): what every rank of an N > 1 run does; not `value`"}
        if world > 1:
            # per-rank roofline of the dominant kernel: algorithmic bytes of the rank's launches / its launch time
            result["roofline"]["per_rank"] = [
                {"rank": r, "achieved": round(b / (ms / 1e3) / 1e9, 2), "frac": round(b / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                 "avg_launch_ms": round(ms / args.steps, 4), "rays": int(n)} for r, (n, b, ms) in enumerate(per_rank)]
        if args.cpu_baseline == "auto" and not args.force_gather:
            # rank 0 only, after the timed regions (the other ranks wait at the closing barrier); N > 1 times a smaller
            # sample so that the whole-node run stays short
            full_opts = lambda: vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce,  # noqa: E731
                                                 bounce_all_hits=bool(args.bounce_all_hits), bounce_depth=args.bounce_depth, **wflags)
            nfr = args.cpu_frames if world == 1 else min(args.cpu_frames, V)
            result["cpu_baseline"], result["parity"] = cpu_baseline(ctx, vx, W, H, shadow, bounce, args, V, frames, views_of,
                                                                    full_opts, nfr, wflags)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    return result if rank == 0 else None


def host_cores() -> int:
    """Cores this process may actually use: the scheduler affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(ctx, vx, W, H, shadow, bounce, args, V, frames, views_of, opts, cpu_frames, wflags=None):
    """The CPU oracle (kind "port": this repo's C restatement of the reference algorithm) timed on the GPU box's
    host cores on a bounded sample of the same workload: the frames of `--cpu-frames / V` steps.  The same frames double
    as a parity gate: the HIP framebuffers of the same steps (same launches as the timed ones) must equal the oracle's
    byte for byte, and so must the primary hit voxel indices."""
    import torch
    from oracle import vxo

    cores = host_cores()
    w = ctx.download_world()
    world = vxo.World.wrap(w["factor"], w["cdims"], w["coarse_bits"], w["brick_slot"], w["bounds"], w["pool"])
    rays = 0
    secs = 0.0
    mismatched = 0
    hit_mismatch = 0
    nframes = 0
    for s in range(max(1, cpu_frames // V)):
        frames.zero_()  # (a checkerboard frame writes half its pixels: both sides start from zeros)
        hits = torch.full((V, H, W), -1, dtype=torch.int64, device=frames.device)
        views = views_of(args.warmup + s, frames, hits)
        ctx.RenderViews(W, H, views, opts())
        gpu_fb, gpu_hit = frames.cpu().numpy(), hits.cpu().numpy()
        for j, v in enumerate(views):
            p = vxo.make_params(W, H, v["origin"], v["fwd"], v["up"], v["right"], frame_number=v["frame_number"], shadow=shadow,
                                bounce_samples=bounce, bounce_all_hits=args.bounce_all_hits, bounce_depth=args.bounce_depth,
                                mode=int((wflags or {}).get("mode", 0)), checkerboard=int(bool((wflags or {}).get("checkerboard", False))))
            t0 = time.perf_counter()
            out = world.render(p, fb=np.zeros((H, W, 4), np.uint8), want_hit=True, nthreads=cores)
            secs += time.perf_counter() - t0
            rays += out["stats"].total_rays()
            nframes += 1
            mismatched += int((gpu_fb[j] != out["fb"]).any(axis=2).sum())
            hit_mismatch += int((gpu_hit[j] != out["hit"]).sum())
    ctx.frame_stats()
    base = {"value": round(rays / secs / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frame(s) of the same workload (the frames of %d step(s)), %.1f s of CPU work" % (
                nframes, W, H, nframes // V, secs)}
    parity = {"frames": nframes, "pixels_differing": mismatched, "hit_voxel_indices_differing": hit_mismatch,
              "oracle": "cpu restatement (parity with the reference itself: unpinned)"}
    return base, parity


if __name__ == "__main__":
    main()
