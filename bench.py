#!/usr/bin/env python3
"""bench.py -- Mrays/s of the brickmap ray-tracing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one 1920x1080 frame of BASELINE.json configs[2]: primary ray + shadow ray for every primary hit +
1 bounce sample (reference gate `lDot == 0`), 8192x512x8192 procedurally generated brickmap (factor 32), the
four fixed cameras used round-robin.  With N > 1 the frame is sharded by interleaved 16-row strips, the
brickmap is replicated per GPU and the packed strips are gathered to rank 0 over RCCL (strong scaling: the
frame is fixed).  `value` = rays actually traced by all ranks / wall time of the K timed steps (inputs resident
in HBM, gather included).  Rank 0 prints ONE JSON line.
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
}

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
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="c3_8k_1080p_shadow_bounce", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--cpu-frames", type=int, default=48, help="frames of the workload timed on the host cores")
    ap.add_argument("--bounce-all-hits", type=int, default=0)
    ap.add_argument("--bounce-depth", type=int, default=1, help="2 = second bounce (BASELINE config 5; extension beyond the reference)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: all ranks share cuda:0 and the gather goes through gloo/host memory, to "
                         "exercise the N>1 code path on a one-GPU box (numbers are meaningless)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import voxelengine_amd as vx
    from voxelengine_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
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

    X, Y, Z, F, gen, W, H, shadow, bounce = WORKLOADS[args.workload]
    ctx = vx.Context(local_rank)
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

    plan = sharding.ShardPlan(W, H, sharding.STRIP_ROWS, world, rank)
    local = torch.zeros(plan.shard_bytes, dtype=torch.uint8, device=dev)
    frame = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
    shards = torch.zeros((world, plan.shard_bytes), dtype=torch.uint8, device=dev) if (rank == 0 and world > 1) else None

    def opts(frame_number, stats=False):
        return vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, bounce_all_hits=bool(args.bounce_all_hits),
                                bounce_depth=args.bounce_depth, frame_number=frame_number, strip_rows=plan.strip_rows, strip_count=world,
                                strip_index=rank, compact=world > 1, collect_stats=stats)

    def deinterleave(sh, fr):
        ctx.deinterleave_strips(W, H, plan.strip_rows, world, sh, plan.shard_bytes, fr)

    # N > 1: two-deep pipeline, the RCCL gather of frame k overlaps the render of frame k+1
    pipe = None
    if world > 1 and not rehearse:
        pipe = sharding.GatherPipeline(plan, lambda n: torch.zeros(n, dtype=torch.uint8, device=dev), frame, deinterleave)

    def step(i, ev=None):
        name, pos, f, u, r = cams[i % len(cams)]
        target = frame if world == 1 else (pipe.local(i) if pipe else local)
        if ev is not None:
            ev[0].record()
        ctx.RenderScreen(W, H, target, pos, f, u, r, opts(i + 1))
        if ev is not None:
            ev[1].record()
        if pipe:
            pipe.submit(i)
        elif world > 1:
            host = torch.zeros((world, plan.shard_bytes), dtype=torch.uint8) if rank == 0 else None
            sharding.gather_frame(plan, local.cpu(), host, frame,
                                  lambda sh, fr: (shards.copy_(sh), deinterleave(shards, fr)))

    def fence():
        if pipe:
            pipe.flush()
        torch.cuda.synchronize()
        if world > 1:
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
        i = args.warmup + k
        name, pos, f, u, r = cams[i % len(cams)]
        ctx.RenderScreen(W, H, frame if world == 1 else local, pos, f, u, r, opts(i + 1, stats=True))  # `local`: scratch
    sp = ctx.frame_stats()
    assert sp.total_rays() == rays_local, "ray counts differ between the timed and the counting pass"
    bytes_local = sp.algorithmic_bytes()

    tot = torch.tensor([float(rays_local), float(bytes_local), float(sum(kernel_ms))], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays_total, bytes_total, kernel_ms_total = [float(v) for v in tot.tolist()]
    dt = float(tmax.item())

    result = None
    if rank == 0:
        mrays = rays_total / dt / 1e6
        # dominant kernel = k_render: algorithmic bytes per launch / average launch duration (HIP events on the
        # launch stream), averaged over all ranks' launches
        n_launch = args.steps * world
        avg_kernel_s = kernel_ms_total / 1e3 / n_launch
        achieved = (bytes_total / n_launch) / avg_kernel_s / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
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
                "cameras": [c[0] for c in CAMERAS], "sharding": "interleaved %d-row strips, gather to rank 0" % plan.strip_rows
                if world > 1 else "none", "rays_per_step": round(rays_total / args.steps, 1),
                "world_build_s": round(t_build, 2), "bricks": int(info.nslots), "world_hbm_gib": round(info.hbm_bytes / 2**30, 3),
            },
            "roofline": {"bound": "hbm", "kernel": "k_render", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(bytes_total / n_launch, 1),
                         "avg_launch_ms": round(avg_kernel_s * 1e3, 4),
                         "bytes_per_ray": round(bytes_total / max(rays_total, 1.0), 1)},
        }
        if rehearse:  # the gathered frame of the last step must equal a single-GPU render of the same frame
            i = args.warmup + args.steps - 1
            name, pos, f, u, r = cams[i % len(cams)]
            full = torch.zeros_like(frame)
            ctx.RenderScreen(W, H, full, pos, f, u, r, vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce,
                                                                   bounce_all_hits=bool(args.bounce_all_hits),
                                                                   bounce_depth=args.bounce_depth, frame_number=i + 1))
            torch.cuda.synchronize()
            result["rehearsal"] = {"gathered_frame_equals_single_gpu_frame": bool(torch.equal(full, frame)),
                                   "note": "all ranks on one GPU over gloo: value is not a measurement"}
            ctx.frame_stats()
        if args.cpu_baseline == "auto" and world == 1:
            result["cpu_baseline"], result["parity"] = cpu_baseline(ctx, vx, cams, W, H, shadow, bounce, args, frame, opts)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


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


def cpu_baseline(ctx, vx, cams, W, H, shadow, bounce, args, frame, opts):
    """The CPU oracle (kind "port": this repo's C restatement of the reference algorithm) timed on the GPU box's
    host cores on a bounded sample of the same workload: `--cpu-frames` full frames.  The same frames double as a
    parity gate: the HIP framebuffer must equal the oracle's byte for byte."""
    import torch
    from oracle import vxo

    cores = host_cores()
    w = ctx.download_world()
    world = vxo.World.wrap(w["factor"], w["cdims"], w["coarse_bits"], w["brick_slot"], w["bounds"], w["pool"])
    rays = 0
    secs = 0.0
    mismatched = 0
    hit_mismatch = 0
    names = []
    for n in range(args.cpu_frames):
        i = args.warmup + n
        name, pos, f, u, r = cams[i % len(cams)]
        names.append(name)
        p = vxo.make_params(W, H, pos, f, u, r, frame_number=i + 1, shadow=shadow, bounce_samples=bounce,
                            bounce_all_hits=args.bounce_all_hits, bounce_depth=args.bounce_depth)
        t0 = time.perf_counter()
        out = world.render(p, fb=np.zeros((H, W, 4), np.uint8), want_hit=True, nthreads=cores)
        secs += time.perf_counter() - t0
        rays += out["stats"].total_rays()
        frame.zero_()
        hit = torch.full((H, W), -1, dtype=torch.int64, device=frame.device)
        ctx.RenderScreen(W, H, frame, pos, f, u, r, opts(i + 1), hit_aov=hit)
        mismatched += int((frame.cpu().numpy() != out["fb"]).any(axis=2).sum())
        hit_mismatch += int((hit.cpu().numpy() != out["hit"]).sum())
    ctx.frame_stats()
    base = {"value": round(rays / secs / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frame(s) of the same workload (cameras %s), %.1f s of CPU work" % (
                args.cpu_frames, W, H, ",".join(names), secs)}
    parity = {"frames": args.cpu_frames, "pixels_differing": mismatched, "hit_voxel_indices_differing": hit_mismatch,
              "oracle": "cpu restatement (parity with the reference itself: unpinned)"}
    return base, parity


if __name__ == "__main__":
    main()
