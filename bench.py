#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow rays) of the sphere-tracing march on
the reference's sample scene at 1920x1080 with the Raymarching.mat settings.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one full frame per GPU through the C ABI (rm_blit_device) with the
framebuffers resident in HBM.  With N > 1 every rank renders its own frame each step
(pose sharding, weak scaling) and the finished frames are gathered to rank 0 with RCCL
on a side stream, --gather-batch frames per collective, overlapped with the following
renders; the timed region ends when every gather has landed.  Frames travel as RGB8
(the blit's alpha is the constant 255): a rank
finishes a 1080p frame every ~90 us, i.e. 70 GB/s of RGB8 towards the root, about what
one xGMI link carries in one direction; RGBA8 would be 94 GB/s.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X vector fp32 peak: 256 CU x 4 SIMD x 32 lanes x 2 (FMA) x 2 (packed) x 2.4 GHz


def algorithmic_flops(st: dict) -> float:
    """Literal flop count of the reference's expression tree: 292 per primary step
    (281 for scene().w + 11), 295 per shadow step, 2152 fixed per hit pixel."""
    return 292.0 * st["sum_steps_p"] + 295.0 * st["sum_steps_s"] + 2152.0 * st["n_hit"]


def cpu_baseline(w: int, h: int, props: dict) -> dict:
    """The CPU oracle (a port: the reference has no CPU path) on this host's cores."""
    from oracle import rm_oracle as O
    threads = min(O.lib().rmo_online_cpus(), int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    uni = O.Uniforms(props)
    O.render(uni, w, h, threads=threads)                       # warm-up
    best, st = None, None
    for _ in range(5):
        t0 = time.perf_counter()
        out = O.render(uni, w, h, threads=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
        st = out["stats"]
    rays = st["n_pixels"] + st["n_hit"]
    return {"value": rays / best / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"one {w}x{h} frame of the same workload, best of 5 after 1 warm-up ({best * 1e3:.1f} ms)",
            "tflops": algorithmic_flops(st) / best / 1e12}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the RCCL gather of finished frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-batch", type=int, default=8, help="frames per RCCL gather")
    ap.add_argument("--force-gather", action="store_true", help="developer check: run the RCCL gather path with a single rank")
    ap.add_argument("--kernel", type=int, default=0, help="kernel variant (0 = auto)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    use_pg = world > 1 or args.force_gather
    if use_pg:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    from raymarching_amd import Material, Raymarching, scenes
    from raymarching_amd.distributed import PipelinedGather
    from raymarching_amd.material import _ptr, current_stream_handle

    w, h = args.width, args.height
    mat = Material(device=local_rank)
    rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms())
    rm.bind()                                   # the SetMatrix block of OnRenderImage, once: the pose is fixed
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    if args.kernel:
        mat.set_kernel(args.kernel)

    depth = torch.empty((h, w), dtype=torch.float32, device=dev)
    hit = torch.empty((h, w), dtype=torch.uint8, device=dev)
    frame = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)          # the blit's RGBA8 target
    gather = (world > 1 and not args.no_gather) or args.force_gather
    pipe = None
    if gather:
        assert (w * h) % 4 == 0, "RGB8 transport packs 4 pixels at a time"
        pipe = PipelinedGather(torch.empty((h, w, 3), dtype=torch.uint8, device=dev), root=0, depth=2,
                               single_rank_too=args.force_gather, batch=args.gather_batch)
    lib, ctx = mat._lib, mat.handle

    def step() -> None:
        mat._ck(lib.rm_blit_device(ctx, None, None, _ptr(frame), _ptr(depth), _ptr(hit), w, h, current_stream_handle(dev)))
        if pipe:
            buf = pipe.acquire()            # slot of the current batch (stream-side wait when a new buffer is started)
            mat.pack_rgb(frame, buf)        # RGBA8 -> RGB8 on the render stream (4 us; on the side stream it was slower:
                                            # the extra events cost the host more than the kernel costs the GPU)
            pipe.submit(buf)                # every gather_batch-th call: the collective, on the side stream

    def fence() -> None:
        if pipe:
            pipe.drain()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    st = mat.stats()
    kernel_ms = e0.elapsed_time(e1) / args.steps        # launch stream: kernel + inter-launch gap

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([st["n_pixels"] + st["n_hit"]], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        rays_per_step = float(cnt.item())
    else:
        rays_per_step = float(st["n_pixels"] + st["n_hit"])

    if rank == 0:
        flops = algorithmic_flops(st)
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("width") == w and tj.get("height") == h:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        spath = os.path.join(ROOT, "profiles", "traffic_by_size.json")
        if traffic is None and os.path.exists(spath):
            try:
                traffic = json.load(open(spath)).get(f"{w}x{h}", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (primary+shadow)",
            "value": rays_per_step * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"default SampleScene {w}x{h}, Raymarching.mat (_MaxSteps 400, _MaxDist 50, _StepFactor .7), "
                                   "primary + soft-shadow rays, one frame per GPU per step",
                       "width": w, "height": h, "rays_per_frame": st["n_pixels"] + st["n_hit"],
                       "primary_rays": st["n_pixels"], "shadow_rays": st["n_hit"],
                       "scene_evals_per_frame": st["sum_steps_p"] + st["sum_steps_s"] + 7 * st["n_hit"],
                       "parallelism": ("single GPU" if world == 1 else
                                       f"pose-sharded x{world}" + (f", RCCL gather of RGB8 frames to rank 0, {args.gather_batch} frames per collective, overlapped with rendering" if gather else ", no gather"))},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic,
                         "flops_per_launch": flops, "kernel_ms": kernel_ms,
                         "sdf_evals_per_s": (st["sum_steps_p"] + st["sum_steps_s"] + 7 * st["n_hit"]) / (kernel_ms * 1e-3),
                         "framebuffer_GBps": (9.0 * st["n_pixels"]) / (kernel_ms * 1e-3) / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import rm_oracle as O
            out["cpu_baseline"] = cpu_baseline(w, h, O.props_from_material(mat))
        print(json.dumps(out), flush=True)

    mat.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
