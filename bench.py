#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + soft-shadow rays) of the sphere-tracing march on the reference's
sample scene with the Raymarching.mat settings, through the C ABI with every framebuffer resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N = 1: a step is one 3840x2160 frame (rm_blit_device).  The JSON line carries the roofline of that frame's
kernel (algorithmic flops / HIP-event time on the launch stream, HBM traffic from two rocprofv3 counter passes
over a child run of this same frame), the CPU oracle timed on this host's cores, and secondary blocks: 1080p,
a moving camera, two frames in flight, the 4K directional depth pass + frame, the 256-primitive stress scene,
and the 64-pose 8K batch on one GPU (the one-GPU point of the scaling curve below).

N > 1: strong scaling of the batched-pose configuration: a step is 64 orbit poses at 7680x4320, pose k on rank
k mod N (rm_blit_poses_device), finished frames gathered to rank 0 over RCCL as compact tile records (only the
8x8 tiles that hold a hit pixel travel; rank 0 expands them over the background into dense RGBA8 frames) on a
side stream, overlapped with the rendering of the next batch.  The same line carries the time without the
gather, and at N = 2 the 4K frame rendered as interleaved row bands with the directional depth pass.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X vector fp32 peak: 256 CU x 4 SIMD x 32 lanes x 2 (FMA) x 2 (packed) x 2.4 GHz
PEAK_VALU_LANE_OPS = 78.6e12       # lane-instructions per second the chip can issue: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
N_POSES = 64                       # batched-pose configuration: 64 orbit cameras at 7680x4320
POSE_W, POSE_H = 7680, 4320
KERNEL_SOURCES = ("rm_kernels_common.h", "rm_math.h", "rm_device.h", "rm_k_queue.hip", "rm_k_tile.hip")


def algorithmic_flops(st: dict) -> float:
    """Literal flop count of the reference's expression tree: 292 per primary step
    (281 for scene().w + 11), 295 per shadow step, 2152 fixed per hit pixel."""
    return 292.0 * st["sum_steps_p"] + 295.0 * st["sum_steps_s"] + 2152.0 * st["n_hit"]


def kernel_source_sha() -> str:
    h = hashlib.sha256()
    for fn in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "raymarching_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# ---- HBM traffic: two rocprofv3 counter passes over a child run of the headline frame ------------------------------
def pmc_child(args) -> None:
    """what the counter passes profile: a few launches of the headline frame, nothing else"""
    import torch
    from raymarching_amd import Material, Raymarching, scenes
    from raymarching_amd.material import _ptr, current_stream_handle
    w, h, dev = args.width, args.height, "cuda:0"
    torch.cuda.set_device(0)
    mat = Material(device=0)
    Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms()).bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    frame = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
    depth = torch.empty((h, w), dtype=torch.float32, device=dev)
    hit = torch.empty((h, w), dtype=torch.uint8, device=dev)
    for _ in range(12):
        mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(frame), _ptr(depth), _ptr(hit), w, h, current_stream_handle(dev)))
    torch.cuda.synchronize()
    mat.close()


def measure_traffic(w: int, h: int) -> dict:
    """HBM bytes per launch of the frame kernel = FETCH_SIZE + WRITE_SIZE (KB units), each from a rocprofv3 --pmc
    pass of its own.  The frame reads at most 4 B per lane (no wide loads), so FETCH_SIZE is taken as it reads.
    A third pass counts the kernel's vector instructions (SQ_INSTS_VALU, wave-level) for the issue-rate figure."""
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not found"}
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        d = tempfile.mkdtemp(prefix="rm_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--width", str(w), "--height", str(h)]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=150, capture_output=True, text=True)
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return {"error": f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"}
            con = sqlite3.connect(dbs[0])
            try:
                rows = con.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection "
                                   "where counter_name = ? and (kernel_name like '%rm_queue_kernel%' or kernel_name like '%rm_tile_kernel%') "
                                   "group by kernel_name order by 3 desc", (counter,)).fetchall()
            finally:
                con.close()
            if not rows:
                return {"error": f"no frame kernel in the {counter} pass"}
            out[counter] = rows[0][1] / max(rows[0][2], 1) * (1.0 if counter == "SQ_INSTS_VALU" else 1024.0)
            out["kernel"] = rows[0][0]
        except subprocess.TimeoutExpired:
            if counter == "SQ_INSTS_VALU" and "WRITE_SIZE" in out:
                break
            return {"error": f"rocprofv3 --pmc {counter} timed out"}
        except Exception as e:                                   # a profiler problem must not take the benchmark down
            if counter == "SQ_INSTS_VALU" and "WRITE_SIZE" in out:
                break
            return {"error": f"{counter}: {type(e).__name__}: {e}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out["bytes"] = out["FETCH_SIZE"] + out["WRITE_SIZE"]
    return out


def committed_traffic(w: int, h: int):
    """fallback: the counter passes kept under profiles/, only if they were taken from these kernel sources"""
    path = os.path.join(ROOT, "profiles", "traffic_by_size.json")
    if not os.path.exists(path):
        return None, "no live counter pass and no profiles/traffic_by_size.json"
    e = json.load(open(path)).get(f"{w}x{h}")
    if not e:
        return None, f"no committed counter pass for {w}x{h}"
    if e.get("kernel_src_sha") != kernel_source_sha():
        return None, f"profiles/traffic_by_size.json[{w}x{h}] was taken from other kernel sources (stale)"
    return e.get("hbm_bytes_per_launch"), None


# ---- CPU baseline ------------------------------------------------------------------------------------------------
def cpu_baseline(props: dict, w_main: int, h_main: int) -> dict:
    """The CPU oracle (a port: the reference has no CPU path) on this host's cores: the headline frame itself (best of 2
    after a warm-up) and the two sizes the reference's own case and the 1080p configuration use, 640x360 and 1920x1080 (best of 5), with every online CPU and with the box's CPU
    share (OMP_NUM_THREADS); the faster thread count is reported.  About 25 core-seconds in all."""
    from oracle import rm_oracle as O
    online = O.lib().rmo_online_cpus()
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    counts = sorted({min(online, 256)} | ({min(share, online)} if share > 0 else set()))
    uni = O.Uniforms(props)
    res = {}
    sizes = [(640, 360, 5), (1920, 1080, 5)]
    if (w_main, h_main) not in [(a, b) for a, b, _ in sizes]:
        sizes.append((w_main, h_main, 2))
    for (w, h, reps) in sizes:
        # the frustum corners carry the aspect ratio: all sizes are 16:9, like the bound camera
        best = None
        for threads in counts:
            O.render(uni, w, h, threads=threads)                       # warm-up
            tb, st = None, None
            for _ in range(reps):
                t0 = time.perf_counter()
                out = O.render(uni, w, h, threads=threads)
                dt = time.perf_counter() - t0
                tb = dt if tb is None or dt < tb else tb
                st = out["stats"]
            if best is None or tb < best[0]:
                best = (tb, threads, st)
        tb, threads, st = best
        res[f"{w}x{h}"] = {"value": (st["n_pixels"] + st["n_hit"]) / tb / 1e6, "cores": threads, "ms": tb * 1e3, "best_of": reps,
                           "tflops": algorithmic_flops(st) / tb / 1e12}
    main = res[f"{w_main}x{h_main}"]
    return {"value": main["value"], "unit": "Mrays/s", "cores": main["cores"], "kind": "port",
            "host_cpus_online": online, "thread_counts_tried": counts,
            "sample": f"the headline frame itself ({w_main}x{h_main}, same scene and settings), best of {main['best_of']} after 1 warm-up "
                      f"({main['ms']:.1f} ms); 1920x1080: {res['1920x1080']['value']:.1f} Mrays/s ({res['1920x1080']['ms']:.1f} ms), "
                      f"640x360: {res['640x360']['value']:.1f} Mrays/s ({res['640x360']['ms']:.1f} ms)",
            "tflops": main["tflops"], "by_size": res}


# ---- helpers ---------------------------------------------------------------------------------------------------------
class Frame:
    """RGBA8 + depth + hit targets of n frames in HBM"""

    def __init__(self, torch, dev, w, h, n=None, color=None):
        shp = (h, w) if n is None else (n, h, w)
        self.color = color if color is not None else torch.empty(shp + (4,), dtype=torch.uint8, device=dev)
        self.depth = torch.empty(shp, dtype=torch.float32, device=dev)
        self.hit = torch.empty(shp, dtype=torch.uint8, device=dev)


RAMP_S = 0.03          # untimed lead-in of the single-GPU legs, see timed()
LEAD_IN_MS = []        # how long each lead-in took in the end


def timed(torch, fn, steps, warmup, fence=None, ramp_s=0.0):
    """W untimed + K timed calls of fn(i); returns (wall seconds, HIP-event ms per step on the current stream).
    ramp_s > 0 (single-process legs only: the number of calls depends on the clock): before the W warm-up calls, fn runs untimed for
    that long.  After an idle gap (creating a context, allocating buffers) the first ~10 ms of GPU work run below the steady clock;
    W = 10 frames of 0.25 ms end well inside that ramp and a 50-frame timed region then reads 4-8 % high.  The lead-in is extended in
    batches until the batch time has settled (the first leg of a process needs 100 ms and more); how long it took is reported."""
    fence = fence or torch.cuda.synchronize
    if ramp_s > 0.0:
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < ramp_s:
            for i in range(4):
                fn(i)
            fence()
        # The first leg of a process needs longer than that (the first 20 frames after a 30 ms lead-in read 8-10 % high there and
        # nowhere else): go on in untimed batches until two consecutive ones take the same time to 1 %, for at most a second.
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prev = None
        while time.perf_counter() - t_ramp < 1.0:
            ea.record()
            for i in range(max(min(steps, 50), 10)):
                fn(i)
            eb.record()
            fence()
            cur = ea.elapsed_time(eb)
            if prev is not None and abs(cur - prev) <= 0.01 * prev:
                break
            prev = cur
        LEAD_IN_MS.append((time.perf_counter() - t_ramp) * 1e3)
    for i in range(warmup):
        fn(i)
    fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    fence()
    el = time.perf_counter() - t0
    return el, e0.elapsed_time(e1) / max(steps, 1)


def pose_arrays(scenes, Raymarching, mat, ks, w, h):
    import numpy as np
    cams = [Raymarching(scenes.orbit_camera(k, N_POSES, w, h), mat, scenes.default_transforms()) for k in ks]
    fcs = np.ascontiguousarray(np.stack([r.GetFrustumCorners().m for r in cams]), dtype=np.float32)
    c2ws = np.ascontiguousarray(np.stack([r.cam.cameraToWorldMatrix.m for r in cams]), dtype=np.float32)
    pos = np.ascontiguousarray(np.stack([r.cam.transform.position for r in cams]), dtype=np.float32)
    return fcs, c2ws, pos


def blit_poses(mat, fcs, c2ws, pos, fr, w, h, stream):
    from raymarching_amd.material import _fp, _ptr
    mat._ck(mat._lib.rm_blit_poses_device(mat.handle, len(pos), _fp(fcs), _fp(c2ws), _fp(pos), _ptr(fr.color), _ptr(fr.depth),
                                          _ptr(fr.hit), w, h, stream))


def rays(st):
    return st["n_pixels"] + st["n_hit"]


def block(st, ms, extra=None):
    fl = algorithmic_flops(st)
    out = {"ms_per_step": ms, "Mrays_per_s": rays(st) / (ms * 1e-3) / 1e6, "rays_per_step": rays(st),
           "tflops": fl / (ms * 1e-3) / 1e12, "frac_of_peak": fl / (ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS}
    out.update(extra or {})
    return out


# ---- N = 1 -------------------------------------------------------------------------------------------------------------
def run_single(args) -> None:
    traffic, traffic_err, traffic_src, valu_insts = None, None, None, None
    if not args.no_traffic:
        t = measure_traffic(args.width, args.height)          # child processes, before this process touches the GPU
        if "bytes" in t:
            traffic, traffic_src = t["bytes"], {"how": "live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over a child run of this frame",
                                                "kernel": t["kernel"], "fetch_bytes": t["FETCH_SIZE"], "write_bytes": t["WRITE_SIZE"]}
            valu_insts = t.get("SQ_INSTS_VALU")
        else:
            traffic_err = t["error"]
    if traffic is None:
        traffic, err2 = committed_traffic(args.width, args.height)
        if traffic is not None:
            traffic_src = {"how": "profiles/traffic_by_size.json (same kernel sources)", "live_error": traffic_err}
        else:
            traffic_err = f"{traffic_err}; {err2}" if traffic_err else err2
            print(f"bench.py: WARNING: no HBM traffic figure: {traffic_err}", file=sys.stderr, flush=True)

    import numpy as np
    import torch
    torch.cuda.set_device(0)
    dev = "cuda:0"
    from raymarching_amd import Material, Raymarching, ShadowMapDirectional, scenes
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    from raymarching_amd.unity import Transform, vec3

    K, W = args.steps, args.warmup
    w, h = args.width, args.height
    stream = current_stream_handle(dev)

    def default_material(ww, hh):
        m = Material(device=0)
        Raymarching(scenes.default_camera(ww, hh), m, scenes.default_transforms()).bind()
        m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        if args.kernel:
            m.set_kernel(args.kernel)
        return m

    # ---- headline: one 3840x2160 frame per step, fixed pose -----------------------------------------------------
    mat = default_material(w, h)
    lib, ctx = mat._lib, mat.handle
    fr = Frame(torch, dev, w, h)

    def step(_i):
        mat._ck(lib.rm_blit_device(ctx, None, None, _ptr(fr.color), _ptr(fr.depth), _ptr(fr.hit), w, h, stream))

    elapsed, kernel_ms = timed(torch, step, K, W, ramp_s=RAMP_S)
    lead_in_ms = LEAD_IN_MS[-1]
    st = mat.stats()
    flops = algorithmic_flops(st)
    achieved = flops / (kernel_ms * 1e-3) / 1e12
    evals = st["sum_steps_p"] + st["sum_steps_s"] + 7 * st["n_hit"]
    out = {
        "metric": "Mrays/s (primary+shadow)",
        "value": rays(st) * K / elapsed / 1e6,
        "unit": "Mrays/s",
        "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"default SampleScene {w}x{h}, Raymarching.mat (_MaxSteps 400, _MaxDist 50, _StepFactor .7), "
                               "primary + soft-shadow rays, one frame per step, fixed pose",
                   "width": w, "height": h, "rays_per_frame": rays(st), "primary_rays": st["n_pixels"], "shadow_rays": st["n_hit"],
                   "scene_evals_per_frame": evals, "parallelism": "single GPU"},
        "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_error": None if traffic is not None else traffic_err,
                     "algorithmic_bytes": 9.0 * st["n_pixels"],
                     "flops_per_launch": flops, "kernel_ms": kernel_ms, "untimed_lead_in_ms": lead_in_ms,
                     "sum_steps_p": st["sum_steps_p"], "sum_steps_s": st["sum_steps_s"], "n_hit": st["n_hit"],
                     "sdf_evals_per_s": evals / (kernel_ms * 1e-3),
                     "framebuffer_GBps": (9.0 * st["n_pixels"]) / (kernel_ms * 1e-3) / 1e9,
                     "kernel_src_sha": kernel_source_sha(),
                     # what the counters say of the same kernel: wave-level vector instructions x 64 lanes per second, against the
                     # rate at which the chip can issue them (a wave64 instruction occupies its SIMD-32 for two cycles); this counts
                     # instructions executed, not the reference's flops, so hoisted uniform work and packed pairs do not flatter it
                     "valu_insts_per_launch": valu_insts,
                     "valu_issue_frac": (valu_insts * 64.0 / (kernel_ms * 1e-3) / PEAK_VALU_LANE_OPS) if valu_insts else None},
    }
    props = None
    if not args.no_cpu_baseline:
        from oracle import rm_oracle as O
        props = O.props_from_material(mat)

    sec = {}
    if not args.headline_only:
        Ks, Ws = min(K, 50), min(W, 10)
        # ---- moving camera at the headline size: a different orbit pose every step -------------------------------
        n_orbit = 16
        arr = [pose_arrays(scenes, Raymarching, mat, [k * (N_POSES // n_orbit)], w, h) for k in range(n_orbit)]
        pose_rays, pose_flops = [], []
        for fcs, c2ws, pos in arr:
            mat._ck(lib.rm_set_matrix(ctx, b"_FrustrumCorners", _fp(fcs[0])))
            mat._ck(lib.rm_set_matrix(ctx, b"_CameraInvViewMatrix", _fp(c2ws[0])))
            mat._ck(lib.rm_set_vector(ctx, b"_WorldSpaceCameraPos", _fp(np.ascontiguousarray(np.append(pos[0], 0.0), dtype=np.float32))))
            step(0)
            s2 = mat.stats()
            pose_rays.append(rays(s2)); pose_flops.append(algorithmic_flops(s2))
        pos4 = [np.ascontiguousarray(np.append(a[2][0], 0.0), dtype=np.float32) for a in arr]

        def step_orbit(i):
            fcs, c2ws, _ = arr[i % n_orbit]
            lib.rm_set_matrix(ctx, b"_FrustrumCorners", _fp(fcs[0]))
            lib.rm_set_matrix(ctx, b"_CameraInvViewMatrix", _fp(c2ws[0]))
            lib.rm_set_vector(ctx, b"_WorldSpaceCameraPos", _fp(pos4[i % n_orbit]))
            step(i)

        n_orb_steps = max(n_orbit, (Ks // n_orbit) * n_orbit)
        el, ms = timed(torch, step_orbit, n_orb_steps, n_orbit, ramp_s=RAMP_S)
        tot_rays = sum(pose_rays) * (n_orb_steps // n_orbit)
        tot_fl = sum(pose_flops) * (n_orb_steps // n_orbit)
        sec["moving_camera_4k"] = {"what": f"{w}x{h}, {n_orbit} orbit poses in turn (SetMatrix x2 + SetVector + Blit per step): the tile "
                                           "queue is ordered by the previous pose's costs",
                                   "ms_per_step": el / n_orb_steps * 1e3, "Mrays_per_s": tot_rays / el / 1e6,
                                   "tflops": tot_fl / (ms * n_orb_steps * 1e-3) / 1e12,
                                   "frac_of_peak": tot_fl / (ms * n_orb_steps * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS}
        # ---- a camera that moves as cameras do: half a degree of the orbit per frame ---------------------------------
        n_fine, per_turn = 48, 720
        fine = []
        for k in range(n_fine):
            r = Raymarching(scenes.orbit_camera(k, per_turn, w, h), mat, scenes.default_transforms())
            fine.append((np.ascontiguousarray(r.GetFrustumCorners().m, dtype=np.float32),
                         np.ascontiguousarray(r.cam.cameraToWorldMatrix.m, dtype=np.float32),
                         np.ascontiguousarray(np.append(r.cam.transform.position, 0.0), dtype=np.float32)))
        fine_rays, fine_flops = 0, 0.0

        def step_fine(i):
            fc, c2w, p4 = fine[i % n_fine]
            lib.rm_set_matrix(ctx, b"_FrustrumCorners", _fp(fc))
            lib.rm_set_matrix(ctx, b"_CameraInvViewMatrix", _fp(c2w))
            lib.rm_set_vector(ctx, b"_WorldSpaceCameraPos", _fp(p4))
            step(i)

        for k in range(n_fine):
            step_fine(k)
            s2 = mat.stats()
            fine_rays += rays(s2); fine_flops += algorithmic_flops(s2)
        el, ms = timed(torch, step_fine, n_fine, n_fine, ramp_s=RAMP_S)
        sec["smooth_camera_4k"] = {"what": f"{w}x{h}, {n_fine} consecutive poses of an orbit in half-degree steps (SetMatrix x2 + SetVector + "
                                           "Blit per step): the tile queue is ordered by the previous frame's costs, and a long tile also "
                                           "lists its eight neighbours",
                                   "ms_per_step": el / n_fine * 1e3, "Mrays_per_s": fine_rays / el / 1e6,
                                   "tflops": fine_flops / (ms * n_fine * 1e-3) / 1e12,
                                   "frac_of_peak": fine_flops / (ms * n_fine * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS}
        mat.close()

        # ---- 1920x1080 ------------------------------------------------------------------------------------------
        m2 = default_material(1920, 1080)
        f2 = Frame(torch, dev, 1920, 1080)

        def step2(_i, m=m2, f=f2, s=stream):
            m._ck(lib.rm_blit_device(m.handle, None, None, _ptr(f.color), _ptr(f.depth), _ptr(f.hit), 1920, 1080, s))

        el, ms = timed(torch, step2, max(Ks, 20), Ws, ramp_s=RAMP_S)
        s2 = m2.stats()
        sec["default_1080p"] = block(s2, ms, {"what": "default scene 1920x1080, one frame per step, fixed pose",
                                              "ms_per_step_wall": el / max(Ks, 20) * 1e3})
        # two frames in flight: two contexts on two streams, so one frame's long tail overlaps the next frame's bulk
        m3 = default_material(1920, 1080)
        f3 = Frame(torch, dev, 1920, 1080)
        ss = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        pair = [(m2, f2, ss[0].cuda_stream), (m3, f3, ss[1].cuda_stream)]

        def step_pair(_i):
            for m, f, s in pair:
                step2(0, m, f, s)

        n2 = max(Ks, 20)
        el, _ = timed(torch, step_pair, n2, Ws, ramp_s=RAMP_S)
        sec["default_1080p_two_in_flight"] = {"what": "the same 1080p frame from two contexts on two streams, frames_in_flight 2 (never the headline)",
                                              "frames_in_flight": 2, "ms_per_frame": el / (2 * n2) * 1e3,
                                              "Mrays_per_s": rays(s2) * 2 * n2 / el / 1e6,
                                              "frac_of_peak": algorithmic_flops(s2) * 2 * n2 / el / 1e12 / PEAK_FP32_VALU_TFLOPS}
        m2.close(); m3.close()
        del f2, f3

        # ---- the opt-in tolerance kernel on the headline frame (never the headline: its frames are not bit-equal) ----
        m7 = default_material(w, h)
        m7.set_kernel(4)

        def step7(_i):
            m7._ck(lib.rm_blit_device(m7.handle, None, None, _ptr(fr.color), _ptr(fr.depth), _ptr(fr.hit), w, h, stream))

        el, ms = timed(torch, step7, Ks, Ws, ramp_s=RAMP_S)
        s7 = m7.stats()
        sec["tolerance_kernel_4k"] = block(s7, ms, {"what": f"{w}x{h}, rm_set_kernel(ctx, 4): fused multiply-adds, raw v_sqrt_f32 and reciprocal multiplies in the two "
                                                            "march loops; within the port's tolerance of the strict frame, not bit-equal"})
        m7.close()

        # ---- the headline frame as Graphics.Blit(src, dest, mat) of the reference passes it: with a source image (what miss
        # pixels show) and the camera's depth texture (here: everything at the far plane, so the same rays are marched)
        m8 = default_material(w, h)
        src8 = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev)
        sd8 = torch.full((h, w), 1000.0, dtype=torch.float32, device=dev)

        def step8(_i):
            m8._ck(lib.rm_blit_device(m8.handle, _ptr(src8), _ptr(sd8), _ptr(fr.color), _ptr(fr.depth), _ptr(fr.hit), w, h, stream))

        el, ms = timed(torch, step8, Ks, Ws, ramp_s=RAMP_S)
        s8 = m8.stats()
        sec["with_source_image_and_depth_texture_4k"] = block(s8, ms, {"what": f"{w}x{h}, the headline frame with _MainTex and the depth texture bound "
                                                                               "(13 B/pixel of reads on top of the 9 B/pixel written)"})
        m8.close()
        del src8, sd8

        # ---- 4K with the directional depth pass (configuration 3 on one GPU) ----------------------------------
        m4 = default_material(w, h)
        t = Transform(position=vec3(-3, 3, 9), rotation=scenes.shadow_camera().transform.rotation, lossyScale=vec3(14, 8, 1))
        smd = ShadowMapDirectional(t, m4)
        smd.Update()
        corners = np.ascontiguousarray(np.stack(smd.corners).astype(np.float32).reshape(12))
        fwd = np.ascontiguousarray(t.forward.astype(np.float32))
        sm_depth = torch.empty((h, w), dtype=torch.float32, device=dev)

        def step3(_i):
            m4._ck(lib.rm_shadowmap_directional_device(m4.handle, _fp(corners), _fp(fwd), _ptr(sm_depth), None, w, h, stream))
            m4._ck(lib.rm_blit_device(m4.handle, None, None, _ptr(fr.color), _ptr(fr.depth), _ptr(fr.hit), w, h, stream))

        el, ms = timed(torch, step3, Ks, Ws, ramp_s=RAMP_S)
        m4._ck(lib.rm_shadowmap_directional_device(m4.handle, _fp(corners), _fp(fwd), _ptr(sm_depth), None, w, h, stream))
        sd = m4.stats()
        fl3 = flops + 292.0 * sd["sum_steps_p"]
        sec["depth_pass_plus_frame_4k"] = {"what": f"{w}x{h} ShadowMapDirectional depth pass + frame, back to back per step",
                                           "ms_per_step": ms, "Mrays_per_s": (rays(st) + sd["n_pixels"]) / (ms * 1e-3) / 1e6,
                                           "tflops": fl3 / (ms * 1e-3) / 1e12, "frac_of_peak": fl3 / (ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS}
        m4.close()
        del sm_depth

        # ---- stress scene: 256 primitives, 512 steps, 1080p (configuration 5) -----------------------------------
        m5 = Material(device=0)
        Raymarching(scenes.stress_camera(1920, 1080), m5, scenes.default_transforms()).bind()
        m5.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        for k_, v_ in scenes.STRESS_FLOATS.items():
            m5.SetFloat(k_, v_)
        prims = scenes.stress_primitives(256)
        m5.SetPrimitives(prims)
        f5 = Frame(torch, dev, 1920, 1080)

        def step5(_i):
            m5._ck(lib.rm_blit_device(m5.handle, None, None, _ptr(f5.color), _ptr(f5.depth), _ptr(f5.hit), 1920, 1080, stream))

        el, ms = timed(torch, step5, 3, 1)
        s5 = m5.stats()
        per_eval = sum(47 + (8, 19, 12, 15)[p["type"]] for p in prims) + 13 * (len(prims) - 1)
        fl5 = (per_eval + 11.0) * s5["sum_steps_p"] + (per_eval + 14.0) * s5["sum_steps_s"] + s5["n_hit"] * (7.0 * per_eval + 36.0 * (len(prims) - 1) + 77.0)
        sec["stress256_1080p"] = {"what": "256 smooth-union primitives (rotated, scaled), _MaxSteps 512, grazing camera, 1920x1080; exact kernel "
                                          "(automatic choice: one block per 8x8 tile, its four waves split the table; the marches skip the "
                                          "subtrees that provably cannot change a bit of the result)",
                                  "ms_per_step": ms, "Mrays_per_s": rays(s5) / (ms * 1e-3) / 1e6, "flops_per_eval": per_eval,
                                  "tflops": fl5 / (ms * 1e-3) / 1e12, "frac_of_peak": fl5 / (ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS}
        ls = m5.leaf_stats()
        leaf_frac = ls["visited"] / max(ls["possible"], 1)
        sec["stress256_1080p"].update({
            "executed_leaf_fraction": leaf_frac,
            "frac_of_peak_executed": sec["stress256_1080p"]["frac_of_peak"] * leaf_frac,
            "note": "tflops / frac_of_peak count the LITERAL tree (every primitive at every evaluation); the kernel executes "
                    "executed_leaf_fraction of those leaf evaluations (the rest provably cannot change a bit of the result), so "
                    "frac_of_peak_executed is the share of peak spent on arithmetic that was actually done"})
        m5.SetFloat("_RmTableSkip", 0.0)                      # the same kernel walking the whole table at every step
        el, ms0 = timed(torch, step5, 3, 1)
        sec["stress256_1080p"]["full_table_walk"] = {"ms_per_step": ms0, "frac_of_peak": fl5 / (ms0 * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS}
        m5.SetFloat("_RmTableSkip", 1.0)
        m5.set_kernel(1)                                      # the exact one-wave-per-tile kernel, for comparison
        el, ms1 = timed(torch, step5, 3, 1)
        sec["stress256_1080p"]["one_wave_per_tile_kernel"] = {"ms_per_step": ms1, "frac_of_peak": fl5 / (ms1 * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS}
        m5.close()
        del f5


        # ---- per-launch ramp and tail at the headline size: the same pose 8 times in ONE launch, and two frames in flight ----
        m9 = default_material(w, h)
        f9 = Frame(torch, dev, w, h, 8)
        r9 = Raymarching(scenes.default_camera(w, h), m9, scenes.default_transforms())
        fc9 = np.ascontiguousarray(np.stack([r9.GetFrustumCorners().m] * 8), dtype=np.float32)
        cw9 = np.ascontiguousarray(np.stack([r9.cam.cameraToWorldMatrix.m] * 8), dtype=np.float32)
        ps9 = np.ascontiguousarray(np.stack([r9.cam.transform.position] * 8), dtype=np.float32)

        def step9(_i):
            blit_poses(m9, fc9, cw9, ps9, f9, w, h, stream)

        el, ms = timed(torch, step9, max(3, Ks // 8), 2, ramp_s=RAMP_S)
        sec["default_4k_eight_in_one_launch"] = block(st, ms / 8.0, {
            "what": f"{w}x{h}, the headline pose 8 times in one rm_blit_poses_device launch, per frame: the same arithmetic with one "
                    "launch ramp and one tail per 8 frames; the difference to the headline is what a launch costs beyond its arithmetic",
            "launch_overhead_us": (kernel_ms - ms / 8.0) * 1e3})
        m9b = default_material(w, h)
        f9b = Frame(torch, dev, w, h)
        ss9 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

        def step9_pair(_i):
            mat9a._ck(lib.rm_blit_device(mat9a.handle, None, None, _ptr(fr.color), _ptr(fr.depth), _ptr(fr.hit), w, h, ss9[0].cuda_stream))
            m9b._ck(lib.rm_blit_device(m9b.handle, None, None, _ptr(f9b.color), _ptr(f9b.depth), _ptr(f9b.hit), w, h, ss9[1].cuda_stream))

        mat9a = default_material(w, h)
        el, _ = timed(torch, step9_pair, Ks, Ws, ramp_s=RAMP_S)
        sec["default_4k_two_in_flight"] = {"what": f"{w}x{h}, the headline frame from two contexts on two streams, frames_in_flight 2 (never the headline)",
                                           "frames_in_flight": 2, "ms_per_frame": el / (2 * Ks) * 1e3,
                                           "Mrays_per_s": rays(st) * 2 * Ks / el / 1e6,
                                           "frac_of_peak": flops * 2 * Ks / el / 1e12 / PEAK_FP32_VALU_TFLOPS}
        m9.close(); m9b.close(); mat9a.close()
        del f9, f9b

        # ---- frames WITHOUT a previous frame of their shape (a context's first frame, a resize): the frame height alternates between
        # h and h - 8 rows, so every launch starts from cleared queue state; with the tile-cost prediction pass and without ----
        nh_ = {}
        for (ww, hh) in ((w, h), (1920, 1080)):
            fr_n = Frame(torch, dev, ww, hh)
            for label, floats in (("predicted", {}), ("unordered", {"_RmPredictTiles": 0.0})):
                mn = default_material(ww, hh)
                for k_, v_ in floats.items():
                    mn.SetFloat(k_, v_)

                def step_nh(_i, mn=mn, fr_n=fr_n, ww=ww, hh=hh):
                    mn._ck(lib.rm_blit_device(mn.handle, None, None, _ptr(fr_n.color), _ptr(fr_n.depth), _ptr(fr_n.hit), ww, hh - 8, stream))
                    mn._ck(lib.rm_blit_device(mn.handle, None, None, _ptr(fr_n.color), _ptr(fr_n.depth), _ptr(fr_n.hit), ww, hh, stream))

                el, ms = timed(torch, step_nh, Ks, Ws, ramp_s=RAMP_S)
                nh_[f"{ww}x{hh}_{label}_ms_per_frame"] = ms / 2.0
                mn.close()
            del fr_n
        sec["frames_without_history"] = dict(nh_, what="ms per frame when no launch finds a previous frame of its own shape (two frame heights in turn): "
                                             "the persistent kernel's queue ordered by the prediction pass (one capped ray per 8x8 tile, "
                                             "run ahead of the frame and included in the time) against the plain scattered order; "
                                             "the headline and default_1080p are the same frames WITH history")

        # ---- the shadow map consumed: depth pass from the light + frame whose shadow term is a lookup (_RmShadowMode 1 / 2) ----
        from raymarching_amd import ShadowMap
        from raymarching_amd.unity import Matrix4x4
        for mode, tag in ((2, "directional"), (1, "perspective")):
            mf, msm = default_material(w, h), default_material(w, h)       # the frame's material and the depth pass's
            smap = torch.empty((h, w), dtype=torch.float32, device=dev)
            if mode == 2:
                smd2 = ShadowMapDirectional(scenes.light_rectangle(), msm)
                smd2.Update()
                cn = np.ascontiguousarray(np.stack(smd2.corners).astype(np.float32).reshape(12))
                fw = np.ascontiguousarray(smd2.transform.forward.astype(np.float32))
                kk = Matrix4x4.identity()
                for i_ in range(4):
                    kk.SetRow(i_, smd2.corners[i_])
                mf.SetMatrix("_RmShadowMapCorners", kk)
                mf.SetVector("_RmShadowMapForward", smd2.transform.forward)

                def depth_pass():
                    msm._ck(lib.rm_shadowmap_directional_device(msm.handle, _fp(cn), _fp(fw), _ptr(smap), None, w, h, stream))
            else:
                lc = scenes.light_camera(w, h)
                smc = ShadowMap(lc, msm)
                msm.SetMatrix("_FrustrumCorners", smc.GetFrustumCorners())
                msm.SetMatrix("_CameraInvViewMatrix", lc.cameraToWorldMatrix)
                msm.SetVector("_WorldSpaceCameraPos", lc.transform.position)
                mf.SetMatrix("_ShadowMapVPMatrix", lc.projectionMatrix * lc.worldToCameraMatrix)

                def depth_pass():
                    msm._ck(lib.rm_shadowmap_render_device(msm.handle, _ptr(smap), None, w, h, stream))
            mf.SetTexture("_ShadowMap", smap)
            mf.SetFloat("_RmShadowMode", float(mode))

            def frame_lookup(_i):
                mf._ck(lib.rm_blit_device(mf.handle, None, None, _ptr(fr.color), _ptr(fr.depth), _ptr(fr.hit), w, h, stream))

            def both(_i):
                depth_pass()
                frame_lookup(_i)

            depth_pass()
            sdp = msm.stats()
            el, ms_f = timed(torch, frame_lookup, Ks, Ws, ramp_s=RAMP_S)
            sl = mf.stats()
            el, ms_b = timed(torch, both, Ks, Ws, ramp_s=RAMP_S)
            fl_f = 292.0 * sl["sum_steps_p"] + (2152.0 - 9.0 + 30.0) * sl["n_hit"]     # no shadow march; the lookup is ~30 flops per hit pixel
            sec[f"shadowmap_lookup_{tag}_4k"] = {
                "what": f"{w}x{h}: the frame with its shadow term looked up in a {w}x{h} {tag} shadow map rendered from the light "
                        f"(_RmShadowMode {mode}; persistent kernel), instead of the soft-shadow march of the headline",
                "frame_only_ms": ms_f, "frame_only_frac_of_peak": fl_f / (ms_f * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                "depth_pass_plus_frame_ms": ms_b, "headline_frame_with_shadow_march_ms": kernel_ms,
                "Mrays_per_s_depth_pass_plus_frame": (sl["n_pixels"] + sdp["n_pixels"]) / (ms_b * 1e-3) / 1e6,
                "primary_evals": sl["sum_steps_p"], "n_hit": sl["n_hit"]}
            mf.close(); msm.close()
            del smap


        # ---- the host-pointer entry points (PCIe-inclusive; never the headline): what RaymarchingNative.cs calls ----------
        import ctypes as C
        mh = default_material(1920, 1080)
        npx = 1920 * 1080
        hb = [[lib.rm_host_alloc(mh.handle, npx * 4), lib.rm_host_alloc(mh.handle, npx * 4), lib.rm_host_alloc(mh.handle, npx)] for _ in range(2)]
        hs = [lib.rm_host_alloc(mh.handle, npx * 4), lib.rm_host_alloc(mh.handle, npx * 4)]       # _MainTex and the depth texture
        if all(all(x) for x in hb) and all(hs):
            C.memset(hs[0], 64, npx * 4)
            np.ctypeslib.as_array(C.cast(hs[1], C.POINTER(C.c_float)), shape=(npx,))[:] = 1000.0
            nh = 60

            def host_loop(with_inputs, pipelined):
                a, b = (hs[0], hs[1]) if with_inputs else (None, None)
                for _ in range(5):
                    mh._ck(lib.rm_blit(mh.handle, a, b, hb[0][0], hb[0][1], hb[0][2], 1920, 1080))
                t0_ = time.perf_counter()
                if not pipelined:
                    for _ in range(nh):
                        mh._ck(lib.rm_blit(mh.handle, a, b, hb[0][0], hb[0][1], hb[0][2], 1920, 1080))
                else:
                    tk = [C.c_longlong(), C.c_longlong()]
                    for i_ in range(nh):
                        o = hb[i_ & 1]
                        mh._ck(lib.rm_blit_async(mh.handle, a, b, o[0], o[1], o[2], 1920, 1080, C.byref(tk[i_ & 1])))
                        if i_ > 0:
                            mh._ck(lib.rm_blit_wait(mh.handle, tk[(i_ - 1) & 1].value))      # frame i-1 is shown while frame i marches
                    mh._ck(lib.rm_blit_wait(mh.handle, tk[(nh - 1) & 1].value))
                return (time.perf_counter() - t0_) / nh * 1e3

            sec["host_pointer_blit_1080p"] = {
                "what": "1920x1080 through the host-pointer entry points into rm_host_alloc memory, ms per frame, PCIe copies included: "
                        "rm_blit (synchronous) against rm_blit_async / rm_blit_wait with two frames in flight; without inputs, and with "
                        "_MainTex + the depth texture uploaded every frame (8 B/pixel up, 9 B/pixel down)",
                "rm_blit_ms": host_loop(False, False), "rm_blit_async_pipelined_ms": host_loop(False, True),
                "rm_blit_with_src_and_depth_ms": host_loop(True, False), "rm_blit_async_pipelined_with_src_and_depth_ms": host_loop(True, True),
                "kernel_only_ms": sec["default_1080p"]["ms_per_step"]}
        for x in [y for o in hb for y in o] + hs:
            if x:
                lib.rm_host_free(mh.handle, x)
        mh.close()

        # ---- the other scene() bodies and one frame with frag's optional lines, 1080p close-ups ---------------------------
        def closeup_material(body, extra=None):
            mm = Material(device=0)
            ts = scenes.default_transforms()
            Raymarching(scenes.closeup_camera(ts[0].position, 1920, 1080), mm, ts).bind()
            mm.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            mm.SetFloat("_RmSceneBody", float(body))
            for k_, v_ in (extra or {}).items():
                mm.SetFloat(k_, v_)
            return mm

        fb = Frame(torch, dev, 1920, 1080)
        # literal flops of one evaluation (wrapper 47 + body; a transcendental counts 1): menger has a fixed trip count,
        # 12 x (3 swaps + euler rotate 45 + 6 + 2) + 8 = 680 distance-only, + 12 x 30 for the orbit traps
        bodies = (("menger_closeup_1080p", 1, None, 47 + 680, 47 + 1040, "exact: fixed 12 iterations"),
                  ("mandelbulb_closeup_1080p", 3, None, 47 + 12 * 32 + 5, 47 + 12 * 62 + 5,
                   "UPPER BOUND: 12 iterations counted, the loop leaves early once r > 10 (the kernel does not count iterations)"),
                  ("fx_bounce_step_heat_1080p", 0, {"_RmBounce": 1.0, "_RmStepHeat": 1.0}, 281, 389, "exact (counters are the camera ray's: the bounce march is not counted)"))
        for tag, body, extra, fw_, ff_, how in bodies:
            if body == 0:
                mm = default_material(1920, 1080)
                for k_, v_ in extra.items():
                    mm.SetFloat(k_, v_)
            else:
                mm = closeup_material(body, extra)

            def stepb(_i, mm=mm):
                mm._ck(lib.rm_blit_device(mm.handle, None, None, _ptr(fb.color), _ptr(fb.depth), _ptr(fb.hit), 1920, 1080, stream))

            el, ms = timed(torch, stepb, 10, 2)
            sb = mm.stats()
            evals_b = sb["sum_steps_p"] + sb["sum_steps_s"] + 7 * sb["n_hit"]
            flb = (fw_ + 11.0) * sb["sum_steps_p"] + (fw_ + 14.0) * sb["sum_steps_s"] + sb["n_hit"] * (6.0 * fw_ + ff_ + 77.0)
            sec[tag] = {"what": f"1920x1080, _RmSceneBody {body}" + (f", {extra}" if extra else "") + " (tile kernels)",
                        "ms_per_step": ms, "Mrays_per_s": rays(sb) / (ms * 1e-3) / 1e6, "sdf_evals_per_s": evals_b / (ms * 1e-3),
                        "flops_per_eval_distance_only": fw_, "flop_count": how,
                        "tflops": flb / (ms * 1e-3) / 1e12, "frac_of_peak": flb / (ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS}
            mm.close()
        del fb

        # ---- 64 orbit poses at 7680x4320 on this one GPU: the N = 1 point of the strong-scaling curve ----------------
        m6 = default_material(POSE_W, POSE_H)
        per_call = 8
        f6 = Frame(torch, dev, POSE_W, POSE_H, per_call)
        batches = [pose_arrays(scenes, Raymarching, m6, list(range(b, b + per_call)), POSE_W, POSE_H) for b in range(0, N_POSES, per_call)]

        def step6(_i):
            for fcs, c2ws, pos in batches:
                blit_poses(m6, fcs, c2ws, pos, f6, POSE_W, POSE_H, stream)

        tot = {"n_pixels": 0, "n_hit": 0, "sum_steps_p": 0, "sum_steps_s": 0}
        for fcs, c2ws, pos in batches:                       # untimed: the counters of every batch (also the warm-up)
            blit_poses(m6, fcs, c2ws, pos, f6, POSE_W, POSE_H, stream)
            s6 = m6.stats()
            for k_ in tot:
                tot[k_] += s6[k_]
        el, ms = timed(torch, step6, 2, 0)
        sec["poses_64x8k_one_gpu"] = block(tot, el / 2 * 1e3, {"what": f"{N_POSES} orbit poses at {POSE_W}x{POSE_H}, {per_call} per rm_blit_poses_device call, no gather: "
                                                                       "what --gpus N > 1 runs, on one GPU"})
        m6.close()
        del f6
    else:
        mat.close()
    out["secondary"] = sec
    if props is not None:
        out["cpu_baseline"] = cpu_baseline(props, w, h)
    print(json.dumps(out), flush=True)


# ---- N > 1 -------------------------------------------------------------------------------------------------------------
def run_multi(args) -> None:
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RCCL prints a version banner on stdout when its first communicator comes up: everything this process and its libraries
    # write to fd 1 goes to stderr until the result line, which is then the only thing on stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # before the first GPU call of this process
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())      # rehearsal: the ranks share what GPUs there are
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    gloo_only = args.backend == "gloo"          # developer rehearsal: several ranks on one GPU, every exchange through the host
    if gloo_only:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    cdev = "cpu" if gloo_only else dev          # where the small reductions of this function live
    try:
        counts_group = dist.new_group(backend="gloo")           # record counts travel as host tensors
    except Exception as e:                                      # no host-side transport on this node: RCCL carries them too
        print(f"bench.py: no gloo group ({e}); record counts go over RCCL", file=sys.stderr, flush=True)
        counts_group = None

    from raymarching_amd import Material, Raymarching, ShadowMapDirectional, scenes
    from raymarching_amd.distributed import DenseFrames, TileGather, poses_of, render_frame_bands
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    from raymarching_amd.unity import Transform, vec3

    K, W = args.steps, args.warmup
    w, h = POSE_W, POSE_H
    stream = current_stream_handle(dev)
    mat = Material(device=local_rank)
    Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms()).bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    if args.kernel:
        mat.set_kernel(args.kernel)
    if rank == 0 and args.root_blocks_per_cu > 0:
        # leave room on the root's CUs for the expansion kernels that run beside the rendering
        mat.SetFloat("_RmBlocksPerCU", float(args.root_blocks_per_cu))

    if N_POSES % world != 0:
        raise SystemExit(f"{N_POSES} poses do not divide over {world} ranks")
    mine = poses_of(N_POSES, rank, world)
    per_rank = len(mine)
    n_batches = max(1, min(args.pose_batches, per_rank))
    per = (per_rank + n_batches - 1) // n_batches
    groups = [mine[i:i + per] for i in range(0, per_rank, per)]          # the same split on every rank
    tiles = ((w + 7) // 8) * ((h + 7) // 8)
    # Rank 0 keeps every pose as a dense RGBA8 frame, ordered [rank that rendered it][its local index] so that one
    # batch of one rank is contiguous: its own frames are rendered in place, the others are expanded in place.
    dense = torch.empty((world, per_rank, h, w, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
    frames = [Frame(torch, dev, w, h, len(g), color=(dense[0, b * per:b * per + len(g)] if rank == 0 else None))
              for b, g in enumerate(groups)]
    arrays = [pose_arrays(scenes, Raymarching, mat, g, w, h) for g in groups]
    gather = TileGather(lambda fr, ht, rec, cnt: mat.compact_frames(fr, ht, rec, cnt), capacity_records=per * tiles, device=dev,
                        root=0, count_group=counts_group, depth=2)

    expand = [True]
    store = DenseFrames(expand=lambda rec, n, dst: mat.expand_frames(rec, n, dst),
                        update=None if args.full_expand else (lambda old, n_old, rec, n, dst: mat.update_frames(old, n_old, rec, n, dst)))

    def on_received_for(b):
        def on_received(p, records, n):
            if p != 0 and expand[0]:                     # the root's own frames are dense already
                store.deliver((p, b), records, n, dense[p, b * per:b * per + len(groups[b])])
        return on_received

    pending = []

    def render(b):
        fcs, c2ws, pos = arrays[b]
        blit_poses(mat, fcs, c2ws, pos, frames[b], w, h, stream)

    def step_render(_i):
        for b in range(len(groups)):
            render(b)

    def step_gather(_i):
        for b in range(len(groups)):
            render(b)
            t = gather.submit(frames[b].color, frames[b].hit, tag=b)
            if pending:                                  # the previous batch: its count is ready while this one renders
                tp = pending.pop(0)
                gather.complete(tp, on_received_for(tp.tag))
            pending.append(t)

    def fence_gather():
        while pending:
            tp = pending.pop(0)
            gather.complete(tp, on_received_for(tp.tag))
        gather.drain()
        dist.barrier()
        torch.cuda.synchronize()

    def fence():
        dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # counters of this rank's share (untimed)
    tot = np.zeros(4, dtype=np.float64)
    for b in range(len(groups)):
        render(b)
        s = mat.stats()
        tot += [s["n_pixels"], s["n_hit"], s["sum_steps_p"], s["sum_steps_s"]]
    tt = torch.tensor(tot, dtype=torch.float64, device=cdev)
    dist.all_reduce(tt, op=dist.ReduceOp.SUM)
    st = dict(zip(["n_pixels", "n_hit", "sum_steps_p", "sum_steps_s"], [int(v) for v in tt.tolist()]))

    el_r, _ = timed(torch, step_render, K, W, fence)
    el_r = reduce_max(el_r)
    expand[0] = False
    el_t, _ = timed(torch, step_gather, K, W, fence_gather)
    el_t = reduce_max(el_t)
    expand[0] = True
    el_g, _ = timed(torch, step_gather, K, W, fence_gather)
    el_g = reduce_max(el_g)
    sent = torch.tensor([float(gather.bytes_sent)], dtype=torch.float64, device=cdev)
    dist.all_reduce(sent, op=dist.ReduceOp.SUM)
    sent_per_step = float(sent.item()) / (2 * (K + W))

    verified = None
    if args.verify and rank == 0:
        # rank 0 renders, itself, one pose of every other rank's last batch and compares it with the frame it was sent
        verified = 0
        chk = Frame(torch, dev, w, h, 1)
        for p in range(1, world):
            theirs = poses_of(N_POSES, p, world)
            j = len(theirs) - 1                                   # local index of that rank's last pose
            fcs, c2ws, pos = pose_arrays(scenes, Raymarching, mat, [theirs[j]], w, h)
            blit_poses(mat, fcs, c2ws, pos, chk, w, h, stream)
            torch.cuda.synchronize()
            if not torch.equal(chk.color[0], dense[p, j]):
                raise SystemExit(f"bench.py --verify: the frame of pose {theirs[j]} delivered by rank {p} differs from rank 0's own rendering")
            verified += 1
        del chk

    # ---- the same workload on ONE GPU: rank 0 alone renders all 64 poses (the others wait), after the timed legs -------------
    # The N = 1 line of this benchmark is the 4K frame, a different workload; a scaling curve must be read from
    # `one_gpu_same_workload` / `speedup` of each N > 1 line, never from the N = 1 line's value.
    one_gpu_ms = None
    if not args.no_one_gpu_reference:
        if rank == 0:
            per_call = min(8, N_POSES)
            f1 = Frame(torch, dev, w, h, per_call)
            b1 = [pose_arrays(scenes, Raymarching, mat, list(range(b, min(b + per_call, N_POSES))), w, h) for b in range(0, N_POSES, per_call)]

            def step_one(_i):
                for fcs, c2ws, pos in b1:
                    blit_poses(mat, fcs, c2ws, pos, f1, w, h, stream)

            k1 = max(1, min(K, 3))
            el1, _ = timed(torch, step_one, k1, 1)
            one_gpu_ms = el1 / k1 * 1e3
            del f1
        fence()

    sec = {}
    if world == 2 and not args.headline_only:
        # ---- configuration 3 on two GPUs: 3840x2160 as interleaved 8-row bands + the directional depth pass on both --------
        bw, bh, band = 3840, 2160, 8
        m3 = Material(device=local_rank)
        Raymarching(scenes.default_camera(bw, bh), m3, scenes.default_transforms()).bind()
        m3.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        t = Transform(position=vec3(-3, 3, 9), rotation=scenes.shadow_camera().transform.rotation, lossyScale=vec3(14, 8, 1))
        smd = ShadowMapDirectional(t, m3)
        smd.Update()
        corners = np.ascontiguousarray(np.stack(smd.corners).astype(np.float32).reshape(12))
        fwd = np.ascontiguousarray(t.forward.astype(np.float32))
        rows = m3._lib.rm_band_rows(bh, band, rank, world)
        sm_depth = torch.empty((rows, bw), dtype=torch.float32, device=dev)          # this rank's bands of the depth map
        sm_recv = [torch.empty_like(sm_depth) for _ in range(world)] if rank == 0 else None
        sm_h = torch.empty(sm_depth.shape, dtype=sm_depth.dtype) if gloo_only else None
        sm_recv_h = [torch.empty_like(sm_h) for _ in range(world)] if (gloo_only and rank == 0) else None
        c = torch.empty((rows, bw, 4), dtype=torch.uint8, device=dev)
        d = torch.empty((rows, bw), dtype=torch.float32, device=dev)
        hm = torch.empty((rows, bw), dtype=torch.uint8, device=dev)
        rgb = torch.empty((rows, bw, 3), dtype=torch.uint8, device=dev)
        recv = [torch.empty_like(rgb) for _ in range(world)] if rank == 0 else None
        rgb_h = torch.empty(rgb.shape, dtype=rgb.dtype) if gloo_only else None
        recv_h = [torch.empty_like(rgb_h) for _ in range(world)] if (gloo_only and rank == 0) else None

        def step3(_i):
            m3._ck(m3._lib.rm_shadowmap_directional_bands_device(m3.handle, _fp(corners), _fp(fwd), _ptr(sm_depth), None, bw, bh,
                                                                 band, rank, world, stream))
            m3._ck(m3._lib.rm_blit_bands_device(m3.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), bw, bh, band, rank, world, stream))
            m3.pack_rgb(c, rgb)
            if gloo_only:
                rgb_h.copy_(rgb); sm_h.copy_(sm_depth)
                dist.gather(rgb_h, recv_h, dst=0)
                dist.gather(sm_h, sm_recv_h, dst=0)
                if rank == 0:
                    for a, b_ in zip(recv + sm_recv, recv_h + sm_recv_h):
                        a.copy_(b_)
            else:
                dist.gather(rgb, recv, dst=0)
                dist.gather(sm_depth, sm_recv, dst=0)

        el3, _ = timed(torch, step3, min(K, 50), min(W, 10), fence)
        el3 = reduce_max(el3)
        s3 = m3.stats()
        c3 = torch.tensor([float(rays(s3) + rows * bw)], dtype=torch.float64, device=cdev)
        dist.all_reduce(c3, op=dist.ReduceOp.SUM)
        one3 = None
        if rank == 0 and not args.no_one_gpu_reference:
            # the same two passes, whole, on rank 0 alone (no gather): the one-GPU point of this configuration
            fc3, fd3, fh3 = (torch.empty((bh, bw, 4), dtype=torch.uint8, device=dev), torch.empty((bh, bw), dtype=torch.float32, device=dev),
                             torch.empty((bh, bw), dtype=torch.uint8, device=dev))
            smf = torch.empty((bh, bw), dtype=torch.float32, device=dev)

            def step3_one(_i):
                m3._ck(m3._lib.rm_shadowmap_directional_device(m3.handle, _fp(corners), _fp(fwd), _ptr(smf), None, bw, bh, stream))
                m3._ck(m3._lib.rm_blit_device(m3.handle, None, None, _ptr(fc3), _ptr(fd3), _ptr(fh3), bw, bh, stream))

            e1, _ = timed(torch, step3_one, min(K, 50), min(W, 10))
            one3 = e1 / min(K, 50) * 1e3
            del fc3, fd3, fh3, smf
        fence()
        sec["bands_4k_with_depth_pass_2gpu"] = {"what": "3840x2160 frame AND the directional depth pass, both as interleaved 8-row bands over 2 GPUs "
                                                        "(rm_blit_bands_device, rm_shadowmap_directional_bands_device) + gather of the frame bands as RGB8 "
                                                        "and of the depth-map bands as fp32 to rank 0, per step",
                                                "ms_per_step": el3 / min(K, 50) * 1e3, "Mrays_per_s": float(c3.item()) * min(K, 50) / el3 / 1e6}
        if one3 is not None:
            sec["bands_4k_with_depth_pass_2gpu"]["one_gpu_same_workload_ms"] = one3
            sec["bands_4k_with_depth_pass_2gpu"]["speedup"] = one3 / (el3 / min(K, 50) * 1e3)
        m3.close()

    if rank == 0:
        flops = algorithmic_flops(st)
        out = {
            "metric": "Mrays/s (primary+shadow)",
            "value": rays(st) * K / el_g / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": el_g / K * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{N_POSES} orbit poses of the default SampleScene at {w}x{h} per step, Raymarching.mat settings, primary + "
                                   "soft-shadow rays, pose k on rank k mod N, finished frames gathered to rank 0 as compact tile records "
                                   "and written there into dense RGBA8 frames that rank 0 keeps (" + ("every frame rewritten whole" if args.full_expand
                                   else "after the first delivery only the tiles of the previous and the new records are written") + ")",
                       "width": w, "height": h, "poses": N_POSES, "rays_per_step": rays(st), "primary_rays": st["n_pixels"],
                       "shadow_rays": st["n_hit"], "pose_batches_per_rank": len(groups),
                       "parallelism": f"pose-sharded x{world}, " + (
                           "gloo REHEARSAL: ranks share what GPUs there are and every exchange goes through the host (times mean nothing)" if gloo_only
                           else "RCCL point-to-point gather of hit tiles to rank 0 on a side stream, overlapped with rendering"),
                       "backend": "gloo" if gloo_only else "nccl (RCCL)",
                       "gather_payload_bytes_per_step": sent_per_step,
                       "dense_rgb8_bytes_per_step_if_not_compacted": 3.0 * w * h * (N_POSES - len(poses_of(N_POSES, 0, world)))},
            "roofline": {"bound": "valu_fp32", "achieved": flops * K / el_g / 1e12, "peak": PEAK_FP32_VALU_TFLOPS * world, "unit": "TFLOP/s",
                         "frac": flops * K / el_g / 1e12 / (PEAK_FP32_VALU_TFLOPS * world), "traffic": None,
                         "flops_per_step": flops, "note": "whole job over all ranks, wall clock with the gather inside; per-kernel figures are the N = 1 line's"},
            "transport_only": {"ms_per_step": el_t / K * 1e3, "value": rays(st) * K / el_t / 1e6,
                               "what": "the gather without the expansion on rank 0: the tile records of every pose are resident in rank 0's HBM"},
            "no_gather": {"ms_per_step": el_r / K * 1e3, "value": rays(st) * K / el_r / 1e6,
                          "what": "the same poses rendered, frames left on the rank that rendered them"},
            "secondary": sec,
        }
        if one_gpu_ms is not None:
            out["one_gpu_same_workload"] = {"ms_per_step": one_gpu_ms, "value": rays(st) / (one_gpu_ms * 1e-3) / 1e6,
                                            "what": f"the same {N_POSES} poses rendered by rank 0 alone ({min(8, N_POSES)} per rm_blit_poses_device call, "
                                                    "no gather), measured in this run after the timed legs"}
            out["speedup"] = one_gpu_ms / (el_g / K * 1e3)
            out["speedup_no_gather"] = one_gpu_ms / (el_r / K * 1e3)
        if verified is not None:
            out["verified_frames_from_other_ranks"] = verified
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    mat.close()
    dist.destroy_process_group()
    os.close(real_stdout)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: starts `python -m torch.distributed.run` with N ranks of this same
    command line as a CHILD process and returns its exit code.  This process never touches a GPU (it does not even import
    torch), so nothing that has initialised one is replaced or forked; the ranks' only line on stdout is rank 0's result."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840, help="N = 1 headline frame")
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--kernel", type=int, default=0, help="kernel variant (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 counter passes")
    ap.add_argument("--headline-only", action="store_true", help="skip the secondary blocks (profiling runs)")
    ap.add_argument("--pose-batches", type=int, default=2, help="N > 1: rm_blit_poses_device calls per rank and step")
    ap.add_argument("--root-blocks-per-cu", type=int, default=0, help="N > 1: cap on resident render blocks per CU on rank 0 (0 = none; "
                    "measured on one GPU: a cap does not make room for the expansion kernels, it only slows the rendering)")
    ap.add_argument("--full-expand", action="store_true", help="N > 1: rank 0 rewrites every dense frame whole at every delivery "
                    "(rm_expand_frames_device) instead of updating the tiles that changed (rm_update_frames_device)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="N > 1: gloo = developer rehearsal of the multi-rank code "
                    "path with every exchange through the host (several ranks may then share one GPU: set LOCAL_RANK=0 for all)")
    ap.add_argument("--verify", action="store_true", help="N > 1: rank 0 re-renders one delivered pose per rank and compares the bytes")
    ap.add_argument("--no-one-gpu-reference", action="store_true", help="N > 1: skip rank 0's untimed rendering of all poses alone "
                    "(one_gpu_same_workload / speedup in the result line)")
    ap.add_argument("--force-multi", action="store_true", help="developer check: run the N > 1 code path with a single rank")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         "or without WORLD_SIZE in the environment (bench.py then starts its own ranks)")
    if world == 1 and not args.force_multi:
        run_single(args)
    else:
        run_multi(args)


if __name__ == "__main__":
    main()
