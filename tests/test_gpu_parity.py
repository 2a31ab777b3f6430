"""Parity of the HIP path, called through the C ABI, against the CPU oracle on
identical uniforms, against the committed golden hashes at full sizes, and through
size-independent properties (band/pose decomposition, determinism, counters)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import cases
from helpers import assert_frame_equal, sha

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "golden.json")) as f:
    GOLD = json.load(f)["cases"]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from raymarching_amd import native
    assert os.path.exists(native.LIB_PATH), "librmhip.so must be built before the GPU tests"
    return torch


KERNELS = {"tile": 1, "queue": 2, "stream": 3, "coop": 5}


def gpu_run(case, torch, host_path=False, kernel=0, frames=1, floats=None):
    """binds the case and renders it through the C ABI (device pointers unless host_path);
    `frames` > 1 re-renders into the same target (exercises the frame-to-frame tile history)"""
    from raymarching_amd import Graphics, Material, RenderTexture
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    with Material() as mat:
        mat.set_kernel(kernel)
        src, sd = case.bind(mat)
        for k, v in (floats or {}).items():
            mat.SetFloat(k, v)
        if case.shadow:
            _bind_gpu_shadow_map(case, torch, mat)
        for _ in range(frames - 1):
            _render_once(case, torch, mat, src, sd, host_path)
        return _render_once(case, torch, mat, src, sd, host_path)


def _bind_gpu_shadow_map(case, torch, mat):
    """renders the case's shadow map with the library's own depth pass and binds it as Raymarching.cs:91-93 would"""
    from raymarching_amd import Material
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    kind, sw, sh = case.shadow
    depth = torch.zeros((sh, sw), dtype=torch.float32, device="cuda:0")
    with Material() as sm_mat:
        case.bind(sm_mat)
        extra = case.bind_shadow_pass(sm_mat)
        if kind == "persp":
            sm_mat._ck(sm_mat._lib.rm_shadowmap_render_device(sm_mat.handle, _ptr(depth), None, sw, sh, current_stream_handle("cuda:0")))
        else:
            c, f = np.ascontiguousarray(extra[0]), np.ascontiguousarray(extra[1])
            sm_mat._ck(sm_mat._lib.rm_shadowmap_directional_device(sm_mat.handle, _fp(c), _fp(f), _ptr(depth), None, sw, sh,
                                                                   current_stream_handle("cuda:0")))
        torch.cuda.synchronize()
    mat.SetTexture("_ShadowMap", depth)
    case.bind_shadow_lookup(mat)


def _render_once(case, torch, mat, src, sd, host_path):
    from raymarching_amd import Graphics, RenderTexture
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    if True:
        dev = None if host_path else "cuda:0"
        rt = RenderTexture(case.w, case.h, device=dev)
        lib, h = mat._lib, mat.handle
        if case.kind == "frag":
            if host_path:
                s = RenderTexture(case.w, case.h) if src is not None else None
                if s is not None:
                    s.color[...] = src
                Graphics.Blit(s, rt, mat, scene_depth=sd)
            else:
                s = None
                if src is not None:
                    s = RenderTexture(case.w, case.h, device=dev)
                    s.color.copy_(torch.from_numpy(src))
                d = torch.from_numpy(sd).to(dev) if sd is not None else None
                Graphics.Blit(s, rt, mat, scene_depth=d)
        elif case.kind == "shadow_persp":
            if host_path:
                mat._ck(lib.rm_shadowmap_render(h, _ptr(rt.depth), case.w, case.h))
            else:
                mat._ck(lib.rm_shadowmap_render_device(h, _ptr(rt.depth), _ptr(rt.hit), case.w, case.h,
                                                       current_stream_handle(dev)))
        else:
            c, f = np.ascontiguousarray(case.corners), np.ascontiguousarray(case.forward)
            if host_path:
                mat._ck(lib.rm_shadowmap_directional(h, _fp(c), _fp(f), _ptr(rt.depth), case.w, case.h))
            else:
                mat._ck(lib.rm_shadowmap_directional_device(h, _fp(c), _fp(f), _ptr(rt.depth), _ptr(rt.hit), case.w,
                                                            case.h, current_stream_handle(dev)))
        st = mat.stats()
        if host_path:
            return {"rgba": rt.color, "depth": rt.depth, "hit": rt.hit, "stats": st}
        torch.cuda.synchronize()
        return {"rgba": rt.color.cpu().numpy(), "depth": rt.depth.cpu().numpy(), "hit": rt.hit.cpu().numpy(), "stats": st}


@pytest.mark.parametrize("kernel", ["tile", "queue", "stream"])
@pytest.mark.parametrize("name", [c.name for c in cases.CASES if not c.heavy])
def test_gpu_equals_oracle(oracle, torch_cuda, name, kernel):
    case = cases.BY_NAME[name]
    ref = cases.oracle_run(case, oracle)
    got = gpu_run(case, torch_cuda, kernel=KERNELS[kernel])
    if case.kind == "frag":
        assert_frame_equal(got, ref, name)
        assert got["stats"]["sum_steps_s"] == ref["stats"]["sum_steps_s"]
    else:
        assert np.array_equal(got["hit"], ref["hit"])
        assert np.array_equal(got["depth"].view(np.uint32), ref["depth"].view(np.uint32))
    assert got["stats"]["n_hit"] == int(ref["hit"].sum()) == int(got["hit"].sum())
    assert got["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]
    assert got["stats"]["n_pixels"] == case.w * case.h


COOP_CASES = [c.name for c in cases.CASES if not c.heavy and c.prims is not None and len(c.prims) >= 3]


@pytest.mark.parametrize("kernel", ["coop", "auto"])
@pytest.mark.parametrize("name", COOP_CASES + ["default_640x360"])
def test_cooperative_table_kernel_equals_oracle(oracle, torch_cuda, name, kernel):
    """tables: one block per 8x8 tile, the four waves share the tile's rays and split the table in aligned power-of-two
    chunks (three or four of them, the last possibly partial; one primitive per wave for the smallest tables), which are
    subtrees of the pairwise tree (smax nodes included); the frames and counters are the oracle's bit for bit.  It is the
    automatic choice for tables; the 4-slot scene takes the kernels it took before, also when this one is asked for"""
    case = cases.BY_NAME[name]
    ref = cases.oracle_run(case, oracle)
    got = gpu_run(case, torch_cuda, kernel=KERNELS.get(kernel, 0))
    if case.kind == "frag":
        assert_frame_equal(got, ref, name)
        assert got["stats"]["sum_steps_s"] == ref["stats"]["sum_steps_s"]
    else:
        assert np.array_equal(got["hit"], ref["hit"])
        assert np.array_equal(got["depth"].view(np.uint32), ref["depth"].view(np.uint32))
    assert got["stats"]["n_hit"] == int(ref["hit"].sum()) == int(got["hit"].sum())
    assert got["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]
    assert got["stats"]["n_pixels"] == case.w * case.h


@pytest.mark.parametrize("name", [c.name for c in cases.SKIP_CASES] + ["stress256_320x180", "stress100_160x90"])
def test_exact_culling_in_table_marches_equals_oracle(oracle, torch_cuda, name):
    """tables of 33 to 512 primitives: the two marches skip the left subtrees of the smin tree whose value is
    provably at least k above their right sibling's (smin then returns the right operand bit for bit); the frames and
    counters are the oracle's, and the same with the full table walk (_RmTableSkip 0)"""
    case = cases.BY_NAME[name]
    ref = cases.oracle_run(case, oracle)
    got = gpu_run(case, torch_cuda)
    off = gpu_run(case, torch_cuda, floats={"_RmTableSkip": 0.0})
    for g in (got, off):
        if case.kind == "frag":
            assert_frame_equal(g, ref, name)
            assert g["stats"]["sum_steps_s"] == ref["stats"]["sum_steps_s"]
        else:
            assert np.array_equal(g["hit"], ref["hit"])
            assert np.array_equal(g["depth"].view(np.uint32), ref["depth"].view(np.uint32))
        assert g["stats"]["n_hit"] == int(ref["hit"].sum()) == int(g["hit"].sum())
        assert g["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]


@pytest.mark.parametrize("name", [c.name for c in cases.PAIR_SKIP_CASES])
def test_pair_skip_in_the_shadow_march_equals_oracle(oracle, torch_cuda, name):
    """4-slot scenes without rotation: the soft-shadow march leaves out the (sphere, box) pair while its lower bound stays .5
    above the right pair's value; frames and counters are the oracle's on every exact kernel, and the same with the skip
    switched off (_RmPairSkip 0)"""
    case = cases.BY_NAME[name]
    ref = cases.oracle_run(case, oracle)
    runs = [gpu_run(case, torch_cuda, kernel=k) for k in (0, KERNELS["tile"], KERNELS["queue"], KERNELS["stream"])]
    runs.append(gpu_run(case, torch_cuda, floats={"_RmPairSkip": 0.0}))
    for g in runs:
        assert_frame_equal(g, ref, name)
        assert g["stats"]["sum_steps_s"] == ref["stats"]["sum_steps_s"]
        assert g["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]
        assert g["stats"]["n_hit"] == int(ref["hit"].sum())


@pytest.mark.parametrize("kernel,frames", [("tile", 1), ("queue", 1), ("queue", 4), ("stream", 2)])
@pytest.mark.parametrize("name", ["default_1920x1080", "default_3840x2160"])
def test_gpu_full_size_matches_golden_hashes(torch_cuda, name, kernel, frames):
    """the full benchmark sizes (1080p, 4K) against the oracle's committed hashes; with frames > 1 the
    persistent kernel orders its tile queue by the previous frame's costs, which must not
    change a single output byte"""
    case, g = cases.BY_NAME[name], GOLD[name]
    got = gpu_run(case, torch_cuda, kernel=KERNELS[kernel], frames=frames)
    assert sha(got["hit"]) == g["sha_hit"]
    assert sha(got["depth"]) == g["sha_depth"]
    assert sha(got["rgba"]) == g["sha_rgba"]
    assert got["stats"]["n_hit"] == g["n_hit"]
    assert got["stats"]["sum_steps_p"] == g["sum_steps_p"] and got["stats"]["sum_steps_s"] == g["sum_steps_s"]


@pytest.mark.parametrize("name,kernel", [("shadow_dir_3840x2160", "tile"), ("shadow_dir_3840x2160", "queue"),
                                         ("orbit_k17_7680x4320", "queue"), ("orbit_k17_7680x4320", "tile"),
                                         ("stress256_1920x1080", "tile"), ("stress256_1920x1080", "queue"),
                                         ("stress256_1920x1080", "coop")])
def test_gpu_benchmark_configurations_match_golden_hashes(torch_cuda, name, kernel):
    """the other benchmark configurations at full size against the oracle's committed hashes: the 4K
    directional depth pass, one 8K pose of the orbit, the 256-primitive scene at 1080p with 512 steps"""
    case, g = cases.BY_NAME[name], GOLD[name]
    got = gpu_run(case, torch_cuda, kernel=KERNELS[kernel])
    assert sha(got["hit"]) == g["sha_hit"]
    assert sha(got["depth"]) == g["sha_depth"]
    assert got["stats"]["n_hit"] == g["n_hit"] and got["stats"]["sum_steps_p"] == g["sum_steps_p"]
    if case.kind == "frag":
        assert sha(got["rgba"]) == g["sha_rgba"]
        assert got["stats"]["sum_steps_s"] == g["sum_steps_s"]


@pytest.mark.parametrize("name", ["default_src_depth_333x187", "shadow_persp_100x100", "shadow_dir_256x144", "tiny_17x3"])
def test_host_pointer_path_equals_device_path(torch_cuda, name):
    case = cases.BY_NAME[name]
    a, b = gpu_run(case, torch_cuda), gpu_run(case, torch_cuda, host_path=True)
    assert np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32))
    if case.kind == "frag":
        assert np.array_equal(a["rgba"], b["rgba"]) and np.array_equal(a["hit"], b["hit"])


@pytest.mark.parametrize("w,h,band,parts", [(1920, 1080, 8, 3), (3840, 2160, 8, 8), (333, 187, 5, 4), (64, 7, 8, 2)])
def test_band_shards_reassemble_to_the_full_frame(torch_cuda, w, h, band, parts):
    """a frame rendered as interleaved row bands (the multi-GPU decomposition) equals the one-call frame"""
    torch = torch_cuda
    from raymarching_amd import Material, Raymarching, RenderTexture, scenes
    from raymarching_amd.material import _ptr, current_stream_handle
    with Material() as mat:
        rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms())
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        full = RenderTexture(w, h, device="cuda:0")
        rm.OnRenderImage(None, full)
        full_stats = mat.stats()
        out_c = torch.zeros_like(full.color)
        out_d = torch.zeros_like(full.depth)
        out_h = torch.zeros_like(full.hit)
        tot = {"n_hit": 0, "sum_steps_p": 0, "sum_steps_s": 0, "n_pixels": 0}
        ys = torch.arange(h, device="cuda:0")
        for p in range(parts):
            rows = mat._lib.rm_band_rows(h, band, p, parts)
            if rows == 0:
                continue
            c = torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda:0")
            d = torch.zeros((rows, w), dtype=torch.float32, device="cuda:0")
            m = torch.zeros((rows, w), dtype=torch.uint8, device="cuda:0")
            mat._ck(mat._lib.rm_blit_bands_device(mat.handle, None, None, _ptr(c), _ptr(d), _ptr(m), w, h, band, p,
                                                  parts, current_stream_handle("cuda:0")))
            st = mat.stats()
            for k in tot:
                tot[k] += st[k]
            sel = ((ys // band) % parts) == p
            out_c[sel], out_d[sel], out_h[sel] = c, d, m
        torch.cuda.synchronize()
        assert torch.equal(out_c, full.color) and torch.equal(out_h, full.hit)
        assert torch.equal(out_d.view(torch.int32), full.depth.view(torch.int32))
        assert tot == {k: full_stats[k] for k in tot}


@pytest.mark.parametrize("name,band,parts,kernel", [("shadow_dir_256x144", 8, 2, 0), ("shadow_dir_256x144", 5, 3, 2),
                                                    ("shadow_dir_3840x2160", 8, 2, 0), ("shadow_dir_3840x2160", 8, 8, 0)])
def test_directional_depth_pass_in_bands_reassembles_to_the_full_map(oracle, torch_cuda, name, band, parts, kernel):
    """the ShadowMapDirectional depth pass rendered as interleaved row bands (rm_shadowmap_directional_bands_device, the
    multi-GPU decomposition of that pass) equals the one-call map bit for bit, which equals the oracle's (small case) or
    its committed hashes (4K)"""
    torch = torch_cuda
    from raymarching_amd import Material
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    case = cases.BY_NAME[name]
    w, h = case.w, case.h
    full = gpu_run(case, torch, kernel=kernel)
    if name in GOLD and case.heavy:
        assert sha(full["depth"]) == GOLD[name]["sha_depth"] and sha(full["hit"]) == GOLD[name]["sha_hit"]
    else:
        ref = cases.oracle_run(case, oracle)
        assert np.array_equal(full["depth"].view(np.uint32), ref["depth"].view(np.uint32)) and np.array_equal(full["hit"], ref["hit"])
    with Material() as mat:
        mat.set_kernel(kernel)
        case.bind(mat)
        c, f = np.ascontiguousarray(case.corners), np.ascontiguousarray(case.forward)
        out_d = torch.zeros((h, w), dtype=torch.float32, device="cuda:0")
        out_h = torch.zeros((h, w), dtype=torch.uint8, device="cuda:0")
        ys = torch.arange(h, device="cuda:0")
        tot = {"n_hit": 0, "sum_steps_p": 0, "n_pixels": 0}
        for p in range(parts):
            rows = mat._lib.rm_band_rows(h, band, p, parts)
            d = torch.zeros((rows, w), dtype=torch.float32, device="cuda:0")
            m = torch.zeros((rows, w), dtype=torch.uint8, device="cuda:0")
            mat._ck(mat._lib.rm_shadowmap_directional_bands_device(mat.handle, _fp(c), _fp(f), _ptr(d), _ptr(m), w, h, band, p, parts,
                                                                   current_stream_handle("cuda:0")))
            st = mat.stats()
            for k in tot:
                tot[k] += st[k]
            sel = ((ys // band) % parts) == p
            out_d[sel], out_h[sel] = d, m
        torch.cuda.synchronize()
        assert np.array_equal(out_d.cpu().numpy().view(np.uint32), full["depth"].view(np.uint32))
        assert np.array_equal(out_h.cpu().numpy(), full["hit"])
        assert tot == {k: full["stats"][k] for k in tot}
        lib = mat._lib
        from raymarching_amd import native
        assert lib.rm_shadowmap_directional_bands_device(mat.handle, _fp(c), _fp(f), _ptr(out_d), None, w, h, 0, 0, 1, None) == native.RM_E_ARG
        assert lib.rm_shadowmap_directional_bands_device(mat.handle, _fp(c), _fp(f), _ptr(out_d), None, w, h, 8, 2, 2, None) == native.RM_E_ARG


def test_leaf_statistics_of_the_exact_culling(torch_cuda):
    """rm_get_leaf_stats: the culled marches execute a part of the leaf evaluations a full table walk would; with the
    culling off, or after a launch of another kernel, there is nothing to report"""
    torch = torch_cuda
    from raymarching_amd import Material, native
    case = cases.BY_NAME["stress256_320x180"]
    with Material() as mat:
        src, sd = case.bind(mat)
        _render_once(case, torch, mat, src, sd, False)
        ls = mat.leaf_stats()
        st = mat.stats()
        evals = st["sum_steps_p"] + st["sum_steps_s"]
        assert 0 < ls["visited"] < ls["possible"]
        # every wave evaluation could have visited its 64 leaves; a tile's four waves make at least one evaluation per
        # 64 pixel evaluations, and never more than one per pixel evaluation (plus 7 per hit tile)
        assert ls["possible"] % 64 == 0 and ls["possible"] >= 256 * (evals // 64)
        assert 0.2 < ls["visited"] / ls["possible"] < 0.9
        mat.SetFloat("_RmTableSkip", 0.0)
        _render_once(case, torch, mat, src, sd, False)
        with pytest.raises(native.RmError):
            mat.leaf_stats()


def test_pose_batch_equals_single_blits(oracle, torch_cuda):
    """rm_blit_poses_device over 5 orbit poses == 5 x (SetMatrix x2 + Blit), and pose 3 == the oracle"""
    torch = torch_cuda
    from raymarching_amd import Material, Raymarching, RenderTexture, scenes
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    w, h, ks = 480, 270, [0, 9, 23, 40, 63]
    with Material() as mat:
        singles = []
        fcs, c2ws, pos = [], [], []
        for k in ks:
            rm = Raymarching(scenes.orbit_camera(k, 64, w, h), mat, scenes.default_transforms())
            mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            rt = RenderTexture(w, h, device="cuda:0")
            rm.OnRenderImage(None, rt)
            singles.append(rt)
            fcs.append(rm.GetFrustumCorners().m)
            c2ws.append(rm.cam.cameraToWorldMatrix.m)
            pos.append(rm.cam.transform.position)
        n = len(ks)
        fcs = np.ascontiguousarray(np.stack(fcs), dtype=np.float32)
        c2ws = np.ascontiguousarray(np.stack(c2ws), dtype=np.float32)
        pos = np.ascontiguousarray(np.stack(pos), dtype=np.float32)
        c = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda:0")
        d = torch.zeros((n, h, w), dtype=torch.float32, device="cuda:0")
        m = torch.zeros((n, h, w), dtype=torch.uint8, device="cuda:0")
        mat._ck(mat._lib.rm_blit_poses_device(mat.handle, n, _fp(fcs), _fp(c2ws), _fp(pos), _ptr(c), _ptr(d), _ptr(m), w, h,
                                              current_stream_handle("cuda:0")))
        st = mat.stats()
        torch.cuda.synchronize()
        assert st["n_pixels"] == n * w * h and st["n_hit"] == int(m.sum())
        for i in range(n):
            assert torch.equal(c[i], singles[i].color) and torch.equal(m[i], singles[i].hit)
            assert torch.equal(d[i].view(torch.int32), singles[i].depth.view(torch.int32))
    ref = cases.oracle_run(cases.BY_NAME["orbit_k23_480x270"], oracle)
    assert np.array_equal(c[2].cpu().numpy(), ref["rgba"])


@pytest.mark.parametrize("kernel", ["auto", "tile", "stream"])
def test_table_scene_in_bands_and_pose_batches(torch_cuda, kernel):
    """a 100-primitive table (the kernels whose four waves share a tile, or one wave per tile) rendered as interleaved row
    bands and as a batch of three poses equals the one-call frames of the same context bit for bit, counters included"""
    torch = torch_cuda
    from raymarching_amd import Material, Raymarching, RenderTexture, scenes
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    w, h, band, parts = 203, 117, 5, 3
    dev = "cuda:0"
    with Material() as mat:
        mat.set_kernel(KERNELS.get(kernel, 0))
        for k_, v_ in scenes.STRESS_FLOATS.items():
            mat.SetFloat(k_, v_)
        mat.SetFloat("_MaxSteps", 96.0)
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        mat.SetPrimitives(scenes.stress_primitives(100, seed=0x5EED0064))
        singles, fcs, c2ws, pos, stats = [], [], [], [], []
        for k in (3, 20, 41):
            rm = Raymarching(scenes.orbit_camera(k, 64, w, h), mat, scenes.default_transforms())
            rt = RenderTexture(w, h, device=dev)
            rm.OnRenderImage(None, rt)
            stats.append(mat.stats())
            singles.append(rt)
            fcs.append(rm.GetFrustumCorners().m); c2ws.append(rm.cam.cameraToWorldMatrix.m); pos.append(rm.cam.transform.position)
        assert stats[0]["n_hit"] > 0
        n = len(singles)
        c = torch.zeros((n, h, w, 4), dtype=torch.uint8, device=dev)
        d = torch.zeros((n, h, w), dtype=torch.float32, device=dev)
        m = torch.zeros((n, h, w), dtype=torch.uint8, device=dev)
        arr = [np.ascontiguousarray(np.stack(a), dtype=np.float32) for a in (fcs, c2ws, pos)]
        mat._ck(mat._lib.rm_blit_poses_device(mat.handle, n, _fp(arr[0]), _fp(arr[1]), _fp(arr[2]), _ptr(c), _ptr(d), _ptr(m), w, h,
                                              current_stream_handle(dev)))
        st = mat.stats()
        torch.cuda.synchronize()
        for key in ("n_hit", "sum_steps_p", "sum_steps_s", "n_pixels"):
            assert st[key] == sum(s_[key] for s_ in stats), key
        for i in range(n):
            assert torch.equal(c[i], singles[i].color) and torch.equal(m[i], singles[i].hit)
            assert torch.equal(d[i].view(torch.int32), singles[i].depth.view(torch.int32))
        # the last pose again, as row bands (the camera properties are still that pose's)
        ys = torch.arange(h, device=dev)
        out_c, out_d, out_h = torch.zeros_like(singles[-1].color), torch.zeros_like(singles[-1].depth), torch.zeros_like(singles[-1].hit)
        tot = {"n_hit": 0, "sum_steps_p": 0, "sum_steps_s": 0, "n_pixels": 0}
        for p_ in range(parts):
            rows = mat._lib.rm_band_rows(h, band, p_, parts)
            cb = torch.zeros((rows, w, 4), dtype=torch.uint8, device=dev)
            db = torch.zeros((rows, w), dtype=torch.float32, device=dev)
            mb = torch.zeros((rows, w), dtype=torch.uint8, device=dev)
            mat._ck(mat._lib.rm_blit_bands_device(mat.handle, None, None, _ptr(cb), _ptr(db), _ptr(mb), w, h, band, p_, parts,
                                                  current_stream_handle(dev)))
            sb = mat.stats()
            for key in tot:
                tot[key] += sb[key]
            sel = ((ys // band) % parts) == p_
            out_c[sel], out_d[sel], out_h[sel] = cb, db, mb
        torch.cuda.synchronize()
        assert torch.equal(out_c, singles[-1].color) and torch.equal(out_h, singles[-1].hit)
        assert torch.equal(out_d.view(torch.int32), singles[-1].depth.view(torch.int32))
        assert tot == {key: stats[-1][key] for key in tot}


def test_pose_batch_at_full_size(torch_cuda):
    """three 7680x4320 orbit poses in one rm_blit_poses_device call; the middle one is pose 17, whose
    hashes are committed"""
    torch = torch_cuda
    from raymarching_amd import Material, Raymarching, scenes
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    w, h, ks = 7680, 4320, [9, 17, 25]
    g = GOLD["orbit_k17_7680x4320"]
    with Material() as mat:
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        cams = [Raymarching(scenes.orbit_camera(k, 64, w, h), mat, scenes.default_transforms()) for k in ks]
        cams[0].bind()
        fcs = np.ascontiguousarray(np.stack([r.GetFrustumCorners().m for r in cams]), dtype=np.float32)
        c2ws = np.ascontiguousarray(np.stack([r.cam.cameraToWorldMatrix.m for r in cams]), dtype=np.float32)
        pos = np.ascontiguousarray(np.stack([r.cam.transform.position for r in cams]), dtype=np.float32)
        n = len(ks)
        color = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda:0")
        depth = torch.zeros((n, h, w), dtype=torch.float32, device="cuda:0")
        hit = torch.zeros((n, h, w), dtype=torch.uint8, device="cuda:0")
        for _ in range(2):                       # the second frame runs with the first one's tile costs
            mat._ck(mat._lib.rm_blit_poses_device(mat.handle, n, _fp(fcs), _fp(c2ws), _fp(pos), _ptr(color), _ptr(depth),
                                                  _ptr(hit), w, h, current_stream_handle("cuda:0")))
        torch.cuda.synchronize()
        st = mat.stats()
        assert st["n_pixels"] == n * w * h
        assert sha(color[1].cpu().numpy()) == g["sha_rgba"]
        assert sha(depth[1].cpu().numpy()) == g["sha_depth"]
        assert sha(hit[1].cpu().numpy()) == g["sha_hit"]
        assert int(hit[1].sum().item()) == g["n_hit"]


@pytest.mark.parametrize("name", ["stress256_320x180", "orbit_k23_480x270", "shadow_dir_256x144"])
def test_tile_history_does_not_change_output(oracle, torch_cuda, name):
    """3rd consecutive frame (heavy tiles first) == 1st frame == oracle, also with a very low threshold"""
    from raymarching_amd import Material
    case = cases.BY_NAME[name]
    ref = cases.oracle_run(case, oracle)
    for thr in (40.0, 3.0):
        with Material() as mat:
            mat.set_kernel(KERNELS["queue"])
            src, sd = case.bind(mat)
            mat.SetFloat("_RmHeavyTileThreshold", thr)
            for _ in range(3):
                got = _render_once(case, torch_cuda, mat, src, sd, False)
                assert np.array_equal(got["depth"].view(np.uint32), ref["depth"].view(np.uint32))
                assert np.array_equal(got["hit"], ref["hit"])
                if case.kind == "frag":
                    assert np.array_equal(got["rgba"], ref["rgba"])
                assert got["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]


@pytest.mark.parametrize("threshold", [0.0, 1.0, 5.0, 200.0])
def test_predicted_tile_order_does_not_change_output(oracle, torch_cuda, threshold):
    """a frame without a previous one of its shape, and a frame after a camera cut, have their queue ordered by a prediction
    pass (one ray per tile): whatever it lists -- nothing (threshold 200), nearly every tile (1), more tiles than the list
    holds -- every frame is the oracle's; the same sequence with the pass switched off gives the same bytes"""
    from raymarching_amd import Material, Raymarching, scenes
    from raymarching_amd.material import _ptr, current_stream_handle
    torch = torch_cuda
    w, h = 1536, 864                      # 20736 tiles: more than the list's 8192 entries when (almost) all are listed
    poses = [0, 0, 23, 23, 24, 5]         # first frame; history; cut; history; small move; cut
    refs = {}
    for k in set(poses):
        case = cases.Case(f"orbit_k{k}_{w}x{h}", w, h, setup=cases._orbit(k))
        refs[k] = cases.oracle_run(case, oracle)
    stream = current_stream_handle("cuda:0")
    outs = {}
    for predict in (1.0, 0.0):
        with Material() as mat:
            mat.set_kernel(KERNELS["queue"])
            mat.SetFloat("_RmPredictTiles", predict)
            if threshold > 0.0:
                mat.SetFloat("_RmPredictThreshold", threshold)
            c = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
            d = torch.zeros((h, w), dtype=torch.float32, device="cuda:0")
            hm = torch.zeros((h, w), dtype=torch.uint8, device="cuda:0")
            for i, k in enumerate(poses):
                cases.Case(f"orbit_k{k}_{w}x{h}", w, h, setup=cases._orbit(k)).bind(mat)
                mat.SetFloat("_RmPredictTiles", predict)
                if threshold > 0.0:
                    mat.SetFloat("_RmPredictThreshold", threshold)
                mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, stream))
                torch.cuda.synchronize()
                got = {"rgba": c.cpu().numpy(), "depth": d.cpu().numpy(), "hit": hm.cpu().numpy()}
                assert_frame_equal(got, refs[k], f"predict {predict} frame {i} pose {k}")
                st = mat.stats()
                assert st["n_hit"] == refs[k]["stats"]["n_hit"] and st["sum_steps_p"] == refs[k]["stats"]["sum_steps_p"]
                assert st["sum_steps_s"] == refs[k]["stats"]["sum_steps_s"]
                outs[(predict, i)] = sha(got["rgba"])
    assert all(outs[(1.0, i)] == outs[(0.0, i)] for i in range(len(poses)))


def test_which_frames_get_a_predicted_order(torch_cuda):
    """the host's decision when the previous frame's tile costs are of no use: a context's first frame and a resize; a camera
    cut; a primitive moved far; the light turned -- but not the same pose again, a small camera step or a primitive nudged"""
    from raymarching_amd import Material, Raymarching, scenes
    from raymarching_amd.material import _ptr, current_stream_handle
    torch = torch_cuda
    w, h = 640, 360
    stream = current_stream_handle("cuda:0")
    c = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
    d = torch.zeros((h, w), dtype=torch.float32, device="cuda:0")
    hm = torch.zeros((h, w), dtype=torch.uint8, device="cuda:0")
    with Material() as mat:
        mat.set_kernel(KERNELS["queue"])

        def frame(k=0, hh=h, move=None, light=scenes.WORLD_SPACE_LIGHT_POS0):
            ts = scenes.default_transforms()
            if move is not None:
                ts[0].position = (np.asarray(ts[0].position, dtype=np.float32) + np.asarray(move, dtype=np.float32)).astype(np.float32)
            Raymarching(scenes.orbit_camera(k, 720, w, hh), mat, ts).bind()          # 720 poses: half a degree per pose
            mat.SetVector("_WorldSpaceLightPos0", np.asarray(light, dtype=np.float32))
            mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, hh, stream))
            torch.cuda.synchronize()
            return int(mat.GetFloat("_RmPredictedFrames"))        # read-only diagnostic the library keeps in the property bag

        assert frame(0) == 1                                   # first frame of the context
        assert frame(0) == 1                                   # the same pose again: history
        assert frame(1) == 1                                   # half a degree: history
        assert frame(2, move=(0.01, 0.0, 0.0)) == 1            # a primitive nudged: history
        assert frame(90) == 2                                  # 44 degrees further: cut
        assert frame(90) == 2
        assert frame(90, move=(3.0, 0.0, 0.0)) == 3            # a primitive carried across the picture
        assert frame(90, move=(3.0, 0.0, 0.0)) == 3
        lt = np.asarray(scenes.WORLD_SPACE_LIGHT_POS0, dtype=np.float32).copy(); lt[0] += 0.5
        assert frame(90, move=(3.0, 0.0, 0.0), light=lt) == 4  # the light turned
        assert frame(90, move=(3.0, 0.0, 0.0), light=lt) == 4
        assert frame(90, hh=h - 8, move=(3.0, 0.0, 0.0), light=lt) == 5      # another frame size
        mat.SetFloat("_RmPredictTiles", 0.0)
        assert frame(400, hh=h) == 5                           # switched off: no prediction whatever happens


@pytest.mark.parametrize("name", ["default_src_depth_640x360", "shadowmap_lookup_dir_320x180", "shadowmap_lookup_persp_320x180", "tiny_17x3", "wide_1025x2", "transformed_400x225",
                                  "pair_skip_scaled_400x225", "edge_nan_camera_32x18"])
def test_first_frame_prediction_on_other_launch_shapes(oracle, torch_cuda, name):
    """the prediction pass ahead of a context's first frame: with source image and depth texture, with the shadow term looked up,
    on frames smaller than a tile row, on the rotated / scaled variants, with a camera that is not a number"""
    if name not in cases.BY_NAME:
        pytest.skip(f"no case {name}")
    case = cases.BY_NAME[name]
    ref = cases.oracle_run(case, oracle)
    for predict in (1.0, 0.0):
        got = gpu_run(case, torch_cuda, kernel=KERNELS["queue"], floats={"_RmPredictTiles": predict})
        assert_frame_equal(got, ref, f"{name} predict {predict}")
        assert got["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]


def test_more_heavy_tiles_than_waves(oracle, torch_cuda):
    """a frame with more tiles than resident waves and a threshold that puts (almost) every tile on the
    heavy list: the part of the list beyond one entry per wave must still be rendered, exactly once"""
    from raymarching_amd import Material, Raymarching, scenes
    w, h = 1536, 864                      # 192 x 108 = 20736 tiles > 8192 list entries > 4096 waves
    case = cases.Case("default_1536x864", w, h)
    ref = cases.oracle_run(case, oracle)
    with Material() as mat:
        mat.set_kernel(KERNELS["queue"])
        src, sd = case.bind(mat)
        mat.SetFloat("_RmHeavyTileThreshold", 3.0)
        for frame in range(3):
            got = _render_once(case, torch_cuda, mat, src, sd, False)
            assert_frame_equal(got, ref, f"frame {frame}")
            assert got["stats"]["n_hit"] == ref["stats"]["n_hit"] and got["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]


def test_repeat_runs_are_identical_and_counters_reset(torch_cuda):
    case = cases.BY_NAME["default_640x360"]
    a, b = gpu_run(case, torch_cuda), gpu_run(case, torch_cuda)
    assert sha(a["rgba"]) == sha(b["rgba"]) and sha(a["depth"]) == sha(b["depth"])
    assert a["stats"] == b["stats"]
    assert a["stats"]["n_hit"] == int(a["hit"].sum())


def test_changing_a_uniform_changes_the_frame(torch_cuda):
    """guards against a cached/stale parameter block"""
    torch = torch_cuda
    from raymarching_amd import Material, Raymarching, RenderTexture, scenes
    with Material() as mat:
        rm = Raymarching(scenes.default_camera(320, 180), mat, scenes.default_transforms())
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        a, b = RenderTexture(320, 180, device="cuda:0"), RenderTexture(320, 180, device="cuda:0")
        rm.OnRenderImage(None, a)
        rm.transforms[0].position = np.array([-0.584, 1.5, 0.602], dtype=np.float32)
        rm.OnRenderImage(None, b)
        torch.cuda.synchronize()
        assert not torch.equal(a.color, b.color)


@pytest.mark.parametrize("kernel", ["tile", "queue"])
@pytest.mark.parametrize("seed", range(72))
def test_random_scenes(oracle, torch_cuda, seed, kernel):
    """seeded random scenes (all primitive kinds, op bits, 4-slot and fractal scenes, random cameras and
    march uniforms): bit-exact against the oracle"""
    case = cases.random_case(seed)
    ref = cases.oracle_run(case, oracle)
    got = gpu_run(case, torch_cuda, kernel=KERNELS[kernel])
    assert_frame_equal(got, ref, case.name)
    assert got["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]
    assert got["stats"]["sum_steps_s"] == ref["stats"]["sum_steps_s"]


@pytest.mark.parametrize("world_scale", [1.0, 0.05, 12.0, 60.0])
@pytest.mark.parametrize("seed", range(40))
def test_random_plain_tables_with_exact_culling(oracle, torch_cuda, seed, world_scale):
    """seeded random smooth unions of 33..512 primitives, the tables whose marches skip subtrees: bit-exact against the
    oracle (automatic kernel choice).  The world scales put the scene's lengths at a twentieth (everything blends), at
    twelve times (coordinates in the hundreds: the largest rounding errors the margin has to cover) and at sixty times the
    blend radius's scale (coordinates beyond the range in which anything is skipped)"""
    if world_scale != 1.0 and seed >= 16:
        pytest.skip("16 seeds per extra scale")
    case = cases.random_table_case(seed, world_scale)
    ref = cases.oracle_run(case, oracle)
    got = gpu_run(case, torch_cuda)
    assert_frame_equal(got, ref, case.name)
    assert got["stats"]["n_hit"] == ref["stats"]["n_hit"]
    assert got["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]
    assert got["stats"]["sum_steps_s"] == ref["stats"]["sum_steps_s"]


@pytest.mark.parametrize("world_scale", [1.0, 8.0])
@pytest.mark.parametrize("seed", range(32))
def test_random_tables_with_op_bits_and_exact_culling(oracle, torch_cuda, seed, world_scale):
    """the same with smax nodes and negated leaves: an smax node returns its right operand exactly once the left one is
    .5 below it, so there the bounds change sides"""
    case = cases.random_table_case(seed, world_scale, csg=True)
    ref = cases.oracle_run(case, oracle)
    got = gpu_run(case, torch_cuda)
    off = gpu_run(case, torch_cuda, floats={"_RmTableSkip": 0.0})
    for g in (got, off):
        assert_frame_equal(g, ref, case.name)
        assert g["stats"]["n_hit"] == ref["stats"]["n_hit"]
        assert g["stats"]["sum_steps_p"] == ref["stats"]["sum_steps_p"]
        assert g["stats"]["sum_steps_s"] == ref["stats"]["sum_steps_s"]


def test_rgb_pack_round_trip(torch_cuda):
    """frame transport: RGBA8 -> RGB8 -> RGBA8 (alpha 255) through the C ABI"""
    torch = torch_cuda
    from raymarching_amd import Material
    g = torch.Generator(device="cpu").manual_seed(7)
    for n in (4, 64, 1920 * 1080, 333 * 188):
        rgba = torch.randint(0, 256, (n, 4), dtype=torch.uint8, generator=g).cuda()
        rgb = torch.zeros((n, 3), dtype=torch.uint8, device="cuda")
        back = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
        with Material() as mat:
            mat.pack_rgb(rgba, rgb)
            mat.unpack_rgb(rgb, back)
            torch.cuda.synchronize()
            assert torch.equal(rgb, rgba[:, :3])
            assert torch.equal(back[:, :3], rgba[:, :3]) and bool((back[:, 3] == 255).all())
            with pytest.raises(RuntimeError):
                mat.pack_rgb(rgba[:3], rgb[:3])            # not a multiple of 4 pixels


def _frames_of(case_names, oracle):
    return {n: cases.oracle_run(cases.BY_NAME[n], oracle) for n in case_names}


def test_one_context_across_sizes_kernels_and_pose_batches(oracle, torch_cuda):
    """One context reused the way a player does: frames at one size, a launch that keeps no tile history (the
    lane-compaction kernel, a pose batch with more tiles than the history bitmap holds), then frames at another size
    and another camera.  The persistent kernel's frame-to-frame state (claim counters, heavy-tile lists and bitmaps)
    must never leak from one shape into the next: every frame equals the oracle's bit for bit."""
    torch = torch_cuda
    from raymarching_amd import Material, Raymarching, RenderTexture, scenes
    from raymarching_amd.material import _fp, _ptr, current_stream_handle
    dev = "cuda:0"
    ref = _frames_of(["default_640x360", "orbit_k5_480x270", "orbit_k23_480x270", "orbit_k40_480x270"], oracle)

    def frame(mat, w, h, cam):
        rm = Raymarching(cam, mat, scenes.default_transforms())
        rt = RenderTexture(w, h, device=dev)
        rm.OnRenderImage(None, rt)
        st = mat.stats()
        torch.cuda.synchronize()
        return {"rgba": rt.color.cpu().numpy(), "depth": rt.depth.cpu().numpy(), "hit": rt.hit.cpu().numpy(), "stats": st}

    def big_pose_batch(mat):
        # 12 poses of 1920x1080 = 388 800 tiles: beyond the history bitmap, so this launch keeps no history
        w, h, n = 1920, 1080, 12
        cams = [Raymarching(scenes.orbit_camera(5 * k, 64, w, h), mat, scenes.default_transforms()) for k in range(n)]
        fcs = np.ascontiguousarray(np.stack([r.GetFrustumCorners().m for r in cams]), dtype=np.float32)
        c2ws = np.ascontiguousarray(np.stack([r.cam.cameraToWorldMatrix.m for r in cams]), dtype=np.float32)
        pos = np.ascontiguousarray(np.stack([r.cam.transform.position for r in cams]), dtype=np.float32)
        c = torch.zeros((n, h, w, 4), dtype=torch.uint8, device=dev)
        d = torch.zeros((n, h, w), dtype=torch.float32, device=dev)
        m = torch.zeros((n, h, w), dtype=torch.uint8, device=dev)
        mat._ck(mat._lib.rm_blit_poses_device(mat.handle, n, _fp(fcs), _fp(c2ws), _fp(pos), _ptr(c), _ptr(d), _ptr(m), w, h,
                                              current_stream_handle(dev)))
        torch.cuda.synchronize()

    for breaker in ("stream", "poses"):
        with Material() as mat:
            mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            mat.SetFloat("_RmHeavyTileThreshold", 6.0)          # many heavy tiles: stale bits would hide many tiles
            mat.set_kernel(KERNELS["queue"])
            for i in range(3):
                assert_frame_equal(frame(mat, 640, 360, scenes.default_camera(640, 360)), ref["default_640x360"], f"{breaker} A{i}")
            if breaker == "stream":
                mat.set_kernel(KERNELS["stream"])
                assert_frame_equal(frame(mat, 640, 360, scenes.default_camera(640, 360)), ref["default_640x360"], "stream frame")
                mat.set_kernel(KERNELS["queue"])
            else:
                big_pose_batch(mat)
            # another size, moving camera: every frame has the previous pose's heavy tiles as its history
            for i, k in enumerate((5, 23, 40, 23, 5)):
                got = frame(mat, 480, 270, scenes.orbit_camera(k, 64, 480, 270))
                assert_frame_equal(got, ref[f"orbit_k{k}_480x270"], f"{breaker} B{i} pose {k}")
                assert got["stats"]["sum_steps_p"] == ref[f"orbit_k{k}_480x270"]["stats"]["sum_steps_p"]
            # and back to the first size
            for i in range(2):
                assert_frame_equal(frame(mat, 640, 360, scenes.default_camera(640, 360)), ref["default_640x360"], f"{breaker} C{i}")


def test_consecutive_launches_on_different_streams(oracle, torch_cuda):
    """a context driven from two caller streams in turn: the library orders each launch behind the previous one
    (shared claim counters and counter slots), so every frame and every rm_get_stats is right"""
    torch = torch_cuda
    from raymarching_amd import Material, Raymarching, RenderTexture, scenes
    from raymarching_amd.material import _ptr
    dev = "cuda:0"
    case = cases.BY_NAME["default_640x360"]
    ref = cases.oracle_run(case, oracle)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    with Material() as mat:
        case.bind(mat)
        mat.set_kernel(KERNELS["queue"])
        rts = [RenderTexture(case.w, case.h, device=dev) for _ in range(6)]
        torch.cuda.synchronize()
        for i, rt in enumerate(rts):
            s = streams[i & 1]
            mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), case.w, case.h,
                                            s.cuda_stream))
        st = mat.stats()
        torch.cuda.synchronize()
        assert st["sum_steps_p"] == ref["stats"]["sum_steps_p"] and st["n_hit"] == ref["stats"]["n_hit"]
        for i, rt in enumerate(rts):
            got = {"rgba": rt.color.cpu().numpy(), "depth": rt.depth.cpu().numpy(), "hit": rt.hit.cpu().numpy()}
            assert_frame_equal(got, ref, f"frame {i}")


@pytest.mark.parametrize("name", ["default_640x360", "default_src_depth_333x187", "stress13_200x112", "menger_closeup_320x180",
                                  "fx_bounce_320x180", "edge_zero_steps_64x36", "edge_big_steps_2000_cap_96x54"])
@pytest.mark.parametrize("host_path", [False, True])
def test_step_count_plane_equals_oracle(oracle, torch_cuda, name, host_path):
    """rm_blit_steps / rm_blit_steps_device: per-pixel evaluation counts of the primary and the shadow march
    (ray.steps, cginc:37,387) equal the oracle's, the other planes are unchanged, the frame sums are rm_get_stats'"""
    torch = torch_cuda
    from raymarching_amd import Material
    from raymarching_amd.material import _ptr, current_stream_handle
    case = cases.BY_NAME[name]
    ref = cases.oracle_run(case, oracle)
    w, h = case.w, case.h
    with Material() as mat:
        src, sd = case.bind(mat)
        if host_path:
            rgba = np.zeros((h, w, 4), np.uint8); depth = np.zeros((h, w), np.float32); hit = np.zeros((h, w), np.uint8)
            steps = np.zeros((h, w, 2), np.uint16)
            mat._ck(mat._lib.rm_blit_steps(mat.handle, _ptr(src), _ptr(sd), _ptr(rgba), _ptr(depth), _ptr(hit), _ptr(steps), w, h))
            st = mat.stats()
        else:
            dev = "cuda:0"
            t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
            dsrc, dsd = t(src), t(sd)
            rgba_t = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
            depth_t = torch.zeros((h, w), dtype=torch.float32, device=dev)
            hit_t = torch.zeros((h, w), dtype=torch.uint8, device=dev)
            steps_t = torch.zeros((h, w, 2), dtype=torch.int16, device=dev)
            mat._ck(mat._lib.rm_blit_steps_device(mat.handle, _ptr(dsrc), _ptr(dsd), _ptr(rgba_t), _ptr(depth_t), _ptr(hit_t),
                                                  _ptr(steps_t), w, h, current_stream_handle(dev)))
            st = mat.stats()
            torch.cuda.synchronize()
            rgba, depth, hit = rgba_t.cpu().numpy(), depth_t.cpu().numpy(), hit_t.cpu().numpy()
            steps = steps_t.cpu().numpy().view(np.uint16)
    assert_frame_equal({"rgba": rgba, "depth": depth, "hit": hit}, ref, name)
    assert np.array_equal(steps[..., 0], ref["steps_p"]), "primary step counts differ"
    assert np.array_equal(steps[..., 1], ref["steps_s"]), "shadow step counts differ"
    assert int(steps[..., 0].astype(np.int64).sum()) == st["sum_steps_p"] == ref["stats"]["sum_steps_p"]
    assert int(steps[..., 1].astype(np.int64).sum()) == st["sum_steps_s"] == ref["stats"]["sum_steps_s"]


@pytest.mark.parametrize("name,with_src", [("default_640x360", False), ("default_src_depth_333x187", True),
                                           ("edge_camera_inside_box_128x72", False), ("tiny_17x3", False),
                                           ("orbit_k23_480x270", False)])
def test_compact_transport_round_trip(torch_cuda, name, with_src):
    """frame transport: compact (hit tiles only) -> expand over the background == the frame the blit wrote, byte for
    byte; the records equal the host restatement's; covers a source image, a frame whose every pixel hits (camera
    inside the box, where most tiles hit), ragged sizes and a batch of frames"""
    torch = torch_cuda
    from helpers import RECORD_WORDS, compact_frames_numpy, expand_frames_numpy
    from raymarching_amd import Material
    from raymarching_amd.material import _ptr, current_stream_handle
    case = cases.BY_NAME[name]
    dev = "cuda:0"
    w, h, n = case.w, case.h, 3
    with Material() as mat:
        src, sd = case.bind(mat)
        dsrc = torch.from_numpy(src).to(dev) if (src is not None and with_src) else None
        frames = torch.zeros((n, h, w, 4), dtype=torch.uint8, device=dev)
        hit = torch.zeros((n, h, w), dtype=torch.uint8, device=dev)
        depth = torch.zeros((h, w), dtype=torch.float32, device=dev)
        for f in range(n):
            if f == 1:
                mat.SetFloat("_MaxDist", 9.0 if name.startswith("default") else 4.0)      # a different frame in the middle
            if f == 2:
                mat.SetFloat("_MaxDist", 50.0)
            mat._ck(mat._lib.rm_blit_device(mat.handle, _ptr(dsrc), None, _ptr(frames[f]), _ptr(depth), _ptr(hit[f]), w, h,
                                            current_stream_handle(dev)))
        tiles = ((w + 7) // 8) * ((h + 7) // 8) * n
        records = torch.full((tiles, RECORD_WORDS), -1, dtype=torch.int32, device=dev)
        count = torch.full((1,), 77, dtype=torch.int32, device=dev)
        mat.compact_frames(frames, hit, records, count)
        k = int(count.item())
        want = compact_frames_numpy(frames.cpu().numpy(), hit.cpu().numpy())
        assert k == want.shape[0]
        got = records[:k].cpu().numpy().view(np.uint32)
        order = np.argsort(got[:, 0], kind="stable")
        assert np.array_equal(got[order], want.view(np.uint32)), "records differ from the host restatement"
        back = torch.full_like(frames, 3)
        mat.expand_frames(records, k, back, src=dsrc)
        torch.cuda.synchronize()
        assert torch.equal(back, frames), "compact -> expand is not the identity"
        if with_src:
            bg = src
        else:
            c = np.clip(mat.GetVector("_BackgroundColor")[:3].astype(np.float32), 0, 1)
            bg = list(np.floor(c * np.float32(255) + np.float32(0.5)).astype(np.uint8)) + [255]
        assert np.array_equal(expand_frames_numpy(want, n, h, w, bg), frames.cpu().numpy())


@pytest.mark.parametrize("w,h", [(64, 40), (77, 13)])
def test_compact_transport_all_hit_and_no_hit(torch_cuda, w, h):
    """the two extremes on arbitrary pixels: every tile holds a hit (one record per tile, the expansion restores every
    RGB byte) and no tile does (no record, the expansion is the background alone)"""
    torch = torch_cuda
    from helpers import RECORD_WORDS
    from raymarching_amd import Material
    dev, n = "cuda:0", 2
    g = torch.Generator(device="cpu").manual_seed(11)
    frames = torch.randint(0, 256, (n, h, w, 4), dtype=torch.uint8, generator=g).to(dev)
    frames[..., 3] = 255
    tiles = ((w + 7) // 8) * ((h + 7) // 8) * n
    records = torch.zeros((tiles, RECORD_WORDS), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    with Material() as mat:
        hit = torch.ones((n, h, w), dtype=torch.uint8, device=dev)
        mat.compact_frames(frames, hit, records, count)
        assert int(count.item()) == tiles
        assert sorted(records[:, 0].cpu().tolist()) == list(range(tiles))
        back = torch.zeros_like(frames)
        mat.expand_frames(records, tiles, back)
        torch.cuda.synchronize()
        assert torch.equal(back, frames)
        hit.zero_()
        hit[1, h - 1, w - 1] = 1                                       # one pixel in the last, ragged tile
        mat.compact_frames(frames, hit, records, count)
        assert int(count.item()) == 1 and int(records[0, 0].item()) == tiles - 1
        hit.zero_()
        mat.compact_frames(frames, hit, records, count)
        assert int(count.item()) == 0
        mat.SetVector("_BackgroundColor", [1.0, 0.5, 0.0, 1.0])
        mat.expand_frames(records, 0, back)
        torch.cuda.synchronize()
        assert bool((back.view(-1, 4) == torch.tensor([255, 128, 0, 255], dtype=torch.uint8, device=dev)).all())


@pytest.mark.parametrize("name,with_src", [("default_640x360", False), ("default_src_depth_333x187", True), ("tiny_17x3", False)])
def test_update_frames_equals_expand(torch_cuda, name, with_src):
    """a displaying GPU that keeps its dense frames: update(old records -> new records) leaves byte for byte what a
    full expansion of the new records builds, for hit regions that shrink, grow and vanish, with and without a source
    image; the pixels outside the two record sets are never written"""
    torch = torch_cuda
    from helpers import RECORD_WORDS
    from raymarching_amd import Material
    from raymarching_amd.material import _ptr, current_stream_handle
    case = cases.BY_NAME[name]
    dev = "cuda:0"
    w, h, n = case.w, case.h, 2
    tiles = ((w + 7) // 8) * ((h + 7) // 8) * n
    with Material() as mat:
        src, _ = case.bind(mat)
        dsrc = torch.from_numpy(src).to(dev) if (src is not None and with_src) else None
        depth = torch.zeros((h, w), dtype=torch.float32, device=dev)
        sets = []
        for dists in ((50.0, 9.0), (6.0, 50.0), (0.5, 0.5), (50.0, 50.0)):      # the last but one: nothing is hit
            frames = torch.zeros((n, h, w, 4), dtype=torch.uint8, device=dev)
            hit = torch.zeros((n, h, w), dtype=torch.uint8, device=dev)
            for f in range(n):
                mat.SetFloat("_MaxDist", dists[f])
                mat._ck(mat._lib.rm_blit_device(mat.handle, _ptr(dsrc), None, _ptr(frames[f]), _ptr(depth), _ptr(hit[f]), w, h,
                                                current_stream_handle(dev)))
            rec = torch.zeros((tiles, RECORD_WORDS), dtype=torch.int32, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            mat.compact_frames(frames, hit, rec, cnt)
            sets.append((frames, rec, int(cnt.item())))
        assert sets[2][2] == 0 and sets[0][2] > 0 and sets[1][2] > 0
        ids = [set(r[:k, 0].cpu().tolist()) for _, r, k in sets]
        assert ids[0] != ids[1] and ids[1] != ids[3], "the record sets should name different tiles"
        dense = torch.full((n, h, w, 4), 9, dtype=torch.uint8, device=dev)
        mat.expand_frames(sets[0][1], sets[0][2], dense, src=dsrc)
        torch.cuda.synchronize()
        assert torch.equal(dense, sets[0][0])
        for old, new in ((0, 1), (1, 3), (3, 2), (2, 0), (0, 0)):
            mat.update_frames(sets[old][1], sets[old][2], sets[new][1], sets[new][2], dense, src=dsrc)
            torch.cuda.synchronize()
            assert torch.equal(dense, sets[new][0]), f"update {old} -> {new} differs from the frames the blit wrote"
        # only the tiles of the two record sets are written
        dense.fill_(7)
        mat.update_frames(sets[1][1], sets[1][2], sets[1][1], sets[1][2], dense, src=dsrc)
        torch.cuda.synchronize()
        touched = (dense != 7).any(dim=-1)
        tile_ids = sets[1][1][:sets[1][2], 0].cpu().numpy().astype(np.int64)
        want = np.zeros((n, h, w), dtype=bool)
        tx, ty = (w + 7) // 8, (h + 7) // 8
        for t in tile_ids:
            f, t2 = divmod(int(t), tx * ty)
            j, i = divmod(t2, tx)
            want[f, j * 8:j * 8 + 8, i * 8:i * 8 + 8] = True
        assert not (touched.cpu().numpy() & ~want).any()


@pytest.mark.parametrize("name", ["shadowmap_lookup_persp_320x180", "shadowmap_lookup_dir_320x180"])
def test_shadow_map_wiring_of_the_script(oracle, torch_cuda, name):
    """Raymarching.OnRenderImage with the three lines the reference keeps commented out (Raymarching.cs:91-93) switched on:
    shadowMap.Render(), SetTexture("_ShadowMap"), SetMatrix("_ShadowMapVPMatrix"), then the blit reads the map"""
    torch = torch_cuda
    from raymarching_amd import Material, Raymarching, RenderTexture, ShadowMap, ShadowMapDirectional, scenes
    case = cases.BY_NAME[name]
    ref = cases.oracle_run(case, oracle)
    kind, sw, sh = case.shadow
    with Material() as mat:
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        target = RenderTexture(sw, sh, device="cuda:0")
        sm = (ShadowMap(cases.light_camera(sw, sh), mat, target) if kind == "persp"
              else ShadowMapDirectional(cases.light_rectangle(), mat, target))
        rm = Raymarching(scenes.default_camera(case.w, case.h), mat, scenes.default_transforms(), shadowMap=sm)
        rt = RenderTexture(case.w, case.h, device="cuda:0")
        for _ in range(2):
            rm.OnRenderImage(None, rt)
        st = mat.stats()
        torch.cuda.synchronize()
        got = {"rgba": rt.color.cpu().numpy(), "depth": rt.depth.cpu().numpy(), "hit": rt.hit.cpu().numpy()}
        assert_frame_equal(got, ref, name)
        assert st["sum_steps_s"] == 0 and st["sum_steps_p"] == ref["stats"]["sum_steps_p"]
        # unbinding the map with the mode still set is an error, not a crash
        mat.SetTexture("_ShadowMap", None)
        from raymarching_amd import native
        with pytest.raises(native.RmError):
            rm.bind()
            from raymarching_amd import Graphics
            Graphics.Blit(None, rt, mat)


def test_host_blit_into_page_locked_buffers(oracle, torch_cuda):
    """rm_blit into memory from rm_host_alloc (the buffers the C# component uses) equals the oracle's frame"""
    import ctypes as C
    from raymarching_amd import Material
    case = cases.BY_NAME["default_src_depth_333x187"]
    ref = cases.oracle_run(case, oracle)
    w, h = case.w, case.h
    with Material() as mat:
        src, sd = case.bind(mat)
        lib = mat._lib
        n = w * h
        bufs = [lib.rm_host_alloc(mat.handle, n * 4) for _ in range(4)] + [lib.rm_host_alloc(mat.handle, n)]
        assert all(bufs)
        psrc, psd, pc, pd, ph = bufs
        C.memmove(psrc, src.ctypes.data, n * 4)
        C.memmove(psd, sd.ctypes.data, n * 4)
        for _ in range(2):
            mat._ck(lib.rm_blit(mat.handle, psrc, psd, pc, pd, ph, w, h))
        got = {"rgba": np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint8)), shape=(h, w, 4)).copy(),
               "depth": np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_float)), shape=(h, w)).copy(),
               "hit": np.ctypeslib.as_array(C.cast(ph, C.POINTER(C.c_uint8)), shape=(h, w)).copy()}
        for b in bufs:
            lib.rm_host_free(mat.handle, b)
        assert lib.rm_host_alloc(mat.handle, 0) is None
    assert_frame_equal(got, ref, "pinned host buffers")


@pytest.mark.parametrize("name", ["default_640x360", "default_src_depth_333x187", "orbit_k23_480x270"])
def test_async_host_blit_equals_rm_blit(oracle, torch_cuda, name):
    """rm_blit_async + rm_blit_wait: frames queued back to back (two in flight, the copies back of one beside the march of
    the next) arrive byte for byte as rm_blit delivers them, in page-locked and in pageable buffers, also when sizes and
    the presence of a source image change from call to call and when synchronous calls are mixed in"""
    import ctypes as C
    from raymarching_amd import Material, native
    case = cases.BY_NAME[name]
    ref = cases.oracle_run(case, oracle)
    small = cases.BY_NAME["tiny_17x3"]
    ref_small = cases.oracle_run(small, oracle)
    lib = native.lib()
    with Material() as mat:
        src, sd = case.bind(mat)
        n = case.w * case.h

        def view(p, count, dt):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(dt)), shape=(count,))

        sets = []
        for pinned in (True, True, False):
            if pinned:
                pr, pd, ph = (lib.rm_host_alloc(mat.handle, n * 4), lib.rm_host_alloc(mat.handle, n * 4), lib.rm_host_alloc(mat.handle, n))
                assert pr and pd and ph
                sets.append((pr, pd, ph, view(pr, n * 4, C.c_uint8), view(pd, n, C.c_float), view(ph, n, C.c_uint8), True))
            else:
                a, b, c_ = np.zeros(n * 4, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.uint8)
                sets.append((a.ctypes.data, b.ctypes.data, c_.ctypes.data, a, b, c_, False))
        sp = src.ctypes.data if src is not None else None
        dp = sd.ctypes.data if sd is not None else None
        tickets = []
        for i in range(7):
            pr, pd, ph, ar, ad, ah, _ = sets[i % 3]
            if i >= 2:                                      # frame i - 2 must be complete before its buffers are reused (i % 3 != (i - 2) % 3,
                mat._ck(lib.rm_blit_wait(mat.handle, tickets[i - 2]))          # but at most two may be in flight anyway)
                _, _, _, br, bd, bh, _ = sets[(i - 2) % 3]
                got = {"rgba": br.reshape(case.h, case.w, 4).copy(), "depth": bd.reshape(case.h, case.w).copy(), "hit": bh.reshape(case.h, case.w).copy()}
                assert_frame_equal(got, ref, f"{name} async frame {i - 2}")
                br[:] = 0; bd[:] = 0; bh[:] = 0
            t = C.c_longlong(-1)
            mat._ck(lib.rm_blit_async(mat.handle, sp, dp, pr, pd, ph, case.w, case.h, C.byref(t)))
            assert t.value == i
            tickets.append(t.value)
        assert mat.stats()["n_hit"] == int(ref["hit"].sum())
        for i in (5, 6):
            mat._ck(lib.rm_blit_wait(mat.handle, tickets[i]))
            _, _, _, br, bd, bh, _ = sets[i % 3]
            got = {"rgba": br.reshape(case.h, case.w, 4).copy(), "depth": bd.reshape(case.h, case.w).copy(), "hit": bh.reshape(case.h, case.w).copy()}
            assert_frame_equal(got, ref, f"{name} async frame {i}")
        mat._ck(lib.rm_blit_wait(mat.handle, tickets[0]))               # long done
        assert lib.rm_blit_wait(mat.handle, 99) == native.RM_E_ARG
        # a smaller frame without source image queued right behind a large one with, then a synchronous call
        pr, pd, ph, ar, ad, ah, _ = sets[0]
        t0, t1 = C.c_longlong(), C.c_longlong()
        mat._ck(lib.rm_blit_async(mat.handle, sp, dp, pr, pd, ph, case.w, case.h, C.byref(t0)))
        small.bind(mat)
        qr, qd, qh = np.zeros((small.h, small.w, 4), np.uint8), np.zeros((small.h, small.w), np.float32), np.zeros((small.h, small.w), np.uint8)
        mat._ck(lib.rm_blit_async(mat.handle, None, None, qr.ctypes.data, qd.ctypes.data, qh.ctypes.data, small.w, small.h, C.byref(t1)))
        sr, sdp, sh = np.zeros_like(qr), np.zeros_like(qd), np.zeros_like(qh)
        mat._ck(lib.rm_blit(mat.handle, None, None, sr.ctypes.data, sdp.ctypes.data, sh.ctypes.data, small.w, small.h))   # waits for both
        assert_frame_equal({"rgba": sr, "depth": sdp, "hit": sh}, ref_small, "sync after async")
        mat._ck(lib.rm_blit_wait(mat.handle, t1.value))
        assert_frame_equal({"rgba": qr, "depth": qd, "hit": qh}, ref_small, "small async")
        mat._ck(lib.rm_blit_wait(mat.handle, t0.value))
        got = {"rgba": ar.reshape(case.h, case.w, 4).copy(), "depth": ad.reshape(case.h, case.w).copy(), "hit": ah.reshape(case.h, case.w).copy()}
        assert_frame_equal(got, ref, "large async before the small one")
        # four frames queued into four sets of buffers without a wait in between, then the OLDEST ticket first: the wait must
        # not return before that frame is in host memory although its staging set has been reused since
        case.bind(mat)
        extra = (np.zeros(n * 4, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.uint8))
        four = [s_[3:6] for s_ in sets] + [extra]
        for ar_, ad_, ah_ in four:
            ar_[:] = 0; ad_[:] = 0; ah_[:] = 0
        tk = []
        for ar_, ad_, ah_ in four:
            t = C.c_longlong(-1)
            mat._ck(lib.rm_blit_async(mat.handle, sp, dp, ar_.ctypes.data, ad_.ctypes.data, ah_.ctypes.data, case.w, case.h, C.byref(t)))
            tk.append(t.value)
        for i, (ar_, ad_, ah_) in enumerate(four):
            mat._ck(lib.rm_blit_wait(mat.handle, tk[i]))
            got = {"rgba": ar_.reshape(case.h, case.w, 4).copy(), "depth": ad_.reshape(case.h, case.w).copy(), "hit": ah_.reshape(case.h, case.w).copy()}
            assert_frame_equal(got, ref, f"{name} four queued, frame {i}")
        for pr, pd, ph, _, _, _, pinned in sets:
            if pinned:
                for p_ in (pr, pd, ph):
                    lib.rm_host_free(mat.handle, p_)
