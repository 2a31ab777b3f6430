"""rm_blit_device_noalloc (the entry point that works in static device scratch and allocates nothing) against
rm_blit_device, through the C ABI: bit-equal outputs on two calls in a row, the second with a smaller frame."""
import numpy as np
import pytest

import cases
from helpers import assert_frame_equal

pytestmark = pytest.mark.gpu


def _blit(mat, torch, entry, case, src, sd, stream=None):
    from raymarching_amd.material import _ptr, current_stream_handle
    dev = "cuda:0"
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    dsrc, dsd = t(src), t(sd)
    rgba = torch.full((case.h, case.w, 4), 7, dtype=torch.uint8, device=dev)
    depth = torch.full((case.h, case.w), -1.0, dtype=torch.float32, device=dev)
    hit = torch.full((case.h, case.w), 9, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    fn = getattr(mat._lib, entry)
    mat._ck(fn(mat.handle, _ptr(dsrc), _ptr(dsd), _ptr(rgba), _ptr(depth), _ptr(hit), case.w, case.h,
               stream if stream is not None else current_stream_handle(dev)))
    free1 = torch.cuda.mem_get_info()[0]
    st = mat.stats()
    torch.cuda.synchronize()
    return {"rgba": rgba.cpu().numpy(), "depth": depth.cpu().numpy(), "hit": hit.cpu().numpy(), "stats": st,
            "free_delta": free0 - free1}


@pytest.mark.parametrize("kernel", [0, 1, 2, 3])
def test_noalloc_equals_blit_device_two_calls_second_smaller(kernel):
    import torch
    assert torch.cuda.is_available()
    from raymarching_amd import Material
    big, small = cases.BY_NAME["default_640x360"], cases.BY_NAME["default_src_depth_333x187"]
    # the expected frames, from the allocating entry point on a context of its own
    want = []
    for case in (big, small):
        with Material() as m:
            m.set_kernel(kernel)
            src, sd = case.bind(m)
            want.append(_blit(m, torch, "rm_blit_device", case, src, sd))
    # two calls in a row on one context that has never touched the allocating path: large, then small
    with Material() as m:
        m.set_kernel(kernel)
        for i, case in enumerate((big, small, big, big)):
            src, sd = case.bind(m)
            got = _blit(m, torch, "rm_blit_device_noalloc", case, src, sd)
            ref = want[0] if case is big else want[1]
            assert_frame_equal(got, ref, f"call {i} ({case.name})")
            assert got["stats"] == ref["stats"], f"call {i}: counters differ"
            if i > 0:
                assert got["free_delta"] == 0, f"call {i}: free device memory changed by {got['free_delta']} bytes"


def test_noalloc_shares_scratch_between_contexts_and_refuses_what_it_cannot_hold():
    import torch
    from raymarching_amd import Material, native, scenes
    from raymarching_amd.material import _ptr
    case = cases.BY_NAME["orbit_k23_480x270"]
    with Material() as ref_m:
        src, sd = case.bind(ref_m)
        want = _blit(ref_m, torch, "rm_blit_device", case, src, sd)
    # two contexts taking turns on the one array of this GPU (in stream order)
    with Material() as a, Material() as b:
        for m in (a, b, a, b):
            src, sd = case.bind(m)
            got = _blit(m, torch, "rm_blit_device_noalloc", case, src, sd)
            assert_frame_equal(got, want, "shared scratch")
            assert got["stats"] == want["stats"]
        lib = native.lib()
        buf = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
        # beyond the scratch: 7680x4320 has 129 600 blocks of 16x16 pixels
        assert lib.rm_blit_device_noalloc(a.handle, None, None, _ptr(buf), None, None, 7680, 4320, None) == native.RM_E_RANGE
        assert b"scratch" in lib.rm_last_error(a.handle)
        a.SetPrimitives(scenes.stress_primitives(8))
        assert lib.rm_blit_device_noalloc(a.handle, None, None, _ptr(buf), None, None, 4, 4, None) == native.RM_E_ARG


def test_counters_of_a_noalloc_call_are_gone_once_the_scratch_is_reused():
    """the counter slots live in the GPU's shared array: after another context's call, the first context's rm_get_stats
    must fail rather than return the other call's counters"""
    import torch
    from raymarching_amd import Material, native
    import ctypes as C
    case = cases.BY_NAME["default_640x360"]
    with Material() as a, Material() as b:
        sa, da = case.bind(a)
        got_a = _blit(a, torch, "rm_blit_device_noalloc", case, sa, da)
        assert got_a["stats"]["n_hit"] > 0
        sb, db = case.bind(b)
        _blit(b, torch, "rm_blit_device_noalloc", case, sb, db)
        st = native.RmStats()
        assert a._lib.rm_get_stats(a.handle, C.byref(st)) == native.RM_E_ARG
        assert b"reused" in a._lib.rm_last_error(a.handle)
        assert a._lib.rm_debug_counter_slots(a.handle, None, 0, None, None) == native.RM_E_ARG
        assert b.stats() == got_a["stats"]                       # b's own counters are still there
        # an allocating call of `a` has its counters in memory of its own again
        again = _blit(a, torch, "rm_blit_device", case, sa, da)
        assert again["stats"] == got_a["stats"]


@pytest.mark.parametrize("kernel", [0, 1])
def test_noalloc_and_allocating_calls_alternate_on_two_streams(kernel):
    import torch
    from raymarching_amd import Material
    case = cases.BY_NAME["default_src_depth_333x187"]
    with Material() as ref_m:
        ref_m.set_kernel(kernel)
        src, sd = case.bind(ref_m)
        want = _blit(ref_m, torch, "rm_blit_device", case, src, sd)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with Material() as m:
        m.set_kernel(kernel)
        src, sd = case.bind(m)
        for i in range(6):
            entry, st = (("rm_blit_device_noalloc", s1) if i % 2 == 0 else ("rm_blit_device", s2))
            got = _blit(m, torch, entry, case, src, sd, stream=st.cuda_stream)
            assert_frame_equal(got, want, f"call {i} ({entry})")
            assert got["stats"] == want["stats"]


@pytest.mark.parametrize("kernel", [0, 1])
def test_noalloc_call_recorded_into_a_graph_replays(kernel):
    """a captured rm_blit_device_noalloc call is self-contained (a memset node for its claim counters, no frame-to-frame
    state): the graph replays any number of times, interleaved with direct calls, always with the frame of rm_blit_device"""
    import torch
    from raymarching_amd import Material
    from raymarching_amd.material import _ptr
    dev = "cuda:0"
    case = cases.BY_NAME["default_640x360"]
    with Material() as ref_m:
        ref_m.set_kernel(kernel)
        src, sd = case.bind(ref_m)
        want = _blit(ref_m, torch, "rm_blit_device", case, src, sd)
    with Material() as m:
        m.set_kernel(kernel)
        case.bind(m)
        rgba = torch.zeros((case.h, case.w, 4), dtype=torch.uint8, device=dev)
        depth = torch.zeros((case.h, case.w), dtype=torch.float32, device=dev)
        hit = torch.zeros((case.h, case.w), dtype=torch.uint8, device=dev)

        def call(stream):
            m._ck(m._lib.rm_blit_device_noalloc(m.handle, None, None, _ptr(rgba), _ptr(depth), _ptr(hit), case.w, case.h, stream))

        def check(what):
            torch.cuda.synchronize()
            got = {"rgba": rgba.cpu().numpy(), "depth": depth.cpu().numpy(), "hit": hit.cpu().numpy()}
            assert_frame_equal(got, want, what)
            rgba.fill_(3); depth.fill_(-2.0); hit.fill_(5)
            torch.cuda.synchronize()

        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            call(side.cuda_stream)                  # set-up outside capture: scratch address, occupancy figures
        check("direct call before the capture")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            call(torch.cuda.current_stream().cuda_stream)
        for i in range(3):
            g.replay()
            check(f"replay {i}")
        with torch.cuda.stream(side):
            call(side.cuda_stream)                  # a direct call between replays keeps its frame-to-frame rotation
        check("direct call after replays")
        g.replay()
        check("replay after a direct call")
        assert m.stats() == want["stats"]
        del g
