"""The opt-in tolerance kernel (rm_set_kernel(ctx, 4): fused multiply-adds, raw v_sqrt_f32 and reciprocal multiplies in
the two march loops) against the strict kernels, whose frames equal the CPU oracle's bit for bit.  Bar: the tolerance
of the port -- hit mask equal except <= 0.01 % of the pixels, depth within 1e-4 relative, colour within 1/255 on
>= 99.9 % of the pixels (helpers.assert_frame_close)."""
import json
import os

import numpy as np
import pytest

import cases
from helpers import assert_frame_close, sha
from test_gpu_parity import GOLD, KERNELS, gpu_run, torch_cuda  # noqa: F401

pytestmark = pytest.mark.gpu
FAST = 4
REPORT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "fast_kernel_gate.jsonl")


def _report(name, res, n):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps({"case": name, "pixels": n, **res}) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("name", ["default_640x360", "default_1920x1080", "default_3840x2160", "orbit_k17_7680x4320",
                                  "orbit_k5_480x270", "orbit_k40_480x270", "default_src_depth_333x187", "fractional_steps_320x180",
                                  "transformed_400x225", "transformed_pow2_400x225", "transformed_truediv_400x225", "tiny_17x3"])
def test_fast_kernel_within_the_tolerance_of_the_port(torch_cuda, name):
    case = cases.BY_NAME[name]
    ref = gpu_run(case, torch_cuda, kernel=KERNELS["queue"])
    if name in GOLD and case.heavy:
        assert sha(ref["rgba"]) == GOLD[name]["sha_rgba"] and sha(ref["depth"]) == GOLD[name]["sha_depth"]      # = the oracle's frame
    got = gpu_run(case, torch_cuda, kernel=FAST, frames=2)
    res = assert_frame_close(got, ref, name)
    _report(name, res, case.w * case.h)
    assert got["stats"]["n_pixels"] == case.w * case.h
    # the counters follow the fast march: close to the strict ones, not equal
    assert abs(got["stats"]["sum_steps_p"] - ref["stats"]["sum_steps_p"]) <= 1e-3 * ref["stats"]["sum_steps_p"] + 64


@pytest.mark.parametrize("name", ["shadow_dir_3840x2160", "shadow_persp_100x100", "shadow_dir_256x144"])
def test_fast_kernel_depth_passes(torch_cuda, name):
    case = cases.BY_NAME[name]
    ref = gpu_run(case, torch_cuda, kernel=KERNELS["tile"])
    got = gpu_run(case, torch_cuda, kernel=FAST)
    n = case.w * case.h
    mask_bad = int((got["hit"] != ref["hit"]).sum())
    same = got["hit"] == ref["hit"]
    rel = np.abs(got["depth"].astype(np.float64) - ref["depth"]) / np.maximum(np.abs(ref["depth"]), 1e-30)
    depth_bad = int((rel[same] > 1e-4).sum())
    _report(name, {"mask_bad": mask_bad, "depth_bad": depth_bad, "col_bad": 0}, n)
    assert mask_bad <= 1e-4 * n and depth_bad <= 1e-4 * n


@pytest.mark.parametrize("name", ["csg7_480x270", "menger_closeup_320x180", "fx_bounce_320x180", "dom_repeat_xz_320x180",
                                  "prims4_as_table_320x180", "stress13_200x112", "stress256_320x180"])
def test_fast_kernel_falls_back_to_the_exact_kernels_elsewhere(oracle, torch_cuda, name):
    """table scenes, fractal bodies, the optional frag lines and the domain operators have no tolerance kernel: asking for
    it renders them with the exact ones (a per-step culling kernel for tables existed; on the 256-primitive scene it missed
    the port's tolerance and lost to the exact culling of the shared-tile kernel, so it was removed)"""
    from helpers import assert_frame_equal
    case = cases.BY_NAME[name]
    assert_frame_equal(gpu_run(case, torch_cuda, kernel=FAST), cases.oracle_run(case, oracle), name)
