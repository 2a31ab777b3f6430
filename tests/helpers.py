"""Shared comparison helpers for the parity tests."""
import hashlib

import numpy as np


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_frame_equal(got: dict, ref: dict, what: str = "") -> None:
    """Bit-exact bar for the exact kernels: mask, depth bits and RGBA8 all equal."""
    assert np.array_equal(got["hit"], ref["hit"]), f"{what}: hit mask differs in {(got['hit'] != ref['hit']).sum()} pixels"
    # a NaN is a NaN: x86 and the GPU produce different sign/payload bits for an invalid operation
    gn, rn = np.isnan(got["depth"]), np.isnan(ref["depth"])
    db = (gn != rn) | ((bits(got["depth"]) != bits(ref["depth"])) & ~gn)
    assert not db.any(), f"{what}: depth differs in {db.sum()} pixels"
    dc = got["rgba"] != ref["rgba"]
    assert not dc.any(), f"{what}: colour differs in {dc.any(axis=-1).sum()} pixels"


def assert_frame_close(got: dict, ref: dict, what: str = "") -> dict:
    """Tolerance bar of the port: hit mask equal except <= 0.01 % of pixels, depth
    within 1e-4 relative, colour within 1/255 on >= 99.9 % of pixels."""
    n = ref["hit"].size
    mask_bad = int((got["hit"] != ref["hit"]).sum())
    same = got["hit"] == ref["hit"]
    rel = np.abs(got["depth"].astype(np.float64) - ref["depth"]) / np.maximum(np.abs(ref["depth"]), 1e-30)
    depth_bad = int((rel[same] > 1e-4).sum())
    cd = np.abs(got["rgba"].astype(np.int32) - ref["rgba"].astype(np.int32)).max(axis=-1)
    col_bad = int((cd > 1).sum())
    assert mask_bad <= 1e-4 * n, f"{what}: {mask_bad} of {n} mask pixels differ"
    assert depth_bad <= 1e-4 * n, f"{what}: {depth_bad} of {n} depths off by more than 1e-4 relative"
    assert col_bad <= 1e-3 * n, f"{what}: {col_bad} of {n} pixels off by more than 1/255"
    return {"mask_bad": mask_bad, "depth_bad": depth_bad, "col_bad": col_bad}
