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


# ---- compact frame transport, restated on the host for the tests (record layout of rm_compact_frames_device) ----
RECORD_WORDS = 49


def compact_frames_numpy(frames: np.ndarray, hit: np.ndarray) -> np.ndarray:
    """frames [n,h,w,4] u8, hit [n,h,w] u8 -> records [T,49] int32 in tile order: tile index, then 64 pixels as RGB8"""
    n, h, w = hit.shape
    tx, ty = (w + 7) // 8, (h + 7) // 8
    out = []
    for f in range(n):
        for j in range(ty):
            for i in range(tx):
                m = hit[f, j * 8:j * 8 + 8, i * 8:i * 8 + 8]
                if not m.any():
                    continue
                px = np.zeros((8, 8, 3), dtype=np.uint8)
                blk = frames[f, j * 8:j * 8 + 8, i * 8:i * 8 + 8, :3]
                px[:blk.shape[0], :blk.shape[1]] = blk
                rec = np.empty(RECORD_WORDS, dtype=np.uint32)
                rec[0] = (f * ty + j) * tx + i
                rec[1:] = px.reshape(-1).view("<u4")
                out.append(rec)
    return (np.stack(out) if out else np.zeros((0, RECORD_WORDS), dtype=np.uint32)).view(np.int32)


def expand_frames_numpy(records: np.ndarray, n: int, h: int, w: int, background) -> np.ndarray:
    """background: RGBA8 colour (4 bytes) or a source image [h,w,4]; returns frames [n,h,w,4] with alpha 255"""
    bg = np.asarray(background, dtype=np.uint8)
    frames = np.empty((n, h, w, 4), dtype=np.uint8)
    frames[...] = bg
    frames[..., 3] = 255
    tx, ty = (w + 7) // 8, (h + 7) // 8
    for rec in np.ascontiguousarray(records).view(np.uint32):
        t = int(rec[0])
        f, t2 = divmod(t, tx * ty)
        j, i = divmod(t2, tx)
        px = np.ascontiguousarray(rec[1:]).view(np.uint8).reshape(8, 8, 3)
        hh, ww = min(8, h - j * 8), min(8, w - i * 8)
        frames[f, j * 8:j * 8 + hh, i * 8:i * 8 + ww, :3] = px[:hh, :ww]
    return frames


def update_frames_numpy(frames: np.ndarray, old_records: np.ndarray, new_records: np.ndarray, background) -> None:
    """in place: the tiles the old records name go back to the background, then the new records are written"""
    n, h, w = frames.shape[:3]
    bg = np.asarray(background, dtype=np.uint8)
    tx, ty = (w + 7) // 8, (h + 7) // 8
    for rec in np.ascontiguousarray(old_records).view(np.uint32):
        f, t2 = divmod(int(rec[0]), tx * ty)
        j, i = divmod(t2, tx)
        sl = (f, slice(j * 8, j * 8 + 8), slice(i * 8, i * 8 + 8))
        frames[sl] = bg if bg.ndim == 1 else bg[sl[1:]]
        frames[sl][..., 3] = 255
    for rec in np.ascontiguousarray(new_records).view(np.uint32):
        f, t2 = divmod(int(rec[0]), tx * ty)
        j, i = divmod(t2, tx)
        px = np.ascontiguousarray(rec[1:]).view(np.uint8).reshape(8, 8, 3)
        hh, ww = min(8, h - j * 8), min(8, w - i * 8)
        frames[f, j * 8:j * 8 + hh, i * 8:i * 8 + ww, :3] = px[:hh, :ww]
