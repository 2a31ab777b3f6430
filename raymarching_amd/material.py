"""Host-side mirror of the Unity objects the reference's scripts drive the shader
through: ``Material`` (a property bag: SetMatrix/SetFloat/..., as used at
Assets/Raymarching.cs:86-90 and Assets/ShadowMap.cs:52-53), ``RenderTexture`` and
``Graphics.Blit`` (Assets/Raymarching.cs:94, Assets/ShadowMap.cs:54).

Every call forwards to the C ABI of librmhip.so; nothing is computed in Python.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import native
from .unity import Matrix4x4

_F = np.float32


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Material:
    """One material bound to one GPU (``rm_ctx`` of include/rmhip.h).

    ``shader`` is informational ("Hidden/Raymarching" or "Hidden/ShadowMap");
    the property names and their meaning are the shader's.
    """

    def __init__(self, shader: str = "Hidden/Raymarching", device: int = 0):
        self.shader = shader
        self.device = device
        self._lib = native.lib()
        self._ctx = self._lib.rm_create(device)
        if not self._ctx:
            raise native.RmError(native.RM_E_ARG, self._lib.rm_last_error(None).decode())

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.rm_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._ctx

    def _ck(self, rc: int) -> None:
        native.check(self._ctx, rc)

    # ---- Unity Material API -------------------------------------------------
    def SetFloat(self, name: str, value: float) -> None:
        self._ck(self._lib.rm_set_float(self._ctx, name.encode(), float(value)))

    def SetInt(self, name: str, value: int) -> None:
        self._ck(self._lib.rm_set_int(self._ctx, name.encode(), int(value)))

    def SetVector(self, name: str, v: Sequence[float]) -> None:
        a = np.zeros(4, dtype=_F)
        v = np.asarray(v, dtype=_F).reshape(-1)
        a[: v.size] = v[:4]
        self._ck(self._lib.rm_set_vector(self._ctx, name.encode(), _fp(a)))

    SetColor = SetVector

    def SetMatrix(self, name: str, m) -> None:
        a = np.ascontiguousarray(m.m if isinstance(m, Matrix4x4) else np.asarray(m, dtype=_F).reshape(16), dtype=_F)
        self._ck(self._lib.rm_set_matrix(self._ctx, name.encode(), _fp(a)))

    def SetTexture(self, name: str, tex) -> None:
        """Material.SetTexture for a device-resident fp32 plane [h,w] (a RenderTexture's depth, or a tensor); only
        `_ShadowMap` is read by the march (rm_set_texture_device).  None unbinds."""
        t = getattr(tex, "depth", tex)
        if t is None:
            self._ck(self._lib.rm_set_texture_device(self._ctx, name.encode(), None, 0, 0))
            return
        if isinstance(t, np.ndarray):
            raise ValueError("SetTexture takes a device-resident plane")
        self._textures = getattr(self, "_textures", {})
        self._textures[name] = t                                  # keeps the plane alive while it is bound
        self._ck(self._lib.rm_set_texture_device(self._ctx, name.encode(), _ptr(t), int(t.shape[1]), int(t.shape[0])))

    def GetFloat(self, name: str) -> float:
        v = C.c_float()
        self._ck(self._lib.rm_get_float(self._ctx, name.encode(), C.byref(v)))
        return float(v.value)

    def GetInt(self, name: str) -> int:
        v = C.c_int()
        self._ck(self._lib.rm_get_int(self._ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def GetVector(self, name: str) -> np.ndarray:
        a = np.zeros(4, dtype=_F)
        self._ck(self._lib.rm_get_vector(self._ctx, name.encode(), _fp(a)))
        return a

    def GetMatrix(self, name: str) -> Matrix4x4:
        a = np.zeros(16, dtype=_F)
        self._ck(self._lib.rm_get_matrix(self._ctx, name.encode(), _fp(a)))
        return Matrix4x4(a)

    def HasProperty(self, name: str) -> bool:
        return bool(self._lib.rm_has_property(self._ctx, name.encode()))

    # ---- beyond the reference's 4-slot scene ----------------------------------
    def SetPrimitives(self, prims: Sequence[dict]) -> None:
        """prims: dicts with type,pos,quat,scale,color,params (see rm_primitive)."""
        n = len(prims)
        arr = (native.RmPrimitive * max(n, 1))()
        for i, p in enumerate(prims):
            arr[i].type = int(p["type"])
            arr[i].op = int(p.get("op", 0))
            arr[i].pos[:] = [float(x) for x in p["pos"]]
            arr[i].quat[:] = [float(x) for x in p["quat"]]
            arr[i].scale[:] = [float(x) for x in p["scale"]]
            arr[i].color[:] = [float(x) for x in p["color"]]
            pr = list(p.get("params", ())) + [0.0] * 4
            arr[i].params[:] = [float(x) for x in pr[:4]]
        self._ck(self._lib.rm_set_primitives(self._ctx, arr, n))

    def stats(self) -> dict:
        s = native.RmStats()
        self._ck(self._lib.rm_get_stats(self._ctx, C.byref(s)))
        return {"n_pixels": s.n_pixels, "n_hit": s.n_hit, "sum_steps_p": s.sum_steps_p, "sum_steps_s": s.sum_steps_s}

    def leaf_stats(self) -> dict:
        """leaf evaluations executed / possible in the last call of the table kernel with exact culling (rm_get_leaf_stats)"""
        v, p = C.c_uint64(), C.c_uint64()
        self._ck(self._lib.rm_get_leaf_stats(self._ctx, C.byref(v), C.byref(p)))
        return {"visited": int(v.value), "possible": int(p.value)}

    def pack_rgb(self, rgba, rgb, stream=None) -> None:
        """device RGBA8 tensor [..., 4] -> device RGB8 tensor [..., 3] (frame transport: the blit's alpha is constant)"""
        n = rgba.numel() // 4
        self._ck(self._lib.rm_pack_rgb_device(self._ctx, _ptr(rgba), _ptr(rgb), n,
                                              stream if stream is not None else current_stream_handle(rgba.device)))

    def unpack_rgb(self, rgb, rgba, stream=None) -> None:
        """device RGB8 tensor -> device RGBA8 tensor with alpha 255"""
        n = rgba.numel() // 4
        self._ck(self._lib.rm_unpack_rgb_device(self._ctx, _ptr(rgb), _ptr(rgba), n,
                                                stream if stream is not None else current_stream_handle(rgba.device)))

    def compact_frames(self, frames, hit, records, count, stream=None) -> None:
        """device RGBA8 frames [n,h,w,4] + hit masks [n,h,w] -> compact tile records [cap,49] int32 and their number
        (rm_compact_frames_device)"""
        n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        need = n * ((w + 7) // 8) * ((h + 7) // 8)
        if records.numel() * records.element_size() < need * self._lib.rm_tile_record_bytes():
            raise ValueError("record buffer smaller than one record per tile")
        self._ck(self._lib.rm_compact_frames_device(self._ctx, _ptr(frames), _ptr(hit), w, h, n, _ptr(records), _ptr(count),
                                                    stream if stream is not None else current_stream_handle(frames.device)))

    def expand_frames(self, records, n_records: int, dst, src=None, stream=None) -> None:
        """inverse of compact_frames into dst [n,h,w,4] on the GPU that displays the frames (rm_expand_frames_device)"""
        n, h, w = int(dst.shape[0]), int(dst.shape[1]), int(dst.shape[2])
        self._ck(self._lib.rm_expand_frames_device(self._ctx, _ptr(records) if n_records else None, int(n_records), _ptr(src),
                                                   _ptr(dst), w, h, n,
                                                   stream if stream is not None else current_stream_handle(dst.device)))

    def update_frames(self, old_records, n_old: int, new_records, n_new: int, dst, src=None, stream=None) -> None:
        """dst [n,h,w,4] was built from old_records by expand_frames/update_frames; afterwards it is what expand_frames
        builds from new_records, with only the tiles of the two record sets written (rm_update_frames_device)"""
        n, h, w = int(dst.shape[0]), int(dst.shape[1]), int(dst.shape[2])
        self._ck(self._lib.rm_update_frames_device(self._ctx, _ptr(old_records) if n_old else None, int(n_old),
                                                   _ptr(new_records) if n_new else None, int(n_new), _ptr(src), _ptr(dst), w, h, n,
                                                   stream if stream is not None else current_stream_handle(dst.device)))

    def set_kernel(self, variant: int) -> None:
        self._ck(self._lib.rm_set_kernel(self._ctx, int(variant)))


class RenderTexture:
    """Colour (RGBA8) + depth (fp32) + hit-mask target.  Row 0 is the bottom row.

    ``device=None`` keeps numpy arrays on the host (rm_blit copies over PCIe);
    ``device="cuda:N"`` keeps torch tensors in HBM (rm_blit_device, no copies).
    """

    def __init__(self, width: int, height: int, device: Optional[str] = None, hit: bool = True):
        self.width, self.height, self.device = int(width), int(height), device
        if device is None:
            self.color = np.zeros((height, width, 4), dtype=np.uint8)
            self.depth = np.zeros((height, width), dtype=np.float32)
            self.hit = np.zeros((height, width), dtype=np.uint8) if hit else None
        else:
            import torch
            self.color = torch.zeros((height, width, 4), dtype=torch.uint8, device=device)
            self.depth = torch.zeros((height, width), dtype=torch.float32, device=device)
            self.hit = torch.zeros((height, width), dtype=torch.uint8, device=device) if hit else None

    @staticmethod
    def GetTemporary(width: int, height: int, device: Optional[str] = None) -> "RenderTexture":
        return RenderTexture(width, height, device)

    @staticmethod
    def ReleaseTemporary(rt: "RenderTexture") -> None:
        pass


def _ptr(a) -> Optional[int]:
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("host images must be C-contiguous")
        return a.ctypes.data
    if not a.is_contiguous():
        raise ValueError("device images must be contiguous")
    return a.data_ptr()


def current_stream_handle(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


class Graphics:
    @staticmethod
    def Blit(src: Optional[RenderTexture], dest: RenderTexture, mat: Material, scene_depth=None) -> None:
        """Graphics.Blit(src, dest, mat): runs the material's fragment program over
        every pixel of ``dest``.  ``src`` supplies _MainTex (None: the constant
        _BackgroundColor); ``scene_depth`` the camera depth texture as linear eye
        depth (None: far plane everywhere)."""
        w, h = dest.width, dest.height
        if src is not None and (src.width != w or src.height != h):
            raise ValueError("src and dest sizes differ")
        s = _ptr(src.color) if src is not None else None
        if dest.device is None:
            mat._ck(mat._lib.rm_blit(mat.handle, s, _ptr(scene_depth), _ptr(dest.color), _ptr(dest.depth),
                                     _ptr(dest.hit), w, h))
        else:
            mat._ck(mat._lib.rm_blit_device(mat.handle, s, _ptr(scene_depth), _ptr(dest.color), _ptr(dest.depth),
                                            _ptr(dest.hit), w, h, current_stream_handle(dest.device)))
