"""raymarching_amd -- MI355X (gfx950) sphere tracing behind the uniforms and entry
points of the Unity project symbiosinx/Raymarching.

The compute path is librmhip.so (hand-written HIP, C ABI in include/rmhip.h);
this package is the thin host-side mirror of the reference's C# scripts.
"""
from .material import Graphics, Material, RenderTexture          # noqa: F401
from .scripts import Raymarching, ShadowMap, ShadowMapDirectional, get_frustum_corners   # noqa: F401
from .unity import Camera, Matrix4x4, Transform                   # noqa: F401
from . import scenes                                              # noqa: F401

__all__ = ["Graphics", "Material", "RenderTexture", "Raymarching", "ShadowMap", "ShadowMapDirectional",
           "get_frustum_corners", "Camera", "Matrix4x4", "Transform", "scenes"]
