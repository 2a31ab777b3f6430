"""The C ABI without a GPU: the library loads, exports every symbol the header
declares, and the property bag behaves like a Unity Material."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raymarching_amd import Material, native
from raymarching_amd.unity import Matrix4x4

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "rmhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rm_[a-z_0-9]+|UnityPlugin[A-Za-z]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = header_functions()
    assert len(names) >= 20
    lib = native.lib()
    bound = {n for n, _, _ in native.SYMBOLS}
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rmhip.h but not exported"
        assert n in bound, f"{n} has no ctypes signature in native.SYMBOLS"
    assert lib.rm_version().decode().startswith("rmhip")


def test_defaults_are_the_serialized_material():
    """Assets/Raymarching.mat:66-93"""
    with Material() as m:
        assert m.GetFloat("_MaxSteps") == 400.0
        assert m.GetFloat("_MaxDist") == 50.0
        assert m.GetFloat("_StepFactor") == pytest.approx(0.7)
        assert m.GetFloat("_NormalSampleScale") == pytest.approx(5e-5)
        assert m.GetFloat("_ContactThreshold") == pytest.approx(0.002)
        assert m.GetFloat("_FractalScale") == 2.0
        pos = m.GetMatrix("_Positions")
        np.testing.assert_allclose(pos.GetRow(2)[:3], [-7.87, 0.0, 1.522], rtol=1e-7)
        np.testing.assert_allclose(m.GetVector("_WorldSpaceLightPos0"), [0, .5, 0.8660254, 0], rtol=1e-7)


def test_property_bag_semantics():
    with Material() as m:
        # unknown names are stored silently, like Material.SetFloat on a property the shader lacks
        m.SetFloat("_NoSuchProperty", 3.5)
        assert m.GetFloat("_NoSuchProperty") == 3.5
        assert not m.HasProperty("_NoSuchProperty")
        for n in ("_FrustrumCorners", "_CameraInvViewMatrix", "_Positions", "_Rotations", "_Scales", "_MaxSteps",
                  "_MaxDist", "_StepFactor", "_ContactThreshold", "_NormalSampleScale", "_FractalScale",
                  "_FractalRotationX", "_Reflections", "_RefractiveIndex", "_Glossiness", "_LightPosition",
                  "_Color1", "_Color2", "_WorldSpaceCameraPos", "_WorldSpaceLightPos0", "_ProjectionParams"):
            assert m.HasProperty(n), n
        assert not m.HasProperty("_frustrumcorners")        # case-sensitive, and the reference's spelling
        assert not m.HasProperty("_FrustumCorners")
        m.SetInt("_Reflections", 3)
        assert m.GetInt("_Reflections") == 3 and m.GetFloat("_Reflections") == 3.0
        m.SetVector("_LightPosition", [1, 2, 3])
        assert m.GetVector("_LightPosition").tolist() == [1, 2, 3, 0]
        mm = Matrix4x4.identity()
        mm.SetRow(1, [5, 6, 7])          # Vector3 row -> w = 0
        m.SetMatrix("_Positions", mm)
        back = m.GetMatrix("_Positions")
        assert back.m.tolist() == mm.m.tolist()
        assert back.m[1] == 5 and back.m[5] == 6 and back.m[9] == 7 and back.m[13] == 0    # column-major fields
        with pytest.raises(native.RmError):
            m.GetFloat("_NeverSet")


def test_error_codes_without_gpu_calls():
    lib = native.lib()
    assert lib.rm_create(-1) is None
    assert b"negative" in lib.rm_last_error(None)
    with Material() as m:
        assert lib.rm_set_float(m.handle, None, 1.0) == native.RM_E_ARG
        assert lib.rm_blit(m.handle, None, None, None, None, None, 0, 10) == native.RM_E_ARG
        assert lib.rm_blit(m.handle, None, None, None, None, None, 40000, 10) == native.RM_E_RANGE
        assert b"32768" in lib.rm_last_error(m.handle)
        prim = (native.RmPrimitive * 1)()
        prim[0].type = 9
        assert lib.rm_set_primitives(m.handle, prim, 1) == native.RM_E_ARG
        assert lib.rm_set_primitives(m.handle, prim, native.RM_MAX_PRIMS + 1) == native.RM_E_RANGE
        assert lib.rm_set_kernel(m.handle, 99) == native.RM_E_ARG
        st = native.RmStats()
        assert lib.rm_get_stats(m.handle, C.byref(st)) == native.RM_E_ARG       # nothing rendered yet
    assert lib.rm_set_float(None, b"x", 1.0) == native.RM_E_ARG
    lib.rm_destroy(None)


def test_band_rows():
    lib = native.lib()
    assert lib.rm_band_rows(1080, 8, 0, 1) == 1080
    for n in (2, 3, 4, 8):
        for h, b in ((1080, 8), (2160, 8), (37, 5), (4320, 16), (7, 8)):
            rows = [lib.rm_band_rows(h, b, p, n) for p in range(n)]
            assert sum(rows) == h and min(rows) >= 0
    assert lib.rm_band_rows(10, 0, 0, 1) == native.RM_E_ARG
    assert lib.rm_band_rows(10, 4, 2, 2) == native.RM_E_ARG


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/librmhip.so")
    with pytest.raises(ImportError):
        native.lib()


def test_package_never_touches_the_oracle():
    """the product path must not import, load or reference anything under oracle/"""
    pkg = os.path.join(ROOT, "raymarching_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".cs")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in txt.lower(), f"{fn} mentions the oracle"


def test_cpp_host_mirror_builds_and_runs(tmp_path):
    """include/rmhip.hpp (C++ mirror of Material / Graphics.Blit / the three scripts) against the library"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "cpp_host_mirror")
    libdir = os.path.join(ROOT, "raymarching_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp_host_mirror.cpp"),
                    "-L" + libdir, "-lrmhip", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("ok")


def test_kernels_do_not_spill():
    """Register spills land inside the march loop and have cost up to 2.7x in frame time; the kernels'
    own metadata says how much scratch each variant uses."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    if not os.path.exists(os.path.join(kr.LLVM, "llvm-readelf")):
        pytest.skip("no llvm-readelf")
    res = {r["demangled"]: r for r in kr.kernel_resources(os.path.join(ROOT, "raymarching_amd", "librmhip.so"))}
    assert len(res) >= 43
    for name, r in res.items():
        scratch = int(r["private_segment_fixed_size"])
        if name in ("rm_queue_kernel4<rot=0, scl=0>", "rm_tile_kernel4<rot=0, scl=0>", "rm_tile_kernelF"):
            assert scratch == 0, f"{name} spills {scratch} B"           # the reference scene's kernels and the fractal kernel
        if name.startswith("rm_queue_kernel4_fast"):
            assert scratch == 0, f"{name} spills {scratch} B"
        elif name.startswith(("rm_queue_kernel4", "rm_tile_kernel")):
            # every kernel the automatic choice can pick; the kernels with frag's optional lines carry more state
            assert scratch <= (128 if "_fx" in name else 96), f"{name} spills {scratch} B"
