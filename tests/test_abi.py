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
    """Every default rm_create() sets, against literals copied from the reference's assets (cited per line).  The parity
    tests feed the oracle from the product's own property bag, so a wrong default would be shared by both sides: this is
    the test that would catch it."""
    f32 = np.float32
    with Material() as m:
        # Assets/Raymarching.mat m_Floats
        mat_floats = {"_ContactThreshold": 0.002,      # Raymarching.mat:66
                      "_FractalRotationX": 0.19,       # :71
                      "_FractalRotationY": -0.26,      # :72
                      "_FractalRotationZ": 0.14,       # :73
                      "_FractalScale": 2.0,            # :74
                      "_Glossiness": 0.2,              # :76
                      "_MaxDist": 50.0,                # :78
                      "_MaxSteps": 400.0,              # :79
                      "_NormalSampleScale": 0.00005,   # :82
                      "_Reflections": 3.12,            # :85
                      "_RefractiveIndex": 1.0,         # :86
                      "_StepFactor": 0.7}              # :93
        for k, v in mat_floats.items():
            assert m.GetFloat(k) == float(f32(v)), k
        # Assets/Raymarching.mat m_Colors
        assert m.GetVector("_Color1").tolist() == [1, 1, 1, 1]                            # Raymarching.mat:99
        assert m.GetVector("_Color2").tolist() == [0.75, 0.5, 1, 1]                       # :100
        assert m.GetVector("_LightPosition").tolist() == [0, 0, 0, 0]                     # :102
        # Assets/Scenes/SampleScene.unity: the four "Position N" transforms in the order of the script's array (:242-246)
        pos, rot, scl = m.GetMatrix("_Positions"), m.GetMatrix("_Rotations"), m.GetMatrix("_Scales")
        want_pos = [(-0.584, 0.049, 0.602),      # Position 0, SampleScene.unity:360
                    (-0.74, 0.929, 0.0),         # Position 1, :489
                    (-7.87, 0.0, 1.522),         # Position 2, :390
                    (-3.074, -0.087, 0.08)]      # Position 3, :139
        for i in range(4):
            assert pos.GetRow(i).tolist() == [float(f32(v)) for v in want_pos[i]] + [0.0], i   # Vector3 row -> w = 0
            assert rot.GetRow(i).tolist() == [0, 0, 0, 1], i                              # :138, :359, :389, :488
            assert scl.GetRow(i).tolist() == [1, 1, 1, 0], i                              # :140, :361, :391, :490
        # Main Camera: position (2, 1, -11.01) (:222), identity rotation (:221), fov 80 (:196), near .3 / far 1000 (:194-195)
        assert m.GetVector("_WorldSpaceCameraPos").tolist() == [2.0, 1.0, float(f32(-11.01)), 0.0]
        assert m.GetVector("_ProjectionParams").tolist() == [1.0, float(f32(0.3)), 1000.0, float(f32(1.0) / f32(1000.0))]
        c2w = m.GetMatrix("_CameraInvViewMatrix")               # TRS(pos, identity, 1) with the Z column negated
        assert c2w.GetRow(0).tolist() == [1, 0, 0, 2.0] and c2w.GetRow(1).tolist() == [0, 1, 0, 1.0]
        assert c2w.GetRow(2).tolist() == [0, 0, -1, float(f32(-11.01))] and c2w.GetRow(3).tolist() == [0, 0, 0, 1]
        t = f32(np.tan(np.float64(f32(80.0) * f32(0.5) * f32(0.0174532924))))      # Mathf.Tan(fov/2 * Deg2Rad), Raymarching.cs:41
        a = t * f32(16.0 / 9.0)
        fc = m.GetMatrix("_FrustrumCorners")                     # rows TL, TR, BR, BL (Raymarching.cs:45-53)
        assert fc.GetRow(0).tolist() == [-float(a), float(t), -1, 0] and fc.GetRow(1).tolist() == [float(a), float(t), -1, 0]
        assert fc.GetRow(2).tolist() == [float(a), -float(t), -1, 0] and fc.GetRow(3).tolist() == [-float(a), -float(t), -1, 0]
        assert m.GetVector("_BackgroundColor").tolist() == [float(f32(0.19215687)), float(f32(0.3019608)), float(f32(0.4745098)), 1.0]   # :182
        # Directional Light rotation (0.9659258, 0, 0, 0.2588191) (:458) = 150 degrees about X: forward (0, -.5, -.8660254);
        # _WorldSpaceLightPos0 points towards the light
        np.testing.assert_allclose(m.GetVector("_WorldSpaceLightPos0"), [0, .5, 0.8660254, 0], rtol=1e-7)
        q = np.array([0.9659258, 0, 0, 0.2588191], dtype=np.float64)
        fwd = np.array([2 * (q[0] * q[2] + q[3] * q[1]), 2 * (q[1] * q[2] - q[3] * q[0]), 1 - 2 * (q[0] ** 2 + q[1] ** 2)])
        np.testing.assert_allclose(-fwd, m.GetVector("_WorldSpaceLightPos0")[:3], atol=2e-7)
        # switches of this build: all off = the shader as it stands
        for k in ("_RmSceneBody", "_RmFractalPower", "_RmStepHeat", "_RmDebugView", "_RmVolumetricLight", "_RmBounce", "_RmDomainOp",
                  "_RmShadowMode", "_RmLightModel", "_RmScatter"):
            assert m.GetFloat(k) == 0.0, k


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
        for k in range(6):                                   # 0 automatic, 1..3 and 5 exact kernels, 4 the tolerance kernel
            assert lib.rm_set_kernel(m.handle, k) == native.RM_OK
        assert lib.rm_set_kernel(m.handle, 6) == native.RM_E_ARG and lib.rm_set_kernel(m.handle, -1) == native.RM_E_ARG
        # the transport entry points check their arguments before they touch the GPU
        buf = C.create_string_buffer(256)
        p = C.cast(buf, C.c_void_p)
        assert lib.rm_compact_frames_device(m.handle, None, p, 8, 8, 1, p, p, None) == native.RM_E_ARG
        assert lib.rm_compact_frames_device(m.handle, p, p, 8, 8, 0, p, p, None) == native.RM_E_ARG
        assert lib.rm_expand_frames_device(m.handle, None, 3, None, p, 8, 8, 1, None) == native.RM_E_ARG      # records missing
        assert lib.rm_expand_frames_device(m.handle, p, 1, None, None, 8, 8, 1, None) == native.RM_E_ARG      # no destination
        assert lib.rm_update_frames_device(m.handle, None, 2, p, 1, None, p, 8, 8, 1, None) == native.RM_E_ARG
        assert lib.rm_update_frames_device(m.handle, p, 1, None, 2, None, p, 8, 8, 1, None) == native.RM_E_ARG
        assert lib.rm_update_frames_device(m.handle, p, 1, p, 1, None, p, 0, 8, 1, None) == native.RM_E_ARG
        assert lib.rm_update_frames_device(m.handle, p, 1, p, 1, None, p, 40000, 8, 1, None) == native.RM_E_RANGE
        odd = C.c_void_p(p.value + 1)
        assert lib.rm_update_frames_device(m.handle, p, 1, p, 1, None, odd, 8, 8, 1, None) == native.RM_E_ARG  # alignment
        assert lib.rm_tile_record_bytes() == 196
        # frames whose miss pixels are not the background cannot travel as hit tiles only
        for prop in ("_RmStepHeat", "_RmScatter", "_RmVolumetricLight"):
            m.SetFloat(prop, 1.0)
            assert lib.rm_compact_frames_device(m.handle, p, p, 8, 8, 1, p, p, None) == native.RM_E_ARG, prop
            assert b"miss pixels" in lib.rm_last_error(m.handle)
            m.SetFloat(prop, 0.0)
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
    assert len(res) >= 50
    for name, r in res.items():
        scratch = int(r["private_segment_fixed_size"])
        # the reference scene's kernels (the instantiations for the plain launch shapes: whole frames, pose batches, frames
        # with source image and depth texture) and the fractal kernel
        if name in ("rm_queue_kernel4_plain<rot=0, scl=0, 1>", "rm_queue_kernel4_plain<rot=0, scl=0, 2>",
                    "rm_queue_kernel4_plain<rot=0, scl=0, 3>", "rm_tile_kernel4<rot=0, scl=0>", "rm_tile_kernelF"):
            assert scratch == 0, f"{name} spills {scratch} B"
        if name.startswith("rm_queue_kernel4_fast") and not name.endswith(", 0>"):
            assert scratch == 0, f"{name} spills {scratch} B"               # (", 0>": the instantiation for any launch shape)"
        elif name.startswith(("rm_queue_kernel4", "rm_tile_kernel", "rm_coop_kernel")):
            # every kernel the automatic choice can pick; the kernels with frag's optional lines carry more state
            assert scratch <= 128, f"{name} spills {scratch} B"


def test_no_scratch_access_inside_the_march_loops():
    """the bytes above are tolerable only outside the loops: disassembles the 4-slot persistent kernels whose metadata shows
    scratch and checks that no innermost loop of theirs (the two marches, the normal taps) touches it"""
    import importlib.util
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    spec = importlib.util.spec_from_file_location("loop_isa", os.path.join(ROOT, "scripts", "loop_isa.py"))
    li = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(li)
    if not os.path.exists(os.path.join(li.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    import tempfile
    lib = os.path.join(ROOT, "raymarching_amd", "librmhip.so")
    for kernel in ("rm_queue_kernel4<rot=1, scl=0>", "rm_queue_kernel4<rot=0, scl=1>", "rm_queue_kernel4_plain<rot=0, scl=0, 1>"):
        with tempfile.TemporaryDirectory() as d:
            co, sym = li.find_kernel(lib, kernel, d)
            assert co is not None, kernel
            ins, labels = li.disassemble(co, sym)
        big = [(a, b) for a, b in li.loops(ins, labels) if sum(1 for m, _, _ in ins[a:b + 1] if m.startswith("v_")) >= 40]
        assert len(big) >= 3, f"{kernel}: expected the two march loops and the tap loop, found {len(big)} loops"
        for a, b in big:
            bad = [m for m, _, _ in ins[a:b + 1] if m.startswith("scratch_")]
            assert not bad, f"{kernel}: {len(bad)} scratch accesses inside the loop at {ins[a][2]}"
