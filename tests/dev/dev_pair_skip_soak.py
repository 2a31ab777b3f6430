"""Developer tool: seeded random 4-slot scenes WITHOUT per-primitive rotation (the kernel variants whose soft-shadow march skips
the sphere / box pair): random positions, scales (unit, power-of-two, arbitrary), light directions of any length, camera
distances, step factors; through the automatic choice, the one-wave tile kernel and the persistent kernel (three frames, so
that the heavy-tile history is in play), every frame compared bit for bit with the CPU oracle.

    python tests/dev/dev_pair_skip_soak.py FIRST_SEED N_SEEDS
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import cases                            # noqa: E402
from oracle import rm_oracle as O       # noqa: E402
from raymarching_amd import scenes      # noqa: E402
from test_gpu_parity import gpu_run     # noqa: E402

F = np.float32


def make_case(seed):
    rng = np.random.default_rng(0x9A1B0000 + seed)
    kind = int(rng.integers(0, 3))
    w, h = [(96, 54), (128, 72), (160, 90)][int(rng.integers(0, 3))]
    off = rng.uniform(-1.0, 1.0, 3) * [0.0, 5.0, 400.0][int(rng.integers(0, 3))]

    def setup(case, mat):
        ts = scenes.default_transforms()
        for i, t in enumerate(ts):
            t.position = (np.asarray(t.position, dtype=F) + rng_pos[i] + off).astype(F)
            if kind == 1:
                t.lossyScale = np.asarray(2.0 ** rng_pow[i], dtype=F)
            elif kind == 2:
                t.lossyScale = np.asarray(rng_scl[i], dtype=F)
        cam = scenes.default_camera(case.w, case.h)
        cam.transform.position = (np.asarray(cam.transform.position, dtype=F) * F(cam_dist) + off).astype(F)
        cases._base(mat, case.w, case.h, camera=cam, transforms=ts)
        mat.SetVector("_WorldSpaceLightPos0", np.asarray(light, dtype=F))
        mat.SetFloat("_StepFactor", float(sf))
        mat.SetFloat("_MaxSteps", float(steps))

    rng_pos = rng.uniform(-1.5, 1.5, (4, 3))
    rng_pow = rng.integers(-1, 2, (4, 3)).astype(np.float64)
    rng_scl = rng.uniform(0.5, 2.2, (4, 3))
    cam_dist = rng.uniform(0.6, 1.6)
    d = rng.normal(size=3)
    d[1] = abs(d[1]) * rng.uniform(0.05, 1.0)            # mostly low lights: long shadow marches
    light = np.append(d / np.linalg.norm(d) * rng.choice([1.0, 1.0, 0.4, 2.7]), 0.0)
    sf = rng.choice([0.7, 0.7, 0.35, 0.95])
    steps = rng.choice([400.0, 150.0, 800.0])
    return cases.Case(f"pair_skip_soak_{seed}_{w}x{h}", w, h, setup=setup)


first, count = int(sys.argv[1]), int(sys.argv[2])
bad = done = 0
for seed in range(first, first + count):
    case = make_case(seed)
    ref = cases.oracle_run(case, O)
    for kernel, frames in ((0, 1), (1, 1), (2, 3)):
        got = gpu_run(case, torch, kernel=kernel, frames=frames)
        gn, rn = np.isnan(got["depth"]), np.isnan(ref["depth"])
        same = (np.array_equal(got["hit"], ref["hit"]) and np.array_equal(gn, rn) and
                np.array_equal(got["depth"].view(np.uint32)[~gn], ref["depth"].view(np.uint32)[~rn]) and
                np.array_equal(got["rgba"], ref["rgba"]) and
                all(got["stats"][k] == ref["stats"][k] for k in ("n_hit", "sum_steps_p", "sum_steps_s")))
        done += 1
        if not same:
            bad += 1
            print(f"MISMATCH {case.name} kernel {kernel}", flush=True)
    if (seed - first) % 50 == 49:
        print(f"seed {seed}: {done} frames, {bad} mismatches, last n_hit {ref['stats']['n_hit']} shadow steps {ref['stats']['sum_steps_s']}", flush=True)
print(f"{done} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
