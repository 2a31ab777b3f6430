"""Developer tool: a long run of the general seeded random scenes (tests/cases.py::random_case: tables, the 4-slot scene,
fractal bodies, optional frag lines, domain operators) through the automatic kernel choice and the persistent kernel,
every frame compared bit for bit with the CPU oracle.

    python tests/dev/dev_fuzz_soak.py FIRST_SEED N_SEEDS
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
from test_gpu_parity import gpu_run     # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = done = 0
for seed in range(first, first + count):
    case = cases.random_case(seed)
    ref = cases.oracle_run(case, O)
    for kernel, frames in ((0, 1), (2, 3)):
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
        print(f"seed {seed}: {done} frames, {bad} mismatches", flush=True)
print(f"{done} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
