"""Developer tool: a long run of seeded random tables (tests/cases.py::random_table_case, with and without op bits, at
several world scales) through the automatic kernel choice, each frame compared bit for bit with the CPU oracle.

    python tests/dev/dev_skip_soak.py FIRST_SEED N_SEEDS
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
bad = 0
done = 0
for seed in range(first, first + count):
    for csg in (False, True):
        for ws in (1.0, 3.0, 12.0, 0.3):
            case = cases.random_table_case(seed, ws, csg=csg)
            ref = cases.oracle_run(case, O)
            got = gpu_run(case, torch)
            gn, rn = np.isnan(got["depth"]), np.isnan(ref["depth"])
            same = (np.array_equal(got["hit"], ref["hit"]) and np.array_equal(gn, rn) and
                    np.array_equal(got["depth"].view(np.uint32)[~gn], ref["depth"].view(np.uint32)[~rn]) and
                    np.array_equal(got["rgba"], ref["rgba"]) and
                    all(got["stats"][k] == ref["stats"][k] for k in ("n_hit", "sum_steps_p", "sum_steps_s")))
            done += 1
            if not same:
                bad += 1
                print(f"MISMATCH {case.name} ({len(case.prims)} primitives)", flush=True)
    if (seed - first) % 20 == 19:
        print(f"seed {seed}: {done} frames, {bad} mismatches", flush=True)
print(f"{done} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
