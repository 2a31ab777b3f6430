"""Generates tests/golden/golden.json and tests/golden/subsamples.npz from the CPU
oracle (oracle/rm_oracle.c) in this repository.

The reference holds no golden image, capture or test vector (it only runs as a
Unity fragment shader), so these fixtures record the oracle's own output after it
was cross-checked against the independent NumPy restatement
(oracle/numpy_restatement.py).  They guard against regressions of the oracle and
let the GPU tests check full frames by hash.

    python tests/golden/make_golden.py                 # every case
    python tests/golden/make_golden.py NAME [NAME ...] # only these, merged into the existing files
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import rm_oracle as O          # noqa: E402
import cases                                # noqa: E402
from helpers import sha                     # noqa: E402

STEP_BINS = [0, 1, 2, 5, 10, 20, 40, 80, 160, 320, 640]


def main():
    meta, arrays = {}, {}
    only = sys.argv[1:]
    if only:
        meta = json.load(open(os.path.join(HERE, "golden.json")))["cases"]
        with np.load(os.path.join(HERE, "subsamples.npz")) as z:
            arrays = {k: z[k] for k in z.files}
    for c in cases.CASES:
        if only and c.name not in only:
            continue
        out = O_run(c)
        e = {"w": c.w, "h": c.h, "kind": c.kind, "sha_depth": sha(out["depth"]), "sha_hit": sha(out["hit"]),
             "n_hit": int(out["hit"].sum()), "sum_steps_p": int(out["stats"]["sum_steps_p"])}
        sy, sx = max(1, c.h // 36), max(1, c.w // 64)
        arrays[c.name + "/depth"] = out["depth"][::sy, ::sx].copy()
        arrays[c.name + "/hit"] = out["hit"][::sy, ::sx].copy()
        if c.kind == "frag":
            e["sha_rgba"] = sha(out["rgba"])
            e["sum_steps_s"] = int(out["stats"]["sum_steps_s"])
            e["hist_steps_p"] = np.histogram(out["steps_p"], bins=STEP_BINS)[0].tolist()
            e["hist_steps_s"] = np.histogram(out["steps_s"][out["hit"] > 0], bins=STEP_BINS)[0].tolist()
            arrays[c.name + "/rgba"] = out["rgba"][::sy, ::sx].copy()
        e["subsample_stride"] = [sy, sx]
        meta[c.name] = e
        print(c.name, e["n_hit"], e["sum_steps_p"])
    with open(os.path.join(HERE, "golden.json"), "w") as fjs:
        json.dump({"step_bins": STEP_BINS, "cases": meta}, fjs, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "subsamples.npz"), **arrays)


def O_run(c):
    return cases.oracle_run(c, O)


if __name__ == "__main__":
    main()
