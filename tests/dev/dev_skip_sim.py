"""CPU simulation: how many leaves of the stress scene's smin tree an evaluation has to visit if a LEFT operand is
skipped whenever the right operand's value (known, per ray) is at least k = .5 below a lower bound of the left operand
over the tile's 64 rays.  smin(a, b) returns b bit for bit when b <= a - k (h clamps to 0), so this culling is exact.
Prints, over sampled 8x8 tiles of the 1080p frame and all primary march steps: the fraction of leaves visited by one
wave walking the whole table, and by the slowest of four waves that each walk a quarter (what the shared-tile kernel
would wait for)."""
import math
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import numpy_restatement as R            # noqa: E402
from raymarching_amd import scenes                    # noqa: E402
from tests import cases                                # noqa: E402

f = np.float32
MIN_BLOCK = int(os.environ.get("MIN_BLOCK", "1"))
STATIC = os.environ.get("STATIC", "0") == "1"


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    w, h = 1920, 1080
    prims = scenes.stress_primitives(256)
    case = next(c for c in cases.CASES if c.name == "stress256_1920x1080")
    from raymarching_amd import Material
    from oracle import rm_oracle as O
    with Material() as mat:
        case.bind(mat)
        props = O.props_from_material(mat)
    sc = R.Scene(props, prims)
    dx, dy, dz = (a.reshape(h, w) for a in R.pixel_rays(props, w, h))
    cam = np.asarray(props["_WorldSpaceCameraPos"], dtype=f)
    ox, oy, oz = (np.full((h, w), cam[j], dtype=f) for j in range(3))
    far = f(np.asarray(props["_ProjectionParams"], dtype=f)[2])
    depth = np.full((h, w), np.fmin(far, f(props["_MaxDist"])), dtype=f)
    n = len(prims)
    C = np.array([p["pos"] for p in prims], dtype=np.float64)
    rl = []
    for p in prims:
        a = p["params"]
        t = p["type"]
        rl.append([abs(a[0]) * .5, .5 * math.sqrt(a[0] ** 2 + a[1] ** 2 + a[2] ** 2), abs(a[0]) + abs(a[1]) + abs(a[2]),
                   math.sqrt(a[0] ** 2 + a[1] ** 2)][t])
    S = np.array([p["scale"] for p in prims], dtype=np.float64)
    Rad = np.array(rl) * S.max(1)
    ratio = S.max(1) / S.min(1)
    rng = np.random.default_rng(1)
    steps, sf = int(props["_MaxSteps"]), f(props["_StepFactor"])
    tot_evals = tot_one = tot_coop = 0
    fine = {}
    bal = {}
    for t in range(n_tiles):
        tx, ty = int(rng.integers(0, w // 8)), int(rng.integers(0, h // 8))
        sl = (slice(ty * 8, ty * 8 + 8), slice(tx * 8, tx * 8 + 8))
        o = [a[sl].ravel() for a in (ox, oy, oz)]
        d = [a[sl].ravel() for a in (dx, dy, dz)]
        dep = depth[sl].ravel()
        dm = np.zeros(64, dtype=f)
        active = np.ones(64, dtype=bool)
        for i in range(steps):
            if not active.any():
                break
            px, py, pz = (o[j] + d[j] * dm for j in range(3))
            D = np.stack([sc.prim(p, px, py, pz) for p in sc.prims])          # [n, 64]
            a = active
            q = np.stack([px[a], py[a], pz[a]], 1).astype(np.float64)
            lo, hi = q.min(0), q.max(0)
            ctr, hd = .5 * (lo + hi), .5 * np.linalg.norm(hi - lo)
            if os.environ.get("LB", "centre") == "sphere":
                LB = (np.linalg.norm(C - ctr, axis=1) - hd - Rad) / ratio - 0.01
            else:       # every leaf is 1-Lipschitz: its value at the box centre minus the half diagonal
                cc = [np.full(1, v, dtype=f) for v in ctr]
                LB = np.array([float(sc.prim(p, cc[0], cc[1], cc[2])[0]) for p in sc.prims]) - hd - 0.01

            # optional: the marching lanes in clusters (rows of the tile), one box each; a block is skipped only if every
            # cluster allows it (the union of the clusters' masks is still exact)
            NCL = int(os.environ.get("CLUSTERS", "1"))
            if NCL > 1:
                lanes = np.nonzero(a)[0]
                cl = [lanes[(lanes // 8) * NCL // 8 == c] for c in range(NCL)]
                LBs, HDs = [], []
                for L_ in cl:
                    if len(L_) == 0:
                        continue
                    qq = np.stack([px[L_], py[L_], pz[L_]], 1).astype(np.float64)
                    lo_, hi_ = qq.min(0), qq.max(0)
                    c_, h_ = .5 * (lo_ + hi_), .5 * np.linalg.norm(hi_ - lo_)
                    cc = [np.full(1, v, dtype=f) for v in c_]
                    LBs.append(np.array([float(sc.prim(p, cc[0], cc[1], cc[2])[0]) for p in sc.prims]) - h_ - 0.01)
                    HDs.append(h_)
            visited = [0]

            def ev(lo_i, hi_i):
                if hi_i - lo_i == 1:
                    visited[0] += 1
                    vis_idx.append(lo_i)
                    return D[lo_i][a]
                mid = (lo_i + hi_i) // 2
                vb = ev(mid, hi_i)
                lvl = int(math.log2(mid - lo_i))
                lbl = LB[lo_i:mid].min() - 0.126 * lvl
                if STATIC and NCL > 1:
                    ok = all((lb_[mid:hi_i].min() + 2 * h_ + 0.02) + 0.5 <= lb_[lo_i:mid].min() - 0.126 * lvl for lb_, h_ in zip(LBs, HDs))
                elif STATIC:
                    ok = (LB[mid:hi_i].min() + 2 * hd + 0.02) + 0.5 <= lbl          # upper bound of the right operand known beforehand
                else:
                    ok = np.all(vb + f(0.5) <= lbl)
                if (mid - lo_i) >= MIN_BLOCK and ok:
                    return vb
                va = ev(lo_i, mid)
                return R.smin_w(va, vb)[0]
            per_chunk = []
            vals = []
            vis_idx = []
            for c in range(4):
                visited[0] = 0
                vals.append(ev(64 * c, 64 * c + 64))
                per_chunk.append(visited[0])
            # blocks of B leaves dealt to the four waves as contiguous runs of about a quarter of the visited leaves each
            for B in (16, 8):
                cnt = np.bincount(np.array(vis_idx) // B, minlength=256 // B).astype(np.float64)
                T = cnt.sum()
                mid = np.cumsum(cnt) - cnt / 2
                owner = np.minimum((mid * 4 / T).astype(int), 3)
                bal[B] = bal.get(B, 0) + max(cnt[owner == wv].sum() for wv in range(4))
            vis_idx = []
            for S in (32, 16):                      # finer subtrees dealt round robin to the four waves
                per_sub = []
                for c in range(256 // S):
                    visited[0] = 0
                    ev(S * c, S * c + S)
                    per_sub.append(visited[0])
                fine[S] = fine.get(S, 0) + max(sum(per_sub[wv::4]) for wv in range(4))
            v = R.smin_w(R.smin_w(vals[0], vals[1])[0], R.smin_w(vals[2], vals[3])[0])[0]
            _, ref = sc.eval(px[a], py[a], pz[a])
            assert np.array_equal(v.view(np.uint32), ref.view(np.uint32)), "culled value differs"
            tot_evals += 1
            tot_one += sum(per_chunk)
            tot_coop += max(per_chunk)
            full = np.zeros(64, dtype=f)
            full[a] = v
            adts = np.abs(full)
            dm = np.where(a, dm + adts * sf, dm)
            stop = a & ((adts < dm * f(.0025)) | (dm > dep))
            active = a & ~stop
        print(f"tile {t} ({tx},{ty}): evals so far {tot_evals}, one wave visits {tot_one / (tot_evals * 256):.3f} of the table, "
              f"slowest quarter {tot_coop / (tot_evals * 64):.3f}, with subtrees of 32 / 16 dealt round robin "
              f"{fine[32] / (tot_evals * 64):.3f} / {fine[16] / (tot_evals * 64):.3f}; blocks of 16 / 8 dealt as contiguous quarters "
              f"{bal[16] / (tot_evals * 64):.3f} / {bal[8] / (tot_evals * 64):.3f}", flush=True)


if __name__ == "__main__":
    main()
