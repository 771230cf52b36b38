#!/usr/bin/env python3
"""Offline experiment: SIMT cost of closest-hit strategies on REAL ray segments.

Logs every segment of an oracle render (orc_set_raylog), groups the segments the way
the device kernel's lanes would see them (64 lanes per wave, one segment per lane per
trip, a lane walks its pixel's samples), and counts, per wave trip:
  stream   : N_prim tests                                  (the v1 kernel)
  packet   : wave-uniform BVH descent — a node is entered when ANY lane's slab test
             passes (node data is wave-uniform → scalar loads, no divergence)
  per-lane : every lane walks the BVH itself; SIMT cost = max over lanes
Test-side tooling (it uses the oracle, so it lives under tests/); it only informs DESIGN.md.
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
import orc  # noqa: E402
import rtow  # noqa: E402


def log_rays(scene, cfg, cap=4_000_000):
    buf = np.zeros((cap, 12))
    L = orc.lib()
    L.orc_set_raylog.argtypes = [C.POINTER(C.c_double), C.c_uint64]
    L.orc_set_raylog.restype = None
    L.orc_raylog_count.restype = C.c_uint64
    L.orc_set_raylog(buf.ctypes.data_as(C.POINTER(C.c_double)), cap)
    orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=1)
    n = L.orc_raylog_count()
    L.orc_set_raylog(None, 0)
    return buf[:n].copy()


def build_bvh(bmin, bmax, leaf_max=4, bins=16):
    """binned-SAH BVH2 in DFS order. returns dict of arrays."""
    n = len(bmin)
    cen = 0.5 * (bmin + bmax)
    nodes = []  # [lo(3), hi(3), left, right, first, count]
    order = []

    def area(lo, hi):
        d = np.maximum(hi - lo, 0)
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]

    def rec(idx):
        me = len(nodes)
        lo, hi = bmin[idx].min(0), bmax[idx].max(0)
        nodes.append(None)
        if len(idx) <= 1:
            nodes[me] = (lo, hi, -1, -1, len(order), len(idx))
            order.extend(idx)
            return me
        best = (np.inf, None, None)
        c = cen[idx]
        clo, chi = c.min(0), c.max(0)
        for ax in range(3):
            if chi[ax] <= clo[ax]:
                continue
            b = np.clip(((c[:, ax] - clo[ax]) * bins / (chi[ax] - clo[ax])).astype(int), 0, bins - 1)
            for k in range(bins - 1):
                lm = b <= k
                nl = lm.sum()
                if nl == 0 or nl == len(idx):
                    continue
                il, ir = idx[lm], idx[~lm]
                cost = area(bmin[il].min(0), bmax[il].max(0)) * nl + \
                    area(bmin[ir].min(0), bmax[ir].max(0)) * (len(idx) - nl)
                if cost < best[0]:
                    best = (cost, il, ir)
        if best[1] is None or (len(idx) <= leaf_max and best[0] >= area(lo, hi) * len(idx)):
            if best[1] is None and len(idx) > leaf_max:
                h = len(idx) // 2
                best = (0, idx[:h], idx[h:])
            else:
                nodes[me] = (lo, hi, -1, -1, len(order), len(idx))
                order.extend(idx)
                return me
        l = rec(best[1])
        r = rec(best[2])
        nodes[me] = (lo, hi, l, r, 0, 0)
        return me

    sys.setrecursionlimit(10000)
    rec(np.arange(n))
    N = len(nodes)
    out = dict(lo=np.array([x[0] for x in nodes]), hi=np.array([x[1] for x in nodes]),
               left=np.array([x[2] for x in nodes]), right=np.array([x[3] for x in nodes]),
               first=np.array([x[4] for x in nodes]), count=np.array([x[5] for x in nodes]),
               order=np.array(order))
    # skip links for threaded traversal: next node in DFS order when the subtree is skipped
    skip = np.full(N, -1)

    def setskip(i, nxt):
        skip[i] = nxt
        if out["left"][i] >= 0:
            setskip(out["left"][i], out["right"][i])
            setskip(out["right"][i], nxt)

    setskip(0, -1)
    out["skip"] = skip
    return out


def slab(lo, hi, o, inv, tmin, tmax):
    t0 = (lo - o) * inv
    t1 = (hi - o) * inv
    near = np.minimum(t0, t1).max(-1)
    far = np.maximum(t0, t1).min(-1)
    return np.maximum(near, tmin) <= np.minimum(far, tmax)


def sphere_t(c, r2, o, d, a, tmax):
    oc = o - c
    h = (oc * d).sum(-1)
    cc = (oc * oc).sum(-1) - r2
    disc = h * h - a * cc
    ok = disc >= 0
    sq = np.sqrt(np.where(ok, disc, 0))
    r1 = (-h - sq) / a
    r2_ = (-h + sq) / a
    t = np.where((r1 >= 1e-3) & (r1 <= tmax), r1, np.where((r2_ >= 1e-3) & (r2_ <= tmax), r2_, np.inf))
    return np.where(ok, t, np.inf)


def tri_t(A, B, Cc, o, d, tmax):
    e1, e2 = B - A, Cc - A
    n = np.cross(e1, e2)
    det = -(d * n).sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        ao = o - A
        dao = np.cross(ao, d)
        u = (e2 * dao).sum(-1) * inv
        v = -(e1 * dao).sum(-1) * inv
        t = (ao * n).sum(-1) * inv
    ok = (det >= 1e-6) & (t >= 1e-3) & (t <= tmax) & (u >= 0) & (v >= 0) & (u + v <= 1)
    return np.where(ok, t, np.inf)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cover"
    if which == "cover":
        W, H = 1200, 800
        scene = orc.OrcScene.cover(11, 1.5, False)
        s = scene.c
        sph = np.ctypeslib.as_array(s.sphere_geom, shape=(s.n_spheres, 4)).copy()
        cfg = rtow.make_config(W, H, 10, 1, 50, seed=1, rank=0, nranks=25, tile_rows=8)
        c, r = sph[:, :3], np.abs(sph[:, 3])
        bmin, bmax = c - r[:, None], c + r[:, None]
        nprims, per_test = len(sph), 13
        ptest = lambda p, o, d, a, tmax: sphere_t(sph[p, :3], sph[p, 3] ** 2, o, d, a, tmax)
        stride = 25
    else:
        W, H = 1920, 1080
        scene = orc.OrcScene.obj(ROOT / "tests/golden/suzanne.obj", 16 / 9)
        s = scene.c
        tri = np.ctypeslib.as_array(s.triangle_geom, shape=(s.n_triangles, 3, 3)).copy()
        cfg = rtow.make_config(W, H, 6, 1, 20, seed=1, rank=0, nranks=45, tile_rows=8)
        bmin, bmax = tri.min(1), tri.max(1)
        nprims, per_test = len(tri), 45
        ptest = lambda p, o, d, a, tmax: tri_t(tri[p, 0], tri[p, 1], tri[p, 2], o, d, tmax)
        stride = 45
    rays = log_rays(scene, cfg)
    print("segments logged:", len(rays))
    rows = [i for i in range(H) if (i // 8) % stride == 0]
    bvh = build_bvh(bmin, bmax)
    print("bvh nodes", len(bvh["lo"]), "leaves", (bvh["count"] > 0).sum(), "max leaf", bvh["count"].max())

    # lanes: lane <-> pixel, processes its samples' segments in order
    pix = rays[:, 0].astype(np.int64)
    order = np.lexsort((rays[:, 2], rays[:, 1], pix))
    rays = rays[order]
    pix = rays[:, 0].astype(np.int64)
    upix, start = np.unique(pix, return_index=True)
    seglist = {p: (st, en) for p, st, en in zip(upix, start, list(start[1:]) + [len(rays)])}

    def waves(tile):
        ws = []
        if tile:  # 8x8 pixel tiles
            for r0 in range(0, len(rows), 8):
                for c0 in range(0, W, 8):
                    ws.append([rows[r0 + i] * W + c0 + j for i in range(8) for j in range(8)])
        else:  # 64 consecutive pixels of a row
            for rr in rows:
                for c0 in range(0, W - 63, 64):
                    ws.append([rr * W + c0 + j for j in range(64)])
        return ws

    rng = np.random.default_rng(0)
    for tile in (False, True):
        ws = waves(tile)
        pick = rng.choice(len(ws), size=25, replace=False)
        stats = []
        ord_stats = []
        for wi in pick:
            lanes = [seglist[p] for p in ws[wi] if p in seglist]
            ntrips = min(en - st for st, en in lanes)
            for trip in rng.choice(ntrips, size=min(6, ntrips), replace=False):
                idx = np.array([st + trip for st, en in lanes])
                R = rays[idx]
                o, d = R[:, 3:6], R[:, 6:9]
                a = (d * d).sum(-1)
                with np.errstate(divide="ignore"):
                    inv = 1.0 / d
                nl = len(idx)
                # --- packet traversal (ordered by node index = DFS, near child unknown) ---
                best = np.full(nl, np.inf)
                pv_nodes = pv_prims = 0
                stack = [0]
                while stack:
                    n = stack.pop()
                    pv_nodes += 1
                    hit = slab(bvh["lo"][n], bvh["hi"][n], o, inv, 1e-3, best)
                    if not hit.any():
                        continue
                    if bvh["count"][n] > 0:
                        for k in range(bvh["first"][n], bvh["first"][n] + bvh["count"][n]):
                            p = bvh["order"][k]
                            pv_prims += 1
                            t = ptest(p, o, d, a, best)
                            best = np.minimum(best, t)
                    else:
                        # visit the child nearer to the mean ray origin first
                        l, r_ = bvh["left"][n], bvh["right"][n]
                        cl = 0.5 * (bvh["lo"][l] + bvh["hi"][l])
                        cr = 0.5 * (bvh["lo"][r_] + bvh["hi"][r_])
                        mo = o.mean(0)
                        if ((cl - mo) ** 2).sum() <= ((cr - mo) ** 2).sum():
                            stack += [r_, l]
                        else:
                            stack += [l, r_]
                # --- per-lane threaded traversal in lockstep ---
                node = np.zeros(nl, dtype=np.int64)
                bestl = np.full(nl, np.inf)
                steps = np.zeros(nl, dtype=np.int64)
                ptests = np.zeros(nl, dtype=np.int64)
                leaves = np.zeros(nl, dtype=np.int64)
                trips_lock = 0
                while (node >= 0).any():
                    act = node >= 0
                    nn = np.where(act, node, 0)
                    hit = slab(bvh["lo"][nn], bvh["hi"][nn], o, inv, 1e-3, bestl) & act
                    steps += act
                    trips_lock += 1
                    isleaf = bvh["count"][nn] > 0
                    for li in np.nonzero(hit & isleaf)[0]:
                        n = nn[li]
                        leaves[li] += 1
                        for k in range(bvh["first"][n], bvh["first"][n] + bvh["count"][n]):
                            p = bvh["order"][k]
                            ptests[li] += 1
                            t = float(ptest(p, o[li:li+1], d[li:li+1], a[li:li+1], bestl[li:li+1])[0])
                            bestl[li] = min(bestl[li], t)
                    nxt = np.where(hit & ~isleaf, nn + 1, bvh["skip"][nn])
                    node = np.where(act, nxt, -1)
                assert np.allclose(np.where(np.isinf(best), 1e30, best), np.where(np.isinf(bestl), 1e30, bestl))
                # --- per-lane ORDERED traversal (visit = both children tested, near child first, stack) ---
                ov = np.zeros(nl, dtype=np.int64); op = np.zeros(nl, dtype=np.int64)
                for li in range(nl):
                    bo = np.inf
                    stk = [0]
                    oo, dd, ii, aa = o[li], d[li], inv[li], a[li]
                    while stk:
                        n = stk.pop()
                        if bvh["count"][n] > 0:
                            for k in range(bvh["first"][n], bvh["first"][n] + bvh["count"][n]):
                                pth = bvh["order"][k]
                                op[li] += 1
                                t = float(ptest(pth, oo[None], dd[None], np.array([aa]), np.array([bo]))[0])
                                bo = min(bo, t)
                            continue
                        ov[li] += 1
                        l, r_ = bvh["left"][n], bvh["right"][n]
                        res = []
                        for c in (l, r_):
                            t0 = (bvh["lo"][c] - oo) * ii; t1 = (bvh["hi"][c] - oo) * ii
                            near = max(np.minimum(t0, t1).max(), 1e-3); far = min(np.maximum(t0, t1).min(), bo)
                            if near <= far:
                                res.append((near, c))
                        res.sort()
                        for near, c in reversed(res):
                            stk.append(c)
                ord_stats.append((ov.mean(), ov.max(), op.mean(), op.max()))
                nprim = (R[:, 2] == 0).sum()
                stats.append((pv_nodes, pv_prims, steps.mean(), steps.max(), ptests.mean(), ptests.max(),
                              leaves.mean(), leaves.max(), nprim / nl))
        S = np.array(stats)
        names = ["packet nodes", "packet prims", "lane steps mean", "lane steps max", "lane prims mean",
                 "lane prims max", "lane leaves mean", "lane leaves max", "primary frac"]
        print(f"--- lanes<->{'8x8 tile' if tile else '64 px of a row'}; {len(S)} wave trips")
        for k, nm in enumerate(names):
            print(f"  {nm:18s} mean {S[:, k].mean():8.2f}  p50 {np.median(S[:, k]):8.2f}  p90 {np.quantile(S[:, k], .9):8.2f}")
        # instruction-count model (f64 VALU instr per wave trip)
        stream = nprims * per_test
        packet = S[:, 0] * 19 + S[:, 1] * per_test
        lane = S[:, 3] * 25 + S[:, 7] * (S[:, 5] / np.maximum(S[:, 7], 1)) * per_test
        print(f"  model VALU instr/trip: stream {stream}, packet {packet.mean():.0f}, per-lane(lockstep) {lane.mean():.0f}")
        O = np.array(ord_stats)
        print(f"  ORDERED per-lane: visits(2 boxes each) mean {O[:,0].mean():.2f} wave-max {O[:,1].mean():.2f}; prims mean {O[:,2].mean():.2f} wave-max {O[:,3].mean():.2f}")
        print(f"  threaded: steps mean {S[:,2].mean():.2f} wave-max {S[:,3].mean():.2f}  -> cost 31*max = {31*S[:,3].mean():.0f}  vs ordered 56*max = {56*O[:,1].mean():.0f}")


if __name__ == "__main__":
    main()
