#!/usr/bin/env python
"""Randomised PICpart / comm-array / balancer check on ONE GPU (run on a GPU box):
   python tools/fuzz_picpart.py [seconds] [seed]
Random meshes (2-D annulus, Kuhn box, torus), 2..7 virtual ranks, random partitions (centroid slabs,
random Voronoi blobs, round-robin stripes -- disconnected parts included), random buffer / safe rules,
bridge dimension and layer counts.  The HIP parts (pp_picpart_create) must equal the oracle's
(oracle/ppo_picpart.py) array by array; SUM / MAX / MIN / BCAST reductions of random int32 and double
arrays over vertices and elements must equal the oracle's bit for bit; the balancer's sbars, plan and
selected amounts must equal the oracle's, the plan must be feasible and must not raise the imbalance."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pumipic_amd_loader  # noqa: E402


def random_owner(rng, coords, e2v, P):
    c = coords[e2v].mean(axis=1)
    kind = int(rng.integers(0, 3))
    ne = len(e2v)
    if kind == 0:  # slabs along a random axis
        ax = int(rng.integers(0, coords.shape[1]))
        order = np.argsort(c[:, ax], kind="stable")
        own = np.empty(ne, dtype=np.int32)
        own[order] = (np.arange(ne) * P // ne).astype(np.int32)
    elif kind == 1:  # Voronoi blobs around random elements
        seeds = c[rng.choice(ne, size=P, replace=False)]
        own = np.argmin(((c[:, None, :] - seeds[None, :, :]) ** 2).sum(axis=2), axis=1).astype(np.int32)
    else:  # stripes: every part is made of several disconnected pieces
        ax = int(rng.integers(0, coords.shape[1]))
        order = np.argsort(c[:, ax], kind="stable")
        own = np.empty(ne, dtype=np.int32)
        own[order] = ((np.arange(ne) * (3 * P) // ne) % P).astype(np.int32)
    for r in range(P):  # every rank owns something
        if not np.any(own == r):
            own[int(rng.integers(0, ne))] = r
    return own


def main(seconds=60.0, seed=0):
    pp = pumipic_amd_loader.load()
    ppo = pumipic_amd_loader.load_oracle()
    opp = pumipic_amd_loader.load_oracle_picpart()
    from pumipic_amd import capi
    capi.init(0)
    synth = pp.synth
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    rounds = reductions = plans = 0
    methods = ["FULL", "BFS", "MINIMUM", "NONE"]
    while time.time() < t_end:
        rounds += 1
        which = int(rng.integers(0, 3))
        if which == 0:
            dim = 2
            c, e, k = synth.annulus_tri(n_b=int(rng.integers(4, 10)), n_theta=int(rng.integers(16, 40)), band_width=3)
        elif which == 1:
            dim = 3
            c, e, k = synth.kuhn_box(int(rng.integers(3, 6)))
        else:
            dim = 3
            c, e, k = synth.torus_tet(n_b=int(rng.integers(3, 6)), n_theta=int(rng.integers(8, 16)),
                                      n_planes=int(rng.integers(4, 8)))
        P = int(rng.integers(2, 8))
        owner = random_owner(rng, c, e, P)
        bm, sm = methods[int(rng.integers(0, 4))], methods[int(rng.integers(0, 4))]
        bridge = 0 if rng.integers(0, 2) else dim - 1
        bl, sl = int(rng.integers(0, 4)), int(rng.integers(0, 3))
        cfg = (which, len(e), P, bm, sm, bridge, bl, sl)
        mo = ppo.Mesh(dim, c, e, k)
        O = opp.PicParts(mo, owner, P, getattr(opp, bm), getattr(opp, sm), bridge_dim=bridge, buffer_layers=bl,
                         safe_layers=sl)
        mg = capi.Mesh(dim, c, e, k)
        comms = capi.Comm.local(P)
        parts = [capi.PicPart(mg, owner, comms[r], getattr(capi, "PART_" + bm), getattr(capi, "PART_" + sm), bridge,
                              bl, sl) for r in range(P)]
        for po, pg in zip(O.parts, parts):
            for d in range(dim + 1):
                for which_arr, want in ((capi.PART_GIDS, po.gids[d]), (capi.PART_OWNERS, po.owners[d]),
                                        (capi.PART_RANK_LIDS, po.rank_lids[d]), (capi.PART_COMM_INDEX, po.comm_index[d]),
                                        (capi.PART_FULL_IDS, po.full_ids[d]), (capi.PART_ENT_IDS, po.ent_ids[d])):
                    assert np.array_equal(pg.array(which_arr, d), want), (cfg, d, which_arr)
                assert np.array_equal(pg.nents_offsets(d), po.nents_offsets[d]), cfg
                assert np.array_equal(pg.complete_parts(d), po.is_complete[d]), cfg
            assert np.array_equal(pg.array(capi.PART_SAFE).astype(np.int32), po.safe), cfg
        for _ in range(3):
            d = int(rng.integers(0, dim + 1))
            op = int(rng.integers(0, 4))
            nv = int(rng.integers(1, 4))
            if rng.integers(0, 2):
                arrs = [rng.standard_normal(p.nents[d] * nv) for p in parts]
            else:
                arrs = [rng.integers(-10**6, 10**6, size=p.nents[d] * nv).astype(np.int32) for p in parts]
            devs = [capi.DevArray.from_host(a) for a in arrs]
            capi.picpart_reduce_all(parts, d, op, devs)
            want = O.reduce(d, op, arrs)
            for g, w in zip(devs, want):
                assert np.array_equal(g.to_host(), w), (cfg, d, op, nv)
            reductions += 1
        # balancer (array form): random particles per element
        ob = opp.Balancer(O)
        bals = [capi.Balancer(p) for p in parts]
        ppe = [rng.integers(0, int(rng.integers(1, 200)), size=p.nents[dim]).astype(np.int32) for p in parts]
        tol = float(rng.choice([1.02, 1.05, 1.2]))
        step = float(rng.choice([0.1, 0.3, 0.5]))
        plan_o, W_o, w_o = ob.partition_counts(ppe, tol, step)
        for b, x in zip(bals, ppe):
            assert b.sbars().tolist() == ob.masks, cfg
            b.partition_begin(x)
        procs = [b.partition_end(tol, step) for b in bals]
        before = [int(w_o[r].sum()) for r in range(P)]
        for r, (b, pr) in enumerate(zip(bals, procs)):
            plan, W = b.last_plan()
            assert plan == plan_o[r] and W.tolist() == W_o, cfg
            want = np.zeros(P, dtype=np.int64)
            for i, q, t in plan:
                want[q] += t
                assert (ob.masks[i] >> q) & 1 and (ob.masks[i] >> r) & 1
            assert np.array_equal(np.bincount(pr[pr != r], minlength=P), want), cfg
        assert max(W_o) <= max(before), (cfg, before, W_o)
        plans += 1
        for cm in comms:
            cm.destroy()
    print("fuzz_picpart: %d configurations, %d reductions, %d balancer plans -- OK" % (rounds, reductions, plans))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
