#!/usr/bin/env python
"""Randomised push + search comparison of the HIP path against the oracle (run on a GPU box):
   python tools/fuzz_search.py [seconds] [seed]
Random annulus / torus meshes and populations, random SCS chunk heights or CSR, random push angles
and loop limits; several steps of the c2 flow (fused push+search re-seeded from the previous ids,
x <-> x_tgt swap) and of the c3 flow (commit + rebuild), plus the stand-alone searches (BCC,
intersection, 2-D, search_mesh_3d, legacy).  Ids, x_tgt and phi must be bit-identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pumipic_amd_loader  # noqa: E402
import common  # noqa: E402

H, K, D = 1.72479370 - .08, .020558260, 0.6


def by_id(ps, values):
    cap = ps.capacity()
    return common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], np.asarray(values)[..., :cap])


def main(seconds=60.0, seed=0):
    pp = pumipic_amd_loader.load()
    ppo = pumipic_amd_loader.load_oracle()
    from pumipic_amd import capi
    capi.init(0)
    synth = pp.synth
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    rounds = checks = 0
    while time.time() < t_end:
        rounds += 1
        dim = int(rng.choice([2, 3]))
        npt = int(rng.integers(1, 6000))
        if dim == 2:
            pop = common.population_2d(synth, n_b=int(rng.integers(4, 16)), n_theta=int(rng.integers(12, 64)),
                                       num_ptcls=npt, mdl_face=int(rng.integers(1, 4)), band_width=int(rng.integers(1, 4)))
        else:
            pop = common.population_3d(synth, n_b=int(rng.integers(3, 8)), n_theta=int(rng.integers(8, 24)),
                                       n_planes=int(rng.integers(4, 10)), num_ptcls=npt, mdl_face=int(rng.integers(1, 5)))
        kind = "csr" if rng.random() < 0.25 else "scs"
        C = int(rng.choice([1, 8, 48, 64]))
        mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, kind, C=C)
        mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, kind, C=C)
        desc = "dim %d %s C=%d np=%d ne=%d" % (dim, kind, C, npt, mo.nelems)
        deg = float(rng.choice([0.0, 0.5, 3.0, 12.0, 40.0]))
        limit = int(rng.choice([0, 1, 2, 5, 200]))
        flow = rng.choice(["c2", "c3", "search"])
        ok = True
        if flow == "search":
            (ppo.elliptical_push if dim == 2 else ppo.toroidal_push)(po, mo, H, K, D, deg, trig=1)
            (capi.elliptical_push if dim == 2 else capi.toroidal_push)(pg, mg, H, K, D, deg)
            cap = po.capacity()
            for mt in (False, True):
                ro = ppo.search_mesh(mo, po, require_intersection=mt, looplimit=limit)
                rg = capi.search_mesh(mg, pg, require_intersection=mt, looplimit=limit)
                c = [ro["found"] == rg["found"], np.array_equal(ro["elem_ids"], rg["elem_ids"].to_host()[:cap])]
                if mt:
                    c.append(np.array_equal(ro["inter_faces"], rg["inter_faces"].to_host()[:cap]))
                    c.append(np.array_equal(ro["inter_points"].ravel(), rg["inter_points"].to_host()[:cap * dim]))
                if not all(c):
                    d = np.flatnonzero(ro["elem_ids"] != rg["elem_ids"].to_host()[:cap])
                    print("search_mesh mt=%s:" % mt, c, "found", ro["found"], rg["found"], "id diffs", d[:5],
                          ro["elem_ids"][d[:5]], rg["elem_ids"].to_host()[:cap][d[:5]], po.slot_info()[1][d[:5]])
                ok &= all(c)
                checks += 1
            # seeded: some particles already deleted (-1), some with a wrong parent element
            se, mk = po.slot_info()
            seed_ids = np.where(mk.astype(bool), se, -1).astype(np.int32)
            lv = np.flatnonzero(mk)
            if len(lv):
                seed_ids[lv[rng.random(len(lv)) < 0.1]] = -1
                wrong = lv[rng.random(len(lv)) < 0.1]
                seed_ids[wrong] = (seed_ids[wrong] + 1 + rng.integers(0, mo.nelems, size=len(wrong))) % mo.nelems
                seed_ids[np.flatnonzero(~mk.astype(bool))] = -1
            ro = ppo.search_mesh(mo, po, elem_ids=seed_ids.copy(), looplimit=limit)
            rg = capi.search_mesh(mg, pg, elem_ids=capi.DevArray.from_host(seed_ids), looplimit=limit)
            c = [ro["found"] == rg["found"], ro["not_in_elem"] == rg["not_in_elem"],
                 np.array_equal(ro["elem_ids"], rg["elem_ids"].to_host()[:cap])]
            if not all(c):
                print("seeded search_mesh:", c, ro["not_in_elem"], rg["not_in_elem"])
            ok &= all(c)
            st = capi.trace_particle_through_mesh(mg, pg, None, elem_ids=capi.DevArray.from_host(seed_ids),
                                                  looplimit=limit)
            c = [ro["found"] == st["found"], ro["loops"] == st["loops"], ro["not_in_elem"] == st["not_in_elem"],
                 np.array_equal(ro["elem_ids"], st["elem_ids"].to_host()[:cap])]
            if not all(c):
                print("stepwise trace:", c)
            ok &= all(c)
            checks += 2
            if dim == 3:
                # a walk started from a wrong element need not terminate (checkParent looks at the ROW
                # element, hpp:371-382): always with a loop limit
                lim3 = limit if limit else 60
                r3o = ppo.search_mesh_3d(mo, po, elem_ids=seed_ids.copy(), looplimit=lim3)
                r3g = capi.search_mesh_3d(mg, pg, elem_ids=capi.DevArray.from_host(seed_ids), looplimit=lim3)
                c = [r3o["found"] == r3g["found"], np.array_equal(r3o["elem_ids"], r3g["elem_ids"].to_host()[:cap]),
                     np.array_equal(r3o["xface"], r3g["xface"].to_host()[:cap])]
                if not all(c):
                    print("seeded search_mesh_3d:", c, r3o["found"], r3g["found"])
                ok &= all(c)
                checks += 1
                for fo, fg in ((ppo.search_mesh_3d, capi.search_mesh_3d), (ppo.search_mesh_legacy3d, capi.search_mesh_legacy3d)):
                    ro, rg = fo(mo, po, looplimit=limit), fg(mg, pg, looplimit=limit)
                    ok &= ro["found"] == rg["found"] and np.array_equal(ro["elem_ids"], rg["elem_ids"].to_host()[:cap])
                    ok &= np.array_equal(ro["xface"], rg["xface"].to_host()[:cap])
                    checks += 1
            else:
                fo, ido, _ = ppo.search_mesh_2d(mo, po, looplimit=limit)
                fg, idg = capi.search_mesh_2d(mg, pg, looplimit=limit)
                c = [bool(fo) == bool(fg), np.array_equal(ido, idg.to_host()[:cap])]
                if not all(c):
                    print("search_mesh_2d:", c, fo, fg)
                ok &= all(c)
                checks += 1
        else:
            ids_o = None
            ids_g = capi.DevArray.from_host(np.full(max(pg.capacity(), 1), -1, dtype=np.int32))
            trust = bool(rng.random() < 0.5)  # from step 1 on the origins are the accepted destinations
            for step in range(int(rng.integers(2, 7))):
                pg.set_origin_trust(trust and step > 0)
                cap = po.capacity()
                if dim == 3:
                    ppo.toroidal_push(po, mo, H, K, D, deg, trig=1)
                    r = ppo.search_mesh(mo, po, elem_ids=ids_o, looplimit=limit)
                    ids_o, fo = r["elem_ids"], r["found"]
                    fg = capi.push_search(mg, pg, H, K, D, deg, ids_g, seeded=step > 0 and flow == "c2", looplimit=limit)
                else:
                    ppo.elliptical_push(po, mo, H, K, D, deg, trig=1)
                    fo, ids_o, _ = ppo.search_mesh_2d(mo, po, elem_ids=ids_o, looplimit=limit)
                    fg = capi.push_search(mg, pg, H, K, D, deg, ids_g, seeded=True, looplimit=limit)
                got = ids_g.to_host()
                c1 = bool(fo) == bool(fg)
                # slot order inside a row is free after a rebuild: compare by particle id
                mko, mkg = po.slot_info()[1], pg.slot_info()[1]
                pido, pidg = po.member(2)[0, :cap], pg.member(2)[0, :pg.capacity()]
                io, eo = common.by_id(pido, mko, ids_o[:cap])
                ig, eg = common.by_id(pidg, mkg, got[:pg.capacity()])
                c2 = np.array_equal(io, ig) and np.array_equal(eo, eg)
                c3 = np.array_equal(common.by_id(pido, mko, po.member(1)[:, :cap])[1],
                                    common.by_id(pidg, mkg, pg.member(1)[:, :pg.capacity()])[1])
                c4 = np.array_equal(common.by_id(pido, mko, po.member(4)[:, :cap])[1],
                                    common.by_id(pidg, mkg, pg.member(4)[:, :pg.capacity()])[1])
                if not (c1 and c2 and c3 and c4):
                    print("step %d: found %s/%s ids %s x_tgt %s phi %s" % (step, fo, fg, c2, c3, c4))
                ok &= c1 and c2 and c3 and c4
                checks += 1
                if not ok:
                    break
                if flow == "c2":
                    a, b = po.member(0), po.member(1)
                    tmp = a.copy()
                    a[:] = b
                    b[:] = tmp
                    pg.swap_members(0, 1)
                    if dim == 2:
                        ids_o = ids_o.copy()
                else:
                    ppo.update_positions(po)
                    po.rebuild(ids_o)
                    pg.rebuild_commit(ids_g)
                    ids_o = None
                    ids_g = capi.DevArray.from_host(np.full(max(pg.capacity(), 1), -1, dtype=np.int32))
                    io, xo = by_id(po, po.member(0))
                    ig, xg = by_id(pg, pg.member(0))
                    c5 = np.array_equal(io, ig) and np.array_equal(xo, xg)
                    if po.nPtcls() == 0 and pg.nPtcls() == 0:
                        c5 = True  # the reference leaves a stale mask behind an emptying rebuild (SCS_rebuild.h:168-176
                        #            runs resetMask after num_ptcls = 0, and parallel_for returns at once); the library clears it
                    if not c5:
                        print("after rebuild_commit step %d: ids %s x %s (oracle %d gpu %d particles, rebuild stats %s, "
                              "oracle shuffled %d)" % (step, np.array_equal(io, ig), io.shape == ig.shape and np.array_equal(xo, xg),
                                                        po.nPtcls(), pg.nPtcls(), pg.rebuild_stats(), po.s.last_rebuild_was_shuffle))
                    ok &= c5
                    if po.nPtcls() == 0:
                        break
        if ok and rng.random() < 0.3 and po.nPtcls() > 0:  # gyro ring maps + scatter, random ring geometry
            gnr, gppr = int(rng.integers(2, 5)), int(rng.choice([4, 6, 8]))
            rmax, theta = float(rng.uniform(0.005, 0.08)), float(rng.uniform(0, 40))
            fo, bo = ppo.create_gyro_ring_mappings(mo, rmax, gnr, gppr, theta, trig=1)
            fg, bg = capi.create_gyro_ring_mappings(mg, rmax, gnr, gppr, theta)
            c = [np.array_equal(fo, fg.to_host()[:len(fo)]), np.array_equal(bo, bg.to_host()[:len(bo)])]
            c.append(np.array_equal(ppo.gyro_scatter(mo, po, fo, rmax, gnr, gppr),
                                    capi.gyro_scatter(mg, pg, fg, rmax, gnr, gppr).to_host()[:mo.nverts]))
            if not all(c):
                print("gyro maps / scatter:", c, gnr, gppr, rmax, theta)
            ok &= all(c)
            checks += 1
        if not ok:
            print("MISMATCH round %d: %s flow %s deg %g looplimit %d" % (rounds, desc, flow, deg, limit))
            return 1
    print("fuzz ok: %d configurations, %d comparisons" % (rounds, checks))
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
