#!/usr/bin/env python
"""Randomised rebuild / search comparison of the HIP path against the oracle (run on a GPU box):
   python tools/fuzz_rebuild.py [seconds] [seed]
Random structures (SCS with random C, V, sigma, padding; CSR), random per-step mixes of stayers,
movers, deletions and new particles, occasional bursts and collapses.  After every rebuild the
populations must agree by particle id (element, members)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pumipic_amd_loader  # noqa: E402
import common  # noqa: E402


def population(ps, members):
    se, mk = ps.slot_info()
    cap = ps.capacity()
    live = mk.astype(bool)
    ids = ps.member(2)[0, :cap][live]
    order = np.argsort(ids, kind="stable")
    out = [ids[order], se[:cap][live][order]]
    for m in range(len(members)):
        out.append(ps.member(m)[:, :cap][:, live][:, order])
    return out


def main(seconds=60.0, seed=0):
    pp = pumipic_amd_loader.load()
    ppo = pumipic_amd_loader.load_oracle()
    from pumipic_amd import capi
    capi.init(0)
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    rounds = steps = 0
    inplace_total = 0
    while time.time() < t_end:
        rounds += 1
        ne = int(rng.integers(1, 4000))
        npt = int(rng.integers(0, 20000))
        elems = np.sort(rng.integers(0, ne, size=npt).astype(np.int32))
        if rng.random() < 0.3 and npt:  # skewed: most particles in a few elements
            elems = np.sort((rng.integers(0, ne, size=npt) ** 3 // max(ne * ne, 1)).astype(np.int32) % ne)
        ppe = np.bincount(elems, minlength=ne).astype(np.int32)
        members = ppo.PARTICLE_XGCM
        info = [rng.random((3, npt)), rng.random((3, npt)), np.arange(npt, dtype=np.int32),
                rng.random(npt).astype(np.float32), rng.random(npt).astype(np.float32)]
        kind = "csr" if rng.random() < 0.3 else "scs"
        if kind == "scs":
            C = int(rng.choice([1, 4, 32, 64]))
            V = int(rng.choice([2, 64, 1024]))
            sigma = int(rng.choice([1, 8, 2**31 - 1]))
            pad = int(rng.integers(0, 3))
            po = ppo.PS.scs(members, ne, ppe, C_max=C, sigma=sigma, V=V, pad_strat=pad, particle_elements=elems,
                            particle_info=info)
            pg = capi.PS.scs(capi.PARTICLE_XGCM, ne, ppe, C_=C, sigma=sigma, V=V, pad_strat=pad,
                             particle_elements=elems, particle_info=info)
            shuffle = bool(rng.random() < 0.6)  # SellCSigma::setShuffling on both sides
            po.set_try_shuffling(shuffle)
            pg.set_try_shuffling(shuffle)
            desc = "scs C=%d V=%d sigma=%d pad=%d shuffle=%d" % (C, V, sigma, pad, shuffle)
        else:
            po = ppo.PS.csr(members, ne, ppe, particle_elements=elems, particle_info=info)
            pg = capi.PS.csr(capi.PARTICLE_XGCM, ne, ppe, particle_elements=elems, particle_info=info)
            desc = "csr"
        next_id = npt
        mask_holes = False
        for it in range(int(rng.integers(2, 9))):
            steps += 1
            mode = rng.random()
            move_frac = rng.random() if mode < 0.8 else 1.0
            if kind == "scs" and shuffle and rng.random() < 0.6:
                move_frac *= 0.1  # few movers: the layout can usually be kept
            del_frac = rng.random() * 0.2 if mode < 0.9 else 0.97
            if kind == "scs" and shuffle and rng.random() < 0.5:
                del_frac *= 0.05
            n_new = int(rng.integers(0, 3000)) if rng.random() < 0.5 else 0
            if rng.random() < 0.1:
                n_new = int(rng.integers(10000, 60000))
            target = int(rng.integers(0, ne)) if rng.random() < 0.1 else -1
            dec = rng.integers(0, ne, size=max(next_id, 1)).astype(np.int32)
            if target >= 0:
                dec[:] = target
            mv = rng.random(max(next_id, 1)) < move_frac
            dl = rng.random(max(next_id, 1)) < del_frac
            add_e = rng.integers(0, ne, size=n_new).astype(np.int32)
            add = None
            if n_new:
                add = [rng.random((3, n_new)), rng.random((3, n_new)),
                       np.arange(next_id, next_id + n_new, dtype=np.int32),
                       rng.random(n_new).astype(np.float32), rng.random(n_new).astype(np.float32)]
            commit = kind == "scs" and rng.random() < 0.5
            news = []
            for ps_ in (po, pg):
                se, mk = ps_.slot_info()
                ids = ps_.member(2)[0, :ps_.capacity()]
                new = np.full(max(len(se), 1), -1, dtype=np.int32)
                live = mk.astype(bool)
                i = ids[live]
                e = np.where(mv[i], dec[i], se[live])
                e = np.where(dl[i], -1, e)
                new[:len(se)][live] = e
                news.append(new)
            if commit:
                ppo.update_positions(po)
            po.rebuild(news[0][:max(po.capacity(), 0)] if po.capacity() else news[0][:0], add_e if n_new else None, add)
            before = pg.rebuild_stats() if kind == "scs" else (0, 0)
            if commit and n_new == 0:
                pg.rebuild_commit(news[1])
            else:
                if commit:
                    capi.update_positions(pg)
                pg.rebuild(news[1], add_e if n_new else None, add)
            next_id += n_new
            a, b = population(po, members), population(pg, members)
            ok = po.nPtcls() == pg.nPtcls() and all(np.array_equal(x, y) for x, y in zip(a, b))
            if po.nPtcls() == 0 and pg.nPtcls() == 0:
                ok = True  # the reference leaves a stale mask behind an emptying rebuild; the library clears it
            in_place = False
            if kind == "scs":  # same reshuffle-or-rebuild decision on both sides
                in_place = pg.rebuild_stats()[0] > before[0]
                inplace_total += in_place
                if bool(po.s.last_rebuild_was_shuffle) != in_place:
                    print("decision differs: oracle shuffle %d gpu in place %d" % (po.s.last_rebuild_was_shuffle, in_place))
                    ok = False
                # the reference's reshuffle fills arbitrary holes (rows are no longer prefix-compact there);
                # a full rebuild makes both sides compact again
                mask_holes = in_place or (mask_holes and bool(po.s.last_rebuild_was_shuffle))
            if ok and po.nPtcls() > 0:  # the layout arrays themselves (row order = stable sort by count)
                lo, lg = po.layout(), pg.layout()
                for k in (("C", "num_chunks", "num_slices", "capacity", "num_rows") if kind == "scs" else ("capacity",)):
                    if lo[k] != lg[k]:
                        print("layout field %s: oracle %s gpu %s" % (k, lo[k], lg[k]))
                    ok &= lo[k] == lg[k]
                keys = ("offsets", "slice_to_chunk", "row_to_element", "element_to_row") \
                    if kind == "scs" else ("offsets",)
                if kind == "scs" and not mask_holes:
                    keys += ("mask",)  # comparable until the reference's reshuffle has left holes in rows
                for k in keys:
                    if ok and k in lo and k in lg:
                        n = min(len(lo[k]), len(lg[k]))
                        same = np.array_equal(np.asarray(lo[k])[:n], np.asarray(lg[k])[:n])
                        if not same:
                            print("layout array %s differs" % k)
                        ok &= same
            if not ok:
                print("MISMATCH round %d step %d: %s ne=%d np=%d move=%.2f del=%.2f new=%d target=%d commit=%d "
                      "oracle %d gpu %d" % (rounds, it, desc, ne, npt, move_frac, del_frac, n_new, target, commit,
                                            po.nPtcls(), pg.nPtcls()))
                return 1
    print("fuzz ok: %d structures, %d rebuilds (%d kept the layout)" % (rounds, steps, inplace_total))
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
