#!/usr/bin/env python
"""Randomised comparison of the record-to-record rebuild (pp_ps::lazy_rec == 3: particle types wider than 64 B, round 5)
against the oracle (run on a GPU box):   python tools/fuzz_records.py [seconds] [seed]

Random Sell-C-sigma / CSR structures of the 160-byte ps_combo160 particle (and a 96-byte type), chains of rebuilds with
NO member access in between -- a particle's destination is a function of its current ELEMENT and the round, which both
sides evaluate from their own layout -- interleaved at random with everything that has to find the member arrays valid:
a member read, a member written from the host, new particles, a pseudo-push (which gives the records up), getPIDs, the
in-place rebuild when the layout can be kept.  At random checkpoints and at the end every member of every particle must
equal the oracle's by particle id."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pumipic_amd_loader  # noqa: E402
import common  # noqa: E402


def compare(po, pg, nmem, idm):
    capo, capg = po.capacity(), pg.capacity()
    so, mko = po.slot_info()
    sg, mkg = pg.slot_info()
    ido, idg = po.member(idm)[0, :capo], pg.member(idm)[0, :capg]
    io, eo = common.by_id(ido, mko, so[:capo])
    ig, eg = common.by_id(idg, mkg, sg[:capg])
    if not (np.array_equal(io, ig) and np.array_equal(eo, eg)):
        return False
    for m in range(nmem):
        _, a = common.by_id(ido, mko, po.member(m)[:, :capo])
        _, b = common.by_id(idg, mkg, pg.member(m)[:, :capg])
        if not np.array_equal(a, b):
            return False
    return True


def main(seconds=60.0, seed=0):
    pp = pumipic_amd_loader.load()
    ppo = pumipic_amd_loader.load_oracle()
    from pumipic_amd import capi
    capi.init(0)
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    structures = rebuilds = fed = checks = 0
    wide96 = [(np.float64, 9), (np.int32, 4), (np.int64, 1)]  # 72 + 16 + 8 = 96 B: 6-quad records
    while time.time() < t_end:
        structures += 1
        ne = int(rng.integers(1, 3000))
        npt = int(rng.integers(1, 30000))
        elems = np.sort(rng.integers(0, ne, size=npt).astype(np.int32))
        ppe = np.bincount(elems, minlength=ne).astype(np.int32)
        ids = np.arange(npt, dtype=np.int64)
        perf = rng.random() < 0.7
        mo, mg = (ppo.PERF160, capi.PERF160) if perf else (wide96, wide96)
        nd = 17 if perf else 9
        info = [np.stack([ids + 0.001 * c for c in range(nd)]), np.stack([4 * ids + c for c in range(4)]).astype(np.int32),
                ids[None, :].copy()]
        kind = "csr" if rng.random() < 0.35 else "scs"
        if kind == "scs":
            C = int(rng.choice([1, 32, 64, 64]))
            sigma = int(rng.choice([1, ne, 2**31 - 1]))
            po = ppo.PS.scs(mo, ne, ppe, C_max=C, sigma=sigma, V=1024, particle_elements=elems, particle_info=info)
            pg = capi.PS.scs(mg, ne, ppe, C_=C, sigma=sigma, V=1024, particle_elements=elems, particle_info=info)
            shuffle = bool(rng.random() < 0.4)
            po.set_try_shuffling(shuffle)
            pg.set_try_shuffling(shuffle)
        else:
            po = ppo.PS.csr(mo, ne, ppe, particle_elements=elems, particle_info=info)
            pg = capi.PS.csr(mg, ne, ppe, particle_elements=elems, particle_info=info)
        next_id = npt
        before = pg.rebuild_stats()
        for rnd in range(int(rng.integers(3, 14))):
            # destination table of the round: a function of the current element
            table = rng.integers(0, ne, size=ne).astype(np.int64)
            stay = rng.random(ne) < rng.choice([0.3, 0.5, 0.95])
            table = np.where(stay, np.arange(ne), table)
            table[rng.random(ne) < 0.02] = -1  # a few elements lose their particles
            news = []
            for ps_ in (po, pg):
                se, mk = ps_.slot_info()
                mk = mk.astype(bool)
                dest = np.where(mk & (se >= 0) & (se < ne), table[np.clip(se, 0, ne - 1)], -1)
                news.append(dest.astype(np.int32))
            n_new = int(rng.integers(1, 2000)) if rng.random() < 0.25 else 0
            add_e, add = None, None
            if n_new:
                nid = np.arange(next_id, next_id + n_new, dtype=np.int64)
                add_e = rng.integers(0, ne, size=n_new).astype(np.int32)
                add = [np.stack([nid + 0.001 * c for c in range(nd)]), np.stack([4 * nid + c for c in range(4)]).astype(np.int32),
                       nid[None, :].copy()]
                next_id += n_new
            po.rebuild(news[0], add_e, add)
            pg.rebuild(news[1], add_e, add)
            rebuilds += 1
            if po.nPtcls() != pg.nPtcls():
                print("FAIL: counts differ after a rebuild (seed %d, structure %d, round %d)" % (seed, structures, rnd))
                return 1
            if po.nPtcls() == 0:
                break
            what = rng.random()
            if what < 0.15:  # checkpoint: every member by id
                checks += 1
                if not compare(po, pg, 3, 2):
                    print("FAIL: members differ at a checkpoint (seed %d, structure %d, round %d)" % (seed, structures, rnd))
                    return 1
            elif what < 0.22:  # one member written from the host (the others must survive it)
                cap = pg.capacity()
                for ps_ in (po, pg):
                    se, mk = ps_.slot_info()
                    live = mk.astype(bool)
                    arr = ps_.member(1).copy()
                    arr[3, :ps_.capacity()][live] = 7 * ps_.member(2)[0, :ps_.capacity()][live].astype(np.int32) + rnd
                    if ps_ is pg:
                        ps_.set_member(1, arr)
                    else:
                        ps_.member(1)[...] = arr
            elif what < 0.27 and perf:  # a pseudo-push: both sides overwrite everything but keep nothing comparable by id
                parent = np.sqrt(np.arange(ne, dtype=np.float64)) * np.arange(ne)
                ppo.pseudo_push160(po, parent)
                capi.pseudo_push160(pg, capi.DevArray.from_host(parent))
                # (ids are slots now on both sides, and slot orders differ: re-tag by element-major order)
                for ps_ in (po, pg):
                    se, mk = ps_.slot_info()
                    live = np.flatnonzero(mk)
                    order = live[np.argsort(se[live], kind="stable")]
                    tag = ps_.member(2).copy()
                    tag[0, order] = np.arange(len(order), dtype=np.int64)
                    dbl = ps_.member(0).copy()
                    dbl[:, order] = np.arange(len(order))[None, :] + 0.5
                    num = ps_.member(1).copy()
                    num[:, order] = np.arange(len(order), dtype=np.int32)[None, :]
                    if ps_ is pg:
                        ps_.set_member(2, tag)
                        ps_.set_member(0, dbl)
                        ps_.set_member(1, num)
                    else:
                        ps_.member(2)[...] = tag
                        ps_.member(0)[...] = dbl
                        ps_.member(1)[...] = num
                next_id = max(next_id, po.nPtcls())
            elif what < 0.32:
                oo, pi = po.get_pids()
                og, pgi = pg.get_pids()
                if not np.array_equal(oo, og):
                    print("FAIL: getPIDs offsets differ (seed %d, structure %d)" % (seed, structures))
                    return 1
        after = pg.rebuild_stats()
        fed += after[2] - before[2]
        if po.nPtcls() > 0:
            checks += 1
            if not compare(po, pg, 3, 2):
                print("FAIL: members differ at the end (seed %d, structure %d)" % (seed, structures))
                return 1
    print("fuzz ok: %d structures, %d rebuilds (%d fed by the previous one's records), %d member comparisons"
          % (structures, rebuilds, fed, checks))
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
