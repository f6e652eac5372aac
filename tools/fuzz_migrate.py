#!/usr/bin/env python
"""Randomised multi-rank migration check on ONE GPU (run on a GPU box):
   python tools/fuzz_migrate.py [seconds] [seed]
2..5 element-block 'ranks' live in one process; every step each rank pushes+searches, routes its
particles (pp_set_unsafe_procs with a random BFS safe zone), packs records, the records are
exchanged through host memory (standing in for the all-to-all-v), and every rank rebuilds -- plain or
with the position commit in the records and the scatters behind the rebuild.  The union of the ranks
must equal the single-structure oracle run by particle id, and the ranks' scatter fields must add up
to the oracle's field."""
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


def main(seconds=60.0, seed=0):
    pp = pumipic_amd_loader.load()
    ppo = pumipic_amd_loader.load_oracle()
    from pumipic_amd import capi
    capi.init(0)
    synth = pp.synth
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    rounds = steps_done = moved = 0
    while time.time() < t_end:
        rounds += 1
        world = int(rng.integers(2, 6))
        fused = bool(rng.integers(0, 2))
        safe_layers = int(rng.integers(0, 3))
        pop = common.population_2d(synth, n_b=int(rng.integers(6, 14)), n_theta=int(rng.integers(24, 64)),
                                   num_ptcls=int(rng.integers(200, 4000)), mdl_face=int(rng.integers(2, 4)),
                                   band_width=int(rng.integers(2, 4)))
        ne = len(pop["e2v"])
        owners = (np.arange(ne, dtype=np.int64) * world // ne).astype(np.int32)
        mesh = capi.Mesh(2, pop["coords"], pop["e2v"], pop["cls"])
        mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
        fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
        fg, bg = capi.create_gyro_ring_mappings(mesh)
        owners_d = capi.DevArray.from_host(owners)
        ranks, safes = [], []
        for r in range(world):
            mine = owners[pop["elem"]] == r
            elem = pop["elem"][mine]
            info = [np.ascontiguousarray(a[..., mine]) for a in pop["info"]]
            ranks.append(capi.PS.scs(capi.PARTICLE_XGCM, ne, np.bincount(elem, minlength=ne).astype(np.int32),
                                     gids=np.arange(ne, dtype=np.int64), particle_elements=elem, particle_info=info))
            safes.append(capi.bfs_buffer_layers(mesh, owners_d, r, world, safe_layers, safe_layers)[0])
        recb = capi.migrate_record_bytes(ranks[0])
        deg = float(rng.choice([2.0, 6.0, 15.0]))
        for step in range(int(rng.integers(2, 6))):
            steps_done += 1
            ppo.elliptical_push(po, mo, H, K, D, deg, trig=1)
            _, ids_o, _ = ppo.search_mesh_2d(mo, po, looplimit=200)
            ppo.update_positions(po)
            po.rebuild(ids_o)
            outbox, fields = [], []
            for r, ps in enumerate(ranks):
                ids = capi.DevArray.from_host(np.full(max(ps.capacity(), 1), -1, dtype=np.int32))
                capi.push_search(mesh, ps, H, K, D, deg, ids, seeded=True, looplimit=200)
                if not fused:
                    capi.update_positions(ps)
                ne_d, np_d = capi.set_unsafe_procs(ps, ids, safes[r], owners_d, r)
                counts = capi.migrate_count(ps, ne_d, np_d, r, world)
                buf = capi.DevArray(max(int(counts.sum()) * recb, 1), np.uint8)
                if fused:
                    capi.migrate_pack_records_commit(ps, ne_d, np_d, r, world, counts, buf.ptr)
                else:
                    capi.migrate_pack_records(ps, ne_d, np_d, r, world, counts, buf.ptr)
                outbox.append((ne_d, counts, buf.to_host()[:int(counts.sum()) * recb].reshape(-1, recb)))
                moved += int(counts.sum())
            for r, ps in enumerate(ranks):
                parts = []
                for src in range(world):
                    _, counts, data = outbox[src]
                    start = int(counts[:r].sum())
                    parts.append(data[start:start + int(counts[r])])
                recv = np.concatenate(parts) if parts else np.zeros((0, recb), np.uint8)
                rbuf = capi.DevArray.from_host(np.ascontiguousarray(recv).reshape(-1)) if len(recv) else capi.DevArray(1, np.uint8)
                if fused:
                    wf, wb = capi.DevArray(mesh.nverts, np.float64), capi.DevArray(mesh.nverts, np.float64)
                    capi.rebuild_records_scatter(ps, outbox[r][0], len(recv), rbuf.ptr, mesh, [fg, bg], [wf, wb])
                    fields.append((wf.to_host(), wb.to_host()))
                else:
                    capi.rebuild_records(ps, outbox[r][0], len(recv), rbuf.ptr)
                    fields.append((capi.gyro_scatter(mesh, ps, fg).to_host(), capi.gyro_scatter(mesh, ps, bg).to_host()))
            c_f = np.array_equal(sum(f for f, _ in fields), ppo.gyro_scatter(mo, po, fo))
            c_b = np.array_equal(sum(b for _, b in fields), ppo.gyro_scatter(mo, po, bo))
            ok = c_f and c_b
            ids_all, elem_all, x_all = [], [], []
            for r, ps in enumerate(ranks):
                if ps.nPtcls() == 0:  # (the oracle keeps a stale mask behind an emptying rebuild, SCS_rebuild.h:168-176)
                    continue
                se, mk = ps.slot_info()
                cap = ps.capacity()
                live = mk.astype(bool)
                ids_all.append(ps.member(2)[0, :cap][live])
                elem_all.append(se[live])
                x_all.append(ps.member(0)[:, :cap][:, live])
            ids_all = np.concatenate(ids_all) if ids_all else np.zeros(0, np.int32)
            elem_all = elem_all or [np.zeros(0, np.int32)]
            x_all = x_all or [np.zeros((3, 0))]
            order = np.argsort(ids_all)
            so, mko = po.slot_info()
            io, eo = common.by_id(po.member(2)[0, :po.capacity()], mko, so)
            _, xo = common.by_id(po.member(2)[0, :po.capacity()], mko, po.member(0)[:, :po.capacity()])
            c_i = np.array_equal(ids_all[order], io)
            c_e = c_i and np.array_equal(np.concatenate(elem_all)[order], eo)
            c_x = c_i and np.array_equal(np.concatenate(x_all, axis=1)[:, order], xo)
            ok &= c_i and c_e and c_x
            if not ok:
                print("fields", c_f, c_b, "ids", c_i, "elems", c_e, "x", c_x, "counts", len(ids_all), len(io),
                      "dups", len(ids_all) - len(np.unique(ids_all)))
                if not c_i:
                    print("  only gpu", np.setdiff1d(ids_all, io)[:10], "only oracle", np.setdiff1d(io, ids_all)[:10])
                print("MISMATCH round %d step %d: world %d fused %s safe_layers %d deg %g ne %d" % (
                    rounds, step, world, fused, safe_layers, deg, ne))
                return 1
    print("fuzz ok: %d configurations, %d steps, %d particles migrated" % (rounds, steps_done, moved))
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
