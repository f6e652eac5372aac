"""world_size-2 gloo test (CPU): the multi-rank migration protocol -- element-block ownership,
setUnsafeProcs routing, one all-to-all-v of packed particle records, rebuild with the received
particles -- reproduces the single-rank run particle for particle.  Compute is the CPU oracle; the
collective layer under test is pumi-pic_amd/dist.py (the same code the GPU path uses)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, K, D = 1.72479370 - .08, .020558260, 0.6
REC = np.dtype([("gid", "<i8"), ("x", "<f8", 3), ("xt", "<f8", 3), ("id", "<i4"), ("b", "<f4"),
                ("phi", "<f4"), ("pad", "V12")])  # layout of pp_ps_migrate_pack_records (80 B)


def _population(synth):
    import common
    return common.population_2d(synth, n_b=12, n_theta=48, num_ptcls=2000, mdl_face=3, band_width=3)


def _run_steps(ppo, mesh, ps, nsteps, on_search):
    for step in range(nsteps):
        ppo.elliptical_push(ps, mesh, H, K, D, 6.0, trig=1)
        _, ids, _ = ppo.search_mesh_2d(mesh, ps, looplimit=200)
        ppo.update_positions(ps)
        on_search(step, ids)


def _snapshot(ps):
    se, mk = ps.slot_info()
    cap = ps.capacity()
    live = mk.astype(bool)
    ids = ps.member(2)[0, :cap][live]
    order = np.argsort(ids)
    return ids[order], se[live][order], ps.member(0)[:, :cap][:, live][:, order], \
        ps.member(4)[0, :cap][live][order]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import pumipic_amd_loader
    pp = pumipic_amd_loader.load()
    ppo = pumipic_amd_loader.load_oracle()
    from pumipic_amd import dist as ppdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pop = _population(pp.synth)
    ne = len(pop["e2v"])
    owners = ppdist.element_block_owners(ne, world)
    safe = (owners == rank).astype(np.uint8)     # safe zone = own core (SURVEY config 5)
    mine = owners[pop["elem"]] == rank
    elem = pop["elem"][mine]
    info = [np.ascontiguousarray(a[..., mine]) for a in pop["info"]]
    ppe = np.bincount(elem, minlength=ne).astype(np.int32)
    mesh = ppo.Mesh(2, pop["coords"], pop["e2v"], pop["cls"])
    ps = ppo.PS.scs(ppo.PARTICLE_XGCM, ne, ppe, C_max=4, particle_elements=elem, particle_info=info)
    ps.set_try_shuffling(False)
    moved = [0]

    def on_search(step, ids):
        new_elems, new_procs = ppo.set_unsafe_procs(ps, ids, safe, owners, rank)
        se, mk = ps.slot_info()
        cap = ps.capacity()
        live = mk.astype(bool)
        send_sel = live & (new_procs != rank) & (new_elems >= 0)
        order = np.argsort(new_procs[send_sel], kind="stable")
        idx = np.flatnonzero(send_sel)[order]
        counts = np.bincount(new_procs[idx], minlength=world)
        rec = np.zeros(len(idx), dtype=REC)
        rec["gid"] = new_elems[idx]                # full-mesh replica: gid == lid
        rec["x"] = ps.member(0)[:, :cap][:, idx].T
        rec["xt"] = ps.member(1)[:, :cap][:, idx].T
        rec["id"] = ps.member(2)[0, :cap][idx]
        rec["b"] = ps.member(3)[0, :cap][idx]
        rec["phi"] = ps.member(4)[0, :cap][idx]
        new_elems = new_elems.copy()
        new_elems[idx] = -1                        # removeSentParticles
        send = torch.from_numpy(rec.view(np.uint8).reshape(len(idx), REC.itemsize).copy())
        recv, recv_counts = ppdist.exchange_records(send, counts)
        got = recv.numpy().reshape(-1).view(REC)
        moved[0] += len(idx)
        add_info = [np.ascontiguousarray(got["x"].T), np.ascontiguousarray(got["xt"].T),
                    got["id"].copy(), got["b"].copy(), got["phi"].copy()]
        assert np.all(owners[got["gid"]] == rank)
        ps.rebuild(new_elems, got["gid"].astype(np.int32) if len(got) else None,
                   add_info if len(got) else None)

    _run_steps(ppo, mesh, ps, 6, on_search)
    t = torch.tensor([float(ps.nPtcls())])
    ppdist.allreduce_sum(t)
    q.put((rank, _snapshot(ps), moved[0], int(t.item())))
    dist.destroy_process_group()


def test_two_rank_migration_matches_single_rank(ppo, synth):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    results = [q.get(timeout=240) for _ in procs]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    # single-rank reference run
    pop = _population(synth)
    ne = len(pop["e2v"])
    mesh = ppo.Mesh(2, pop["coords"], pop["e2v"], pop["cls"])
    ps = ppo.PS.scs(ppo.PARTICLE_XGCM, ne, pop["ppe"], C_max=4, particle_elements=pop["elem"],
                    particle_info=pop["info"])
    ps.set_try_shuffling(False)
    _run_steps(ppo, mesh, ps, 6, lambda step, ids: ps.rebuild(ids))
    rid, relem, rx, rphi = _snapshot(ps)
    ids = np.concatenate([r[1][0] for r in results])
    elem = np.concatenate([r[1][1] for r in results])
    x = np.concatenate([r[1][2] for r in results], axis=1)
    phi = np.concatenate([r[1][3] for r in results])
    order = np.argsort(ids)
    assert np.array_equal(ids[order], rid)            # nobody lost or duplicated
    assert np.array_equal(elem[order], relem)         # element ids bit-exact
    assert np.array_equal(x[:, order], rx) and np.array_equal(phi[order], rphi)
    assert sum(r[2] for r in results) > 0             # particles really crossed the partition
    assert all(r[3] == len(rid) for r in results)     # all-reduce of the per-rank counts
    from pumipic_amd import dist as ppdist
    owners = ppdist.element_block_owners(ne, 2)
    for r in results:                                 # every rank holds only elements it owns
        assert np.all(owners[r[1][1]] == r[0])
