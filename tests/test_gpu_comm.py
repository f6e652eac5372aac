"""-m gpu: the multi-rank path behind the C-ABI (pp_comm_*, pp_ps_migrate*, pp_allreduce_sum).

A 1-GPU box cannot form a >1-rank RCCL job (RCCL refuses two ranks on one device), so the
multi-rank behaviour is covered three ways:
  * `local` communicator: 2-3 virtual ranks in one process through pp_ps_migrate_begin / _end,
  * TWO PROCESSES sharing the GPU, exchanging through the built-in TCP transport and through a
    caller-supplied transport (gloo callbacks) -- the same pp_ps_migrate_scatter entry point, the
    same pack / unpack / rebuild kernels, only the byte mover differs from the RCCL path,
  * RCCL itself with one rank (library resolution, communicator creation, all-gather of the
    counts, in-place all-reduce, stream ordering).
Every run is compared particle by particle with the single-structure CPU oracle run."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, K, D = 1.72479370 - .08, .020558260, 0.6
NSTEPS = 5


@pytest.fixture(scope="module")
def capi(pp):
    from pumipic_amd import capi as c
    c.build()
    c.init(0)
    return c


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _population(synth, dim):
    if dim == 2:
        return common.population_2d(synth, n_b=12, n_theta=48, num_ptcls=3000, mdl_face=3, band_width=3)
    return common.population_3d(synth, num_ptcls=3000)


def _rank_structure(capi, pop, owners, r):
    ne = len(pop["e2v"])
    mine = owners[pop["elem"]] == r
    elem = pop["elem"][mine]
    info = [np.ascontiguousarray(a[..., mine]) for a in pop["info"]]
    return capi.PS.scs(capi.PARTICLE_XGCM, ne, np.bincount(elem, minlength=ne).astype(np.int32),
                       gids=np.arange(ne, dtype=np.int64), particle_elements=elem, particle_info=info)


def _oracle_run(ppo, pop, nsteps, deg=6.0):
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    maps = ppo.create_gyro_ring_mappings(mo, trig=1)
    fields = []
    for _ in range(nsteps):
        if pop["dim"] == 2:
            ppo.elliptical_push(po, mo, H, K, D, deg, trig=1)
            _, ids, _ = ppo.search_mesh_2d(mo, po, looplimit=200)
        else:
            ppo.toroidal_push(po, mo, H, K, D, deg, trig=1)
            ids = ppo.search_mesh(mo, po, looplimit=200)["elem_ids"]
        ppo.update_positions(po)
        po.rebuild(ids)
        fields.append((ppo.gyro_scatter(mo, po, maps[0]), ppo.gyro_scatter(mo, po, maps[1])))
    so, mko = po.slot_info()
    cap = po.capacity()
    io, eo = common.by_id(po.member(2)[0, :cap], mko, so)
    _, xo = common.by_id(po.member(2)[0, :cap], mko, po.member(0)[:, :cap])
    _, pho = common.by_id(po.member(2)[0, :cap], mko, po.member(4)[0, :cap])
    return io, eo, xo, pho, fields


def _snapshot(ps):
    se, mk = ps.slot_info()
    cap = ps.capacity()
    live = mk.astype(bool)
    return (ps.member(2)[0, :cap][live], se[live], ps.member(0)[:, :cap][:, live],
            ps.member(4)[0, :cap][live])


def _check_union(snaps, owners, ref):
    io, eo, xo, pho, _ = ref
    ids = np.concatenate([s[0] for s in snaps])
    order = np.argsort(ids)
    assert np.array_equal(ids[order], io)                                   # nobody lost / duplicated
    assert np.array_equal(np.concatenate([s[1] for s in snaps])[order], eo)  # element ids bit-exact
    assert np.array_equal(np.concatenate([s[2] for s in snaps], axis=1)[:, order], xo)
    assert np.array_equal(np.concatenate([s[3] for s in snaps])[order], pho)
    for r, s in enumerate(snaps):
        assert np.all(owners[s[1]] == r)  # every rank holds only elements it owns


def _step(capi, mesh, ps, dim, deg, ids):
    if dim == 2:
        ids.fill_bytes(0xff)
        capi.push_search(mesh, ps, H, K, D, deg, ids, seeded=True, looplimit=200)
    else:
        capi.push_search(mesh, ps, H, K, D, deg, ids, seeded=False, looplimit=200)


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("fused", [False, True])
def test_migrate_local_virtual_ranks(ppo, synth, capi, dim, world, fused):
    """pp_ps_migrate_begin / _end on a `local` communicator: element-block virtual ranks of one
    process; union of the ranks == the single-structure oracle run, and the all-reduced scatter
    fields == the oracle's (gyroSync)."""
    pop = _population(synth, dim)
    ne = len(pop["e2v"])
    owners = (np.arange(ne) * world // ne).astype(np.int32)
    mesh = capi.Mesh(dim, pop["coords"], pop["e2v"], pop["cls"])
    ref = _oracle_run(ppo, pop, NSTEPS)
    comms = capi.Comm.local(world)
    assert [c.kind() for c in comms] == ["local"] * world
    ranks = [_rank_structure(capi, pop, owners, r) for r in range(world)]
    owners_d = capi.DevArray.from_host(owners)
    safes = [capi.DevArray.from_host((owners == r).astype(np.uint8)) for r in range(world)]
    fg, bg = capi.create_gyro_ring_mappings(mesh)
    moved = 0
    for step in range(NSTEPS):
        routes, fields = [], []
        for r, ps in enumerate(ranks):
            ids = capi.DevArray.from_host(np.full(max(ps.capacity(), 1), -1, dtype=np.int32))
            _step(capi, mesh, ps, dim, 6.0, ids)
            if not fused:
                capi.update_positions(ps)
            ne_d, np_d = capi.set_unsafe_procs(ps, ids, safes[r], owners_d, r)
            wf, wb = capi.DevArray(mesh.nverts, np.float64), capi.DevArray(mesh.nverts, np.float64)
            capi.migrate_begin(ps, ne_d, np_d, comms[r], commit=fused,
                               scatter=(mesh, [fg, bg], [wf, wb]) if fused else None)
            routes.append((ne_d, np_d))
            fields.append((wf, wb))
        for r, ps in enumerate(ranks):
            ns, nr = capi.migrate_end(ps, comms[r])
            moved += ns
        if fused:  # gyroSync: pack both fields, SUM over the virtual ranks
            packed = [capi.gyro_sync_pack(mesh.nverts, wf, wb) for wf, wb in fields]
            for r in range(world):
                comms[r].allreduce_sum(packed[r])
            fo, bo = ref[4][step]
            for r in range(world):
                got = packed[r].to_host()
                assert np.array_equal(got[0::2], fo) and np.array_equal(got[1::2], bo)
    assert moved > 0
    _check_union([_snapshot(ps) for ps in ranks], owners, ref)
    with pytest.raises(capi.PPError):  # one-call form needs every virtual rank to begin first
        capi.migrate(ranks[0], routes[0][0], routes[0][1], comms[0])
    for c in comms:
        c.destroy()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_migrate_send_right_then_back_reference_properties(synth, capi, world):
    """particle_structs/test/test_migrate.cpp, migrateSendRight (:4-101), on virtual ranks of one process.
    Every rank sends the particles of its LAST element to rank + 1 (`new_process(p) = (local_rank + 1) %
    local_csize` for `e == num_elems - 1`, :21-24) after stamping them with its own rank (`rnks(p) =
    local_rank`, :26).  The reference then asserts, per live particle (:40-51): in the last element no particle
    still carries the local rank ("Failed to send particle"), in any other element none carries a foreign one
    ("Incorrectly received particle").  Second half (:55-99): everything is sent back to the rank it is stamped
    with (`new_process(p) = rnks(p)`, :70) and `rank != local_rank` must hold for no particle (:82-86).  The
    particle counts those assertions imply (a ring: what leaves the last element arrives at the right
    neighbour's last element; the way back restores the original populations BY PARTICLE ID) are checked on
    top.  Stamp = member b (float, exact for small integers), id = member 2."""
    pop = _population(synth, 3)
    ne = len(pop["e2v"])
    n = len(pop["elem"])
    ranks, ids0 = [], []
    for r in range(world):  # every rank: the same elements (the reference runs one mesh per rank), own particle ids
        info = [a.copy() for a in pop["info"]]
        info[2] = (np.arange(n) + r * n).astype(np.int32)
        info[3] = np.full(n, float(r), dtype=np.float32)  # rnks(p) = local_rank
        ranks.append(capi.PS.scs(capi.PARTICLE_XGCM, ne, pop["ppe"], C_=64, gids=np.arange(ne, dtype=np.int64),
                                 particle_elements=pop["elem"], particle_info=info))
        ids0.append(np.sort(info[2]))
    comms = capi.Comm.local(world)
    last = int(np.flatnonzero(pop["ppe"])[-1])  # the reference's fixture has particles in element num_elems - 1;
    n_last = int(pop["ppe"][last])              # here: the last element that holds any

    def live(ps):
        se, mk = ps.slot_info()
        cap = ps.capacity()
        m = mk.astype(bool)
        return se[m], ps.member(2)[0, :cap][m], ps.member(3)[0, :cap][m].astype(np.int32)

    def migrate_all(dest_of):
        keep = []
        for r, ps in enumerate(ranks):
            se, mk = ps.slot_info()
            cap = max(ps.capacity(), 1)
            stamp = ps.member(3)[0, :cap].astype(np.int32)
            ne_h = np.where(mk > 0, se, -1).astype(np.int32)  # new_element(p) = e
            np_h = np.full(cap, r, dtype=np.int32)
            np_h[mk > 0] = dest_of(r, se[mk > 0], stamp[mk > 0])
            ne_d, np_d = capi.DevArray.from_host(ne_h), capi.DevArray.from_host(np_h)
            capi.migrate_begin(ps, ne_d, np_d, comms[r])
            keep.append((ne_d, np_d))
        return sum(capi.migrate_end(ps, comms[r])[0] for r, ps in enumerate(ranks))

    sent = migrate_all(lambda r, e, stamp: np.where(e == last, (r + 1) % world, r))
    assert sent == world * n_last
    for r, ps in enumerate(ranks):
        e, pid, stamp = live(ps)
        assert len(e) == n                                     # n_last left, n_last arrived
        assert not np.any((e == last) & (stamp == r))          # :42-46 "Failed to send particle"
        assert not np.any((e != last) & (stamp != r))          # :47-50 "Incorrectly received particle"
        assert np.all(stamp[e == last] == (r - 1) % world)     # they came from the left neighbour
    back = migrate_all(lambda r, e, stamp: stamp)              # new_process(p) = rnks(p), :70
    assert back == world * n_last
    for r, ps in enumerate(ranks):
        e, pid, stamp = live(ps)
        assert np.all(stamp == r)                              # :82-86 "Incorrectly received / failed to send"
        assert np.array_equal(np.sort(pid), ids0[r])           # everybody is home, nobody twice
    for c in comms:
        c.destroy()


def test_config5_step_eight_owner_blocks_vs_oracle(ppo, synth, capi):
    """BASELINE configs[4]'s time step at a size the oracle walks in seconds: a coarse torus split into
    EIGHT contiguous element blocks (bench.py's owner rule for 8 GPUs), every virtual rank calls what
    bench.py's c5 step calls -- pp_push_search (origins trusted from the second step on), then
    pp_migrate_ptcls_begin / pp_ps_migrate_end with the commit and both gyroScatter maps riding along,
    then gyroSync as pp_gyro_sync_pack + pp_allreduce_sum.  Union of the ranks and the synced fields
    equal the single-structure oracle run (migrate_lb_ptcls src/pumipic_ptcl_ops.hpp:53-85,
    SCS_migrate.h:29-213, gyroSync test/gyroScatter.hpp:231-258)."""
    pop = _population(synth, 3)
    ne, world = len(pop["e2v"]), 8
    from pumipic_amd import dist as ppdist
    owners = ppdist.element_block_owners(ne, world)
    mesh = capi.Mesh(3, pop["coords"], pop["e2v"], pop["cls"])
    ref = _oracle_run(ppo, pop, NSTEPS)
    comms = capi.Comm.local(world)
    ranks = [_rank_structure(capi, pop, owners, r) for r in range(world)]
    owners_d = capi.DevArray.from_host(owners)
    safes = [capi.DevArray.from_host((owners == r).astype(np.uint8)) for r in range(world)]
    fg, bg = capi.create_gyro_ring_mappings(mesh)
    moved = 0
    for step in range(NSTEPS):
        fields, keep = [], []
        for r, ps in enumerate(ranks):
            ids = capi.DevArray(max(ps.capacity(), 1), np.int32)
            capi.push_search(mesh, ps, H, K, D, 6.0, ids, seeded=False, looplimit=200)
            assert capi.push_search_counters()[2] == 0
            ps.set_origin_trust(True)
            wf, wb = capi.DevArray(mesh.nverts, np.float64), capi.DevArray(mesh.nverts, np.float64)
            capi.migrate_ptcls_begin(ps, ids, safes[r], owners_d, comms[r], commit=True,
                                     scatter=(mesh, [fg, bg], [wf, wb]))
            fields.append((wf, wb))
            keep.append(ids)
        for r, ps in enumerate(ranks):
            ns, nr = capi.migrate_end(ps, comms[r])
            moved += ns
        packed = [capi.gyro_sync_pack(mesh.nverts, wf, wb) for wf, wb in fields]
        for r in range(world):
            comms[r].allreduce_sum(packed[r])
        fo, bo = ref[4][step]
        for r in range(world):
            got = packed[r].to_host()
            assert np.array_equal(got[0::2], fo) and np.array_equal(got[1::2], bo)
    assert moved > 0
    _check_union([_snapshot(ps) for ps in ranks], owners, ref)
    for c in comms:
        c.destroy()


def test_migrate_new_particles_ride_along(ppo, synth, capi):
    """the caller's own new particles (new_particle_elements / new_particle_info of
    SellCSigma::migrate, SCS_migrate.h:198-206) enter the same rebuild as the received ones"""
    pop = _population(synth, 2)
    ne = len(pop["e2v"])
    world = 2
    owners = (np.arange(ne) * world // ne).astype(np.int32)
    comms = capi.Comm.local(world)
    ranks = [_rank_structure(capi, pop, owners, r) for r in range(world)]
    before = [ps.nPtcls() for ps in ranks]
    added = []
    for r, ps in enumerate(ranks):
        cap = max(ps.capacity(), 1)
        se, mk = ps.slot_info()
        ne_h = np.where(mk.astype(bool), se, -1).astype(np.int32)
        # route the first 7 live particles of rank r to the other rank (into an element it owns)
        live = np.flatnonzero(mk)[:7]
        tgt = int(np.flatnonzero(owners == 1 - r)[3])
        ne_h[live] = tgt
        np_h = np.full(cap, r, dtype=np.int32)
        np_h[live] = 1 - r
        own = int(np.flatnonzero(owners == r)[5])
        k = 4 + r
        info = [np.full((3, k), 0.25 + r), np.zeros((3, k)), np.arange(10**6 + 100 * r, 10**6 + 100 * r + k,
                                                                     dtype=np.int32),
                np.full(k, 0.5, np.float32), np.full(k, 0.125, np.float32)]
        added.append((own, k, info[2]))
        capi.migrate_begin(ps, capi.DevArray.from_host(ne_h), capi.DevArray.from_host(np_h), comms[r],
                           new_particles=(np.full(k, own, np.int32), info))
    for r, ps in enumerate(ranks):
        ns, nr = capi.migrate_end(ps, comms[r])
        assert ns == 7 and nr == 7
    for r, ps in enumerate(ranks):
        own, k, ids = added[r]
        assert ps.nPtcls() == before[r] + k
        pid, se, x, _ = _snapshot(ps)
        sel = np.isin(pid, ids)
        assert sel.sum() == k and np.all(se[sel] == own) and np.all(x[:, sel] == 0.25 + r)
        tgt = int(np.flatnonzero(owners == r)[3])
        assert (se == tgt).sum() >= 7
    for c in comms:
        c.destroy()


def test_rccl_single_rank(ppo, synth, capi):
    """RCCL behind the C-ABI with one rank: the library opens librccl, forms a communicator from
    its own unique id, and the collectives run on the library stream."""
    uid = capi.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = capi.Comm.rccl(uid, 0, 1)
    assert comm.kind() == "rccl" and comm.size() == 1
    v = np.arange(1000, dtype=np.float64) / 8
    d = capi.DevArray.from_host(v)
    comm.allreduce_sum(d)
    assert np.array_equal(d.to_host(), v)
    assert list(comm.allreduce_sum_host([3, 4])) == [3, 4]
    assert list(comm.exchange_counts([5])) == [5]
    comm.barrier()
    # migrate on one rank == rebuild (SCS_migrate.h:20-25), with commit + scatter folded in
    pop = _population(synth, 3)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg)
    ne = len(pop["e2v"])
    owners = capi.DevArray.from_host(np.zeros(ne, np.int32))
    safe = capi.DevArray.from_host(np.ones(ne, np.uint8))
    for _ in range(3):
        ppo.toroidal_push(po, mo, H, K, D, 6.0, trig=1)
        ids_o = ppo.search_mesh(mo, po, looplimit=200)["elem_ids"]
        ppo.update_positions(po)
        po.rebuild(ids_o)
        ids = capi.DevArray(max(pg.capacity(), 1), np.int32)
        capi.push_search(mg, pg, H, K, D, 6.0, ids, seeded=False, looplimit=200)
        ne_d, np_d = capi.set_unsafe_procs(pg, ids, safe, owners, 0)
        wf, wb = capi.DevArray(mg.nverts, np.float64), capi.DevArray(mg.nverts, np.float64)
        capi.migrate(pg, ne_d, np_d, comm, commit=True, scatter=(mg, [fg, bg], [wf, wb]))
        assert np.array_equal(wf.to_host(), ppo.gyro_scatter(mo, po, fo))
        assert np.array_equal(wb.to_host(), ppo.gyro_scatter(mo, po, bo))
    so, mko = po.slot_info()
    io, eo = common.by_id(po.member(2)[0, :po.capacity()], mko, so)
    pid, se, x, _ = _snapshot(pg)
    order = np.argsort(pid)
    assert np.array_equal(pid[order], io) and np.array_equal(se[order], eo)
    comm.destroy()


# ---------------------------------------------------------------- two processes, one GPU
def _proc_worker(rank, world, port, transport, dim, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import pumipic_amd_loader
        pp = pumipic_amd_loader.load()
        from pumipic_amd import capi
        capi.init(0)  # both ranks on device 0
        if transport == "gloo":
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(port)
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)
            comm = capi.Comm.torch()
        elif transport == "env-tcp":
            os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                              MASTER_PORT=str(port), PP_COMM="tcp", PP_COMM_PORT=str(port))
            comm = capi.Comm.env()
        else:
            comm = capi.Comm.tcp("127.0.0.1", port, rank, world)
        pop = _population(pp.synth, dim)
        ne = len(pop["e2v"])
        owners = (np.arange(ne) * world // ne).astype(np.int32)
        mesh = capi.Mesh(dim, pop["coords"], pop["e2v"], pop["cls"])
        ps = _rank_structure(capi, pop, owners, rank)
        owners_d = capi.DevArray.from_host(owners)
        safe = capi.DevArray.from_host((owners == rank).astype(np.uint8))
        fg, bg = capi.create_gyro_ring_mappings(mesh)
        wf, wb = capi.DevArray(mesh.nverts, np.float64), capi.DevArray(mesh.nverts, np.float64)
        fields = []
        for _ in range(NSTEPS):
            ids = capi.DevArray.from_host(np.full(max(ps.capacity(), 1), -1, dtype=np.int32))
            _step(capi, mesh, ps, dim, 6.0, ids)
            if transport == "tcp":  # setUnsafeProcs as its own pass, then SellCSigma::migrate
                ne_d, np_d = capi.set_unsafe_procs(ps, ids, safe, owners_d, rank)
                capi.migrate(ps, ne_d, np_d, comm, commit=True, scatter=(mesh, [fg, bg], [wf, wb]))
            else:                   # migrate_lb_ptcls as one call (the routing rule rides in the pack)
                capi.migrate_ptcls(ps, ids, safe, owners_d, comm, commit=True, scatter=(mesh, [fg, bg], [wf, wb]))
            packed = capi.gyro_sync_pack(mesh.nverts, wf, wb)
            comm.allreduce_sum(packed)
            fields.append(packed.to_host())
        total = int(comm.allreduce_sum_host([ps.nPtcls()])[0])
        comm.barrier()
        q.put((rank, "ok", _snapshot(ps), fields, total))
        comm.destroy()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + repr(e) + traceback.format_exc(), None, None, None))


@pytest.mark.parametrize("transport,dim,world", [("tcp", 2, 2), ("tcp", 3, 2), ("gloo", 2, 2), ("env-tcp", 3, 2),
                                                 ("tcp", 2, 4), ("gloo", 3, 3)])
def test_two_processes_share_one_gpu(ppo, synth, capi, transport, dim, world):
    """Two (to four) rank PROCESSES on one GPU run the c5 step through pp_ps_migrate_scatter +
    pp_allreduce_sum over a host-staged transport; union == single-structure oracle run.  (Three and more
    ranks: not every rank takes part in every step's migration -- the exchange stays a collective.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_proc_worker, args=(r, world, port, transport, dim, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    pop = _population(synth, dim)
    ne = len(pop["e2v"])
    owners = (np.arange(ne) * world // ne).astype(np.int32)
    ref = _oracle_run(ppo, pop, NSTEPS)
    _check_union([r[2] for r in res], owners, ref)
    for r in res:
        assert r[4] == len(ref[0])
        for step in range(NSTEPS):
            fo, bo = ref[4][step]
            assert np.array_equal(r[3][step][0::2], fo) and np.array_equal(r[3][step][1::2], bo)


@pytest.mark.parametrize("world", [2, 4])
def test_cpp_driver_pseudoxgcm_on_picparts(synth, capi, tmp_path, world):
    """The same driver with PP_PARTS=3:1: PICparts from a pumipic::Input (BFS buffer of 3 layers, safe zone of
    1) as the reference's pseudoXGCm builds them -- every rank on its part's own mesh, migrate_lb_ptcls
    routing by the part's safe tag / owners, particles travelling with element gids (the structure's own
    gid -> element table), gyroScatter on the part, gyroSync through the owners (reduceCommArray fan-in /
    fan-out).  No particle is lost, and the synced field, every vertex counted once by its owner, carries
    the same mass as the run on the full-mesh replica."""
    import re
    import subprocess
    drv = os.path.join(ROOT, "pumi-pic_amd", "drivers")
    subprocess.check_call(["make", "-C", drv, "-s"])
    c, e, cl = synth.annulus_tri(n_b=24, n_theta=96, band_width=3)
    mesh_file = str(tmp_path / "annulus.bin")
    synth.write_mesh_bin(mesh_file, 2, c, e, cl)
    npt = 20000
    results = {}
    for mode in ("replica", "parts"):
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), PP_DEVICE="0",
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PP_COMM="tcp", PP_COMM_PORT=str(port))
            if mode == "parts":
                env["PP_PARTS"] = "3:1"
            procs.append(subprocess.Popen([os.path.join(drv, "pseudoXGCm"), mesh_file, str(npt), "6", "10", "2.0", "1"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=300) for p in procs]
        for p, (so, se) in zip(procs, outs):
            assert p.returncode == 0, (mode, so[-1500:], se[-2000:])
        m = re.search(r"RESULT particles (\d+) scatter_mass (\S+) touched_elements (\d+)", outs[0][0])
        assert m, outs[0][0][-2000:]
        per_rank = [int(re.search(r"RANK %d particles (\d+)" % r, outs[r][0]).group(1)) for r in range(world)]
        created = npt if world == 2 else 3 * (npt // world)  # (4 ranks: the outer block holds no source element)
        assert int(m.group(1)) == created == sum(per_rank), (mode, per_rank)
        results[mode] = (float(m.group(2)), per_rank)
        if mode == "parts":
            assert "PICparts from an Input" in outs[0][0]
    assert results["parts"][0] == results["replica"][0]      # integer-valued sums: exact
    # who holds a particle differs: the parts' safe zone is one layer wider than the replica's own-block rule,
    # and migrate_lb_ptcls balances on parts (the outer block of the 4-rank run creates no particle: it is fed
    # through the elements it shares a safe zone with)
    print("per rank: replica", results["replica"][1], "parts", results["parts"][1])
    if world == 4:
        assert results["parts"][1][3] > results["replica"][1][3]
        assert max(results["parts"][1]) <= max(results["replica"][1])


def test_cpp_driver_pseudoxgcm_two_ranks(synth, capi, tmp_path):
    """The pseudoXGCm driver on the mirror headers (Mesh::partition, migrate_lb_ptcls -> ParticleStructure::
    migrate -> pp_ps_migrate, gyroSync -> reduceCommArray -> pp_allreduce_sum, SummarizeTimeAcrossProcesses)
    as TWO rank processes sharing the GPU over PP_COMM=tcp: no particle is lost or duplicated across the
    migrations, and the synced scatter field carries every particle's contribution."""
    import re
    import subprocess
    drv = os.path.join(ROOT, "pumi-pic_amd", "drivers")
    subprocess.check_call(["make", "-C", drv, "-s"])
    c, e, cl = synth.annulus_tri(n_b=24, n_theta=96, band_width=3)
    mesh_file = str(tmp_path / "annulus.bin")
    synth.write_mesh_bin(mesh_file, 2, c, e, cl)
    npt, world = 20000, 2
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), PP_DEVICE="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PP_COMM="tcp", PP_COMM_PORT=str(port))
        procs.append(subprocess.Popen([os.path.join(drv, "pseudoXGCm"), mesh_file, str(npt), "6", "10", "2.0", "1"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    m = re.search(r"RESULT particles (\d+) scatter_mass (\S+) touched_elements (\d+)", outs[0][0])
    assert m, outs[0][0][-2000:]
    per_rank = [int(re.search(r"RANK %d particles (\d+)" % r, outs[r][0]).group(1)) for r in range(world)]
    assert int(m.group(1)) == npt == sum(per_rank)          # interior bands: nobody leaves the domain
    assert all(n > 0 for n in per_rank)
    mass = float(m.group(2))
    assert 0.9 * 18 * npt <= mass <= 18 * npt               # 2 rings x 3 verts x (8 pts x 3 mapped)/8 each
    assert "world ranks 2 (tcp)" in outs[0][0]
    assert "Reduced Timing Summary" in outs[0][1] and "gyro reduction" in outs[0][1]


def test_cpp_driver_distributor_rank_subset_is_enforced(synth, capi, tmp_path):
    """support/psDistributor.hpp:10-138, the rank-subset form: the driver lists itself + the buffered ranks of
    its part (the two-rank test above runs with that list); with PP_DIST_SELF_ONLY=1 it lists nobody else, and
    the first migration that has a particle for the other rank is refused by ParticleStructure::migrate (the
    reference's Distributor::index() is undefined for a rank that is not listed) -- on EVERY rank: the ranks
    exchange their verdicts before anyone leaves, so no peer is left waiting in the exchange."""
    import subprocess
    drv = os.path.join(ROOT, "pumi-pic_amd", "drivers")
    subprocess.check_call(["make", "-C", drv, "-s"])
    c, e, cl = synth.annulus_tri(n_b=24, n_theta=96, band_width=3)
    mesh_file = str(tmp_path / "annulus.bin")
    synth.write_mesh_bin(mesh_file, 2, c, e, cl)
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), PP_DEVICE="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PP_COMM="tcp", PP_COMM_PORT=str(port),
                   PP_COMM_TIMEOUT="20", PP_DIST_SELF_ONLY="1")
        procs.append(subprocess.Popen([os.path.join(drv, "pseudoXGCm"), mesh_file, "20000", "6", "10", "2.0", "1"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]  # (no rank is left waiting: the verdict is collective)
    assert any("which the Distributor does not list" in se for _, se in outs), [se[-500:] for _, se in outs]
    for p, (so, se) in zip(procs, outs):  # EVERY rank stops, with a non-zero exit code, on its own
        assert p.returncode not in (0, None), se[-500:]
        assert "Distributor rank subset" in se, se[-500:]


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_multi_rank_line_rehearsal(tmp_path, launcher):
    """bench.py's N > 1 line (the migrating c5 workload: rank start-up under torch.distributed.run, element-block
    owners, pp_migrate_ptcls, gyroSync all-reduce, the max-over-ranks timing, ONE JSON line from rank 0) rehearsed
    with two ranks on this box's single GPU: PP_BENCH_REHEARSAL=1 puts every rank on GPU 0, --comm tcp carries the
    exchange and gloo the timing barrier (RCCL cannot place two ranks on one device).  Not a measurement -- the
    line says so -- but every line of the multi-rank plumbing except the RCCL calls themselves runs."""
    import json
    import subprocess
    port = _free_port()
    env = dict(os.environ, PP_BENCH_REHEARSAL="1", PP_BENCH_PREWARM="0", PP_COMM_PORT=str(_free_port()),
               PP_BENCH_CACHE=str(tmp_path))
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--workload", "c5",
            "--mesh", "100k", "--particles", "300000", "--comm", "tcp", "--no-cpu-baseline", "--deg", "4.0"]
    if launcher == "self":      # (this variant also takes the strong-scaling split: 600 000 over the two ranks)
        i = args.index("--particles")
        args[i:i + 2] = ["--scaling", "strong", "--total-particles", "600000"]
    if launcher == "torchrun":  # the driver's way
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + args
    else:                       # `python bench.py --gpus 2`: the parent starts the ranks itself
        cmd = [sys.executable] + args
        env["PP_BENCH_ASSUME_GPUS"] = "2"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 2 and "rehearsal" in j
    assert j["scaling"] == ("strong" if launcher == "self" else "weak")
    assert j["metric"].startswith("particles pushed+searched+scattered")
    assert "migrate" in j["config"]["workload"] and "tcp" in j["config"]["workload"]
    # the pre-flight ran (checked exchange + all-reduce through the migration's own calls) and the watchdog is armed
    assert j["preflight"]["ok"] and j["preflight"]["transport"] == "tcp" and j["watchdog_s"] == 120
    assert j["rank0_sent_per_step"] > 0                       # particles really crossed between the ranks
    # whole-job particles / time (the population draws land within a fraction of a per cent of 2 x 300 000)
    assert abs(j["value"] - 600000 * 4 / (j["ms_per_step"] * 4e-3)) / j["value"] < 1e-2


def test_bench_watchdog_turns_a_stalled_rank_into_exit_code_3(tmp_path):
    """a rank that stops making progress (here: rank 1 sleeps inside its second step, PP_BENCH_STALL_RANK) must
    END the job with a non-zero exit code within the watchdog's limit -- not hold it until somebody's timeout.
    Rank 0 then sits in the exchange with a silent peer and its own watchdog fires; bench.py's parent relays
    code 3."""
    import subprocess
    import time
    env = dict(os.environ, PP_BENCH_REHEARSAL="1", PP_BENCH_PREWARM="0", PP_COMM_PORT=str(_free_port()),
               PP_BENCH_CACHE=str(tmp_path), PP_BENCH_ASSUME_GPUS="2", PP_BENCH_STALL_RANK="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--workload", "c5", "--mesh", "100k", "--particles", "200000", "--comm", "tcp", "--no-cpu-baseline",
           "--deg", "4.0", "--watchdog", "8"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "made no progress" in r.stderr and "exit code 3" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no line from a job that failed
    assert time.time() - t0 < 300


@pytest.mark.parametrize("order", ["torch_first", "library_first"])
def test_rccl_pairs_with_the_librarys_own_hip_runtime(order):
    """PyTorch bundles a second ROCm stack (libamdhip64 / libhsa-runtime64 / librccl under torch/lib).  Whatever
    the import order, the library must open the librccl that sits next to the HIP runtime it is itself bound to:
    an RCCL from the other stack finds its HSA runtime uninitialised ("no ROCm-capable device is detected") or,
    worse, is handed streams and pointers of a runtime it does not know.  One-rank communicator + the checked
    exchange in a fresh process per order."""
    import subprocess
    code = r"""
import sys
sys.path.insert(0, %r)
order = %r
if order == "torch_first":
    import torch
    torch.cuda.set_device(0)
    torch.zeros(4, device="cuda").sum().item()     # torch's own HIP runtime is live
import pumipic_amd_loader
pumipic_amd_loader.load()
from pumipic_amd import capi
capi.init(0)
if order == "library_first":
    import torch                                     # torch's stack arrives after ours
    import torch.distributed                         # (loads torch's librccl; its HIP runtime stays idle)
import numpy as np
comm = capi.Comm.rccl(capi.Comm.unique_id(), 0, 1)
assert comm.kind() == "rccl"
comm.selftest(3)
d = capi.DevArray.from_host(np.arange(16, dtype=np.float64))
comm.allreduce_sum(d)
assert np.array_equal(d.to_host(), np.arange(16, dtype=np.float64))
comm.destroy()
print("ok")
""" % (ROOT, order)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout.split(), (r.returncode, r.stdout[-500:], r.stderr[-3000:])
