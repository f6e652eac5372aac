"""-m gpu: PICpart construction and comm-array reduction through the C-ABI (pp_picpart_*) against the
oracle (oracle/ppo_picpart.py): numberings, part meshes and exchange plans bit-exact, reductions bit-exact
for integers and for doubles (the owner adds in the same order on both sides: own value, then increasing
rank).  Ranks: the virtual ranks of one process (Comm.local) and two rank PROCESSES sharing the GPU over
the TCP transport; what the reference's tests assert (test/test_comm_array.cpp, test_input_construct.cpp)
is asserted on the GPU results too."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

import pumipic_amd_loader
from test_picpart_oracle import slab_owners

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi(pp):
    from pumipic_amd import capi as c
    c.build()
    c.init(0)
    return c


@pytest.fixture(scope="module")
def opp():
    return pumipic_amd_loader.load_oracle_picpart()


def _mesh_arrays(synth, which):
    if which == 0:
        c, e, k = synth.kuhn_box(5)
        return 3, c, e, k, slab_owners(c, e, 4)
    if which == 1:
        c, e, k = synth.annulus_tri(n_b=8, n_theta=32, band_width=3)
        return 2, c, e, k, slab_owners(c, e, 4, axis=1)
    c, e, k = synth.torus_tet(n_b=6, n_theta=16, n_planes=8)
    return 3, c, e, k, slab_owners(c, e, 3, axis=2)


CASES = [  # (mesh, buffer, safe, bridge (0 = vertices, 1 = sides), buffer layers, safe layers)
    (0, "BFS", "BFS", 0, 1, 0), (1, "BFS", "BFS", 0, 1, 0), (0, "FULL", "FULL", 0, 3, 1), (1, "FULL", "BFS", 1, 3, 3),
    (0, "MINIMUM", "NONE", 0, 3, 1), (2, "BFS", "FULL", 0, 2, 1), (2, "BFS", "BFS", 1, 2, 1), (1, "MINIMUM", "BFS", 0, 3, 2),
]


def _build(ppo, synth, capi, opp, case):
    which, bm, sm, bridge, bl, sl = case
    dim, c, e, k, owner = _mesh_arrays(synth, which)
    P = int(owner.max()) + 1
    mo = ppo.Mesh(dim, c, e, k)
    bd = 0 if bridge == 0 else dim - 1
    O = opp.PicParts(mo, owner, P, getattr(opp, bm), getattr(opp, sm), bridge_dim=bd, buffer_layers=bl, safe_layers=sl)
    mg = capi.Mesh(dim, c, e, k)
    comms = capi.Comm.local(P)
    parts = [capi.PicPart(mg, owner, comms[r], getattr(capi, "PART_" + bm), getattr(capi, "PART_" + sm), bd, bl, sl)
             for r in range(P)]
    return dim, mo, mg, O, parts, comms, owner


@pytest.mark.parametrize("case", CASES)
def test_construction_matches_oracle(ppo, synth, capi, opp, case):
    dim, mo, mg, O, parts, comms, owner = _build(ppo, synth, capi, opp, case)
    for po, pg in zip(O.parts, parts):
        assert pg.is_full_mesh == po.is_full_mesh
        assert pg.num_buffers == int(po.has_part.sum())
        for d in range(dim + 1):  # every entity dimension (test/test_comm_array.cpp:48-66), tet edges included
            assert pg.nents[d] == po.nents[d]
            assert np.array_equal(pg.array(capi.PART_GIDS, d), po.gids[d])
            assert np.array_equal(pg.array(capi.PART_OWNERS, d), po.owners[d])
            assert np.array_equal(pg.array(capi.PART_RANK_LIDS, d), po.rank_lids[d])
            assert np.array_equal(pg.array(capi.PART_COMM_INDEX, d), po.comm_index[d])
            assert np.array_equal(pg.array(capi.PART_FULL_IDS, d), po.full_ids[d])
            assert np.array_equal(pg.array(capi.PART_ENT_IDS, d), po.ent_ids[d])
            assert np.array_equal(pg.nents_offsets(d), po.nents_offsets[d])
            assert np.array_equal(pg.complete_parts(d), po.is_complete[d])
            assert pg.buffered_ranks(d).tolist() == po.buffered_parts[d]
        assert np.array_equal(pg.array(capi.PART_SAFE).astype(np.int32), po.safe)
        # the part's mesh: same vertices and elements as the oracle's, a working pp_mesh
        m = pg.mesh
        assert (m.nverts, m.nelems) == (po.nents[0], po.nents[dim])
        if not pg.is_full_mesh:
            assert np.array_equal(m.array(0).reshape(-1, dim), po.coords)          # PP_MESH_COORDS
            assert np.array_equal(m.array(1).reshape(-1, dim + 1), po.elem2verts)   # PP_MESH_ELEM2VERTS
    for c in comms:
        c.destroy()


@pytest.mark.parametrize("which", [0, 2])
def test_full_mesh_parts_reference_checks(ppo, synth, capi, opp, which):
    """test/test_full_mesh.cpp:32-59 on `Input(mesh, owners, FULL, FULL)`: every rank's picpart has the entity counts
    of the serial mesh in EVERY dimension (:36-42) and its elements carry the serial mesh's global ids in the serial
    order -- the `global_serial` tag of `:45-57`: the part's full-mesh ids are the identity -- and every element is
    safe.  The 8-GPU configuration starts from exactly these parts."""
    dim, c, e, k, owner = _mesh_arrays(synth, which)
    P = int(owner.max()) + 1
    mg = capi.Mesh(dim, c, e, k)
    comms = capi.Comm.local(P)
    parts = [capi.PicPart(mg, owner, comms[r], capi.PART_FULL, capi.PART_FULL, 0, 3, 1) for r in range(P)]
    mo = ppo.Mesh(dim, c, e, k)
    O = opp.PicParts(mo, owner, P, opp.FULL, opp.FULL, bridge_dim=0, buffer_layers=3, safe_layers=1)
    serial = [len(c) if d == 0 else len(e) if d == dim else O.parts[0].nents[d] for d in range(dim + 1)]
    for r, pg in enumerate(parts):
        assert pg.is_full_mesh
        for d in range(dim + 1):
            assert pg.nents[d] == serial[d], (r, d)                                    # :36-42
            assert np.array_equal(pg.array(capi.PART_FULL_IDS, d), np.arange(serial[d]))   # :45-57, every dimension
            # (the PICpart's own global numbering goes owner by owner: a permutation, the same on every rank)
            assert np.array_equal(np.sort(pg.array(capi.PART_GIDS, d)), np.arange(serial[d]))
            assert np.array_equal(pg.array(capi.PART_GIDS, d), parts[0].array(capi.PART_GIDS, d))
        assert pg.array(capi.PART_SAFE).all()
        assert np.array_equal(pg.array(capi.PART_OWNERS, dim), owner)
        assert (pg.mesh.nverts, pg.mesh.nelems) == (len(c), len(e))
    for cm in comms:
        cm.destroy()


def test_tet_edges_match_oracle(ppo, synth, capi, opp):
    """entity dimension 1 of a tet mesh (Omega_h ask_down(3,1) / ask_up(1,3)): PP_MESH_ELEM2EDGES /
    EDGE2VERTS / EDGE2ELEMS equal the oracle's first-seen derivation; every edge is an edge of its tets"""
    c, e, k = synth.torus_tet(n_b=4, n_theta=12, n_planes=6)
    mg = capi.Mesh(3, c, e, k)
    ev, e2e, off, up = opp.tet_edges(e)
    assert mg.num_edges() == len(ev)
    assert np.array_equal(mg.array(15).reshape(-1, 2), ev)       # PP_MESH_EDGE2VERTS
    assert np.array_equal(mg.array(14).reshape(-1, 6), e2e)      # PP_MESH_ELEM2EDGES
    assert np.array_equal(mg.array(16), off)                     # PP_MESH_EDGE2ELEMS_OFF
    got_up, got_off = mg.array(17), mg.array(16)
    for ed in range(0, len(ev), 11):
        assert sorted(got_up[got_off[ed]:got_off[ed + 1]].tolist()) == sorted(up[off[ed]:off[ed + 1]].tolist())
        for el in got_up[got_off[ed]:got_off[ed + 1]]:
            assert set(ev[ed].tolist()) <= set(np.asarray(e).reshape(-1, 4)[el].tolist())


def _reduce_both(capi, O, parts, d, op, host_arrays):
    devs = [capi.DevArray.from_host(a) for a in host_arrays]
    capi.picpart_reduce_all(parts, d, op, devs)
    return [x.to_host() for x in devs], O.reduce(d, op, host_arrays)


@pytest.mark.parametrize("case", CASES)
def test_reduce_matches_oracle_and_reference_properties(ppo, synth, capi, opp, case):
    dim, mo, mg, O, parts, comms, owner = _build(ppo, synth, capi, opp, case)
    rng = np.random.default_rng(5)
    P = len(parts)
    for d in range(dim + 1):
        n = [p.nents[d] for p in parts]
        # random doubles, 3 values per entity: SUM / MAX / MIN / BCAST bit-exact
        for op in (capi.OP_SUM, capi.OP_MAX, capi.OP_MIN, capi.OP_BCAST):
            arrs = [rng.standard_normal(k * 3) for k in n]
            got, want = _reduce_both(capi, O, parts, d, op, arrs)
            for g, w in zip(got, want):
                assert np.array_equal(g, w), (d, op)
        arrs = [rng.integers(-1000, 1000, size=k).astype(np.int32) for k in n]
        got, want = _reduce_both(capi, O, parts, d, capi.OP_SUM, arrs)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        # minOwnership (test_comm_array.cpp:120-146)
        arrs = [np.where(p.array(capi.PART_OWNERS, d) == r, r, np.iinfo(np.int32).max).astype(np.int32)
                for r, p in enumerate(parts)]
        got, _ = _reduce_both(capi, O, parts, d, capi.OP_MIN, arrs)
        for p, g in zip(parts, got):
            assert np.array_equal(g, p.array(capi.PART_OWNERS, d))
    # sumEntities on vertices (:148-179)
    cnt, _ = _reduce_both(capi, O, parts, 0, capi.OP_SUM, [np.ones(p.nents[0], np.int32) for p in parts])
    got, _ = _reduce_both(capi, O, parts, 0, capi.OP_SUM, [np.repeat(1.0 / c, 3) for c in cnt])
    for g in got:
        assert np.all(np.abs(g - 1.0) < 1e-5)
    # elements: 1 on the owner only (:92-117); full buffers: every entity counted comm_size times (:181-207)
    got, _ = _reduce_both(capi, O, parts, dim, capi.OP_SUM,
                          [np.repeat((p.array(capi.PART_OWNERS, dim) == r).astype(np.int32), 3)
                           for r, p in enumerate(parts)])
    for g in got:
        assert np.all(g == 1)
    if parts[0].is_full_mesh:
        for d in range(dim + 1):
            got, _ = _reduce_both(capi, O, parts, d, capi.OP_SUM, [np.ones(p.nents[d], np.int32) for p in parts])
            for g in got:
                assert np.all(g == P)
    for c in comms:
        c.destroy()


def test_gyro_sync_on_bfs_parts_equals_the_full_mesh_field(ppo, synth, capi, opp):
    """gyroSync's use (test/gyroScatter.hpp:231-258) without the full-mesh replica: every rank scatters the
    particles of its own elements on its PART's mesh; the SUM over the parts' vertex arrays equals the
    field of the whole population on the whole mesh, at every vertex a part holds."""
    import common
    pop = common.population_2d(synth, n_b=12, n_theta=48, num_ptcls=4000, mdl_face=3, band_width=3)
    dim, P = 2, 3
    owner = slab_owners(pop["coords"], pop["e2v"], P, axis=0)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    counts_full = np.bincount(pop["elem"], minlength=len(pop["e2v"])).astype(np.int32)
    mg = capi.Mesh(dim, pop["coords"], pop["e2v"], pop["cls"])
    comms = capi.Comm.local(P)
    parts = [capi.PicPart(mg, owner, comms[r], capi.PART_BFS, capi.PART_BFS, 0, 2, 1) for r in range(P)]
    # per-element counts as an element comm array: each rank contributes its own elements' particles
    arrs = []
    for r, p in enumerate(parts):
        fid = p.array(capi.PART_FULL_IDS, dim)
        own = p.array(capi.PART_OWNERS, dim) == r
        arrs.append(capi.DevArray.from_host(np.where(own, counts_full[fid], 0).astype(np.int32)))
    capi.picpart_reduce_all(parts, dim, capi.OP_SUM, arrs)
    for p, a in zip(parts, arrs):
        assert np.array_equal(a.to_host(), counts_full[p.array(capi.PART_FULL_IDS, dim)])
    # vertex field: each rank's partial sums of a per-element quantity onto its vertices, reduced
    w_full = np.zeros(mg.nverts)
    np.add.at(w_full, pop["e2v"].ravel(), np.repeat(counts_full.astype(np.float64), dim + 1))
    varrs = []
    for r, p in enumerate(parts):
        fid_e = p.array(capi.PART_FULL_IDS, dim)
        own = p.array(capi.PART_OWNERS, dim) == r
        e2v = p.mesh.array(1).reshape(-1, dim + 1)
        w = np.zeros(p.nents[0])
        np.add.at(w, e2v[own].ravel(), np.repeat(counts_full[fid_e][own].astype(np.float64), dim + 1))
        varrs.append(capi.DevArray.from_host(w))
    capi.picpart_reduce_all(parts, 0, capi.OP_SUM, varrs)
    for p, a in zip(parts, varrs):
        assert np.array_equal(a.to_host(), w_full[p.array(capi.PART_FULL_IDS, 0)])  # integers in doubles: exact
    for c in comms:
        c.destroy()


def test_error_paths(synth, capi):
    dim, c, e, k, owner = _mesh_arrays(synth, 1)
    mg = capi.Mesh(dim, c, e, k)
    comms = capi.Comm.local(4)
    with pytest.raises(capi.PPError):
        capi.PicPart(mg, np.full(len(e), 7, np.int32), comms[0])          # owner out of range
    with pytest.raises(capi.PPError):
        capi.PicPart(mg, owner, comms[0], capi.PART_BFS, capi.PART_BFS, bridge_dim=5)
    p = capi.PicPart(mg, owner, comms[0], capi.PART_BFS, capi.PART_BFS, 0, 1, 0)
    a = p.create_comm_array(0, 1, 0.0)
    with pytest.raises(capi.PPError):
        p.reduce(0, capi.OP_SUM, a)                                        # virtual ranks need the phases
    with pytest.raises(capi.PPError):
        p.reduce_mid()                                                     # nothing in flight
    with pytest.raises(capi.PPError):
        p.reduce_begin(5, capi.OP_SUM, a)                                  # no such entity dimension
    assert np.array_equal(capi.owner_by_classification(mg, np.arange(int(k.max()) + 1) % 4, int(k[0]) % 4),
                          (np.asarray(k) % 4).astype(np.int32))
    with pytest.raises(capi.PPError):
        capi.owner_by_classification(mg, np.zeros(int(k.max()) + 1, np.int32), 3)  # a rank that owns nothing
    for cm in comms:
        cm.destroy()


# ---------------------------------------------------------------- two processes, one GPU, TCP transport
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _proc_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import pumipic_amd_loader as L
        pp = L.load()
        from pumipic_amd import capi
        capi.init(0)
        comm = capi.Comm.tcp("127.0.0.1", port, rank, world)
        c, e, k = pp.synth.kuhn_box(5)
        from test_picpart_oracle import slab_owners as so
        owner = so(c, e, world)
        mesh = capi.Mesh(3, c, e, k)
        part = capi.PicPart(mesh, owner, comm, capi.PART_BFS, capi.PART_BFS, 0, 1, 0)
        out = {}
        rng = np.random.default_rng(100 + rank)
        for d in (0, 3):
            a = rng.standard_normal(part.nents[d] * 2)
            dev = capi.DevArray.from_host(a)
            part.reduce(d, capi.OP_SUM, dev)
            out[d] = (a, dev.to_host())
        comm.barrier()
        q.put((rank, "ok", out))
        comm.destroy()
    except Exception as ex:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + repr(ex) + traceback.format_exc(), None))


def test_two_processes_reduce_over_tcp(ppo, synth, capi, opp):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_proc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    c, e, k = synth.kuhn_box(5)
    owner = slab_owners(c, e, world)
    O = opp.PicParts(ppo.Mesh(3, c, e, k), owner, world, opp.BFS, opp.BFS, buffer_layers=1, safe_layers=0)
    for d in (0, 3):
        want = O.reduce(d, opp.SUM_OP, [res[r][2][d][0] for r in range(world)])
        for r in range(world):
            assert np.array_equal(res[r][2][d][1], want[r])


@pytest.mark.parametrize("world", [1, 3])
def test_cpp_driver_comm_array(synth, capi, tmp_path, world):
    """drivers/comm_array.cpp: the reference's test_comm_array.cpp checks on the C++ mirror (pumipic::Input,
    Mesh(Input&), createCommArray, reduceCommArray), as rank processes sharing the GPU over PP_COMM=tcp"""
    import subprocess
    drv = os.path.join(ROOT, "pumi-pic_amd", "drivers")
    subprocess.check_call(["make", "-C", drv, "-s"])
    c, e, k = synth.kuhn_box(5)
    mesh_file = str(tmp_path / "box.bin")
    synth.write_mesh_bin(mesh_file, 3, c, e, k)
    ptn = str(tmp_path / "box.ptn")
    np.savetxt(ptn, slab_owners(c, e, world), fmt="%d")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), PP_DEVICE="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PP_COMM="tcp", PP_COMM_PORT=str(port))
        procs.append(subprocess.Popen([os.path.join(drv, "comm_array"), mesh_file, ptn, "1", "0"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1500:], se[-1500:])
        assert "all checks passed" in so


# ---------------------------------------------------------------- particle load balancer
def _lb_setup(ppo, synth, capi, opp, which, P=4):
    dim, c, e, k, owner = _mesh_arrays(synth, which)
    owner = slab_owners(c, e, P, axis=0 if which != 1 else 1)
    mo = ppo.Mesh(dim, c, e, k)
    O = opp.PicParts(mo, owner, P, opp.BFS, opp.FULL, buffer_layers=3, safe_layers=1)  # test/test_lb.cpp:61-63
    mg = capi.Mesh(dim, c, e, k)
    comms = capi.Comm.local(P)
    parts = [capi.PicPart(mg, owner, comms[r], capi.PART_BFS, capi.PART_FULL, 0, 3, 1) for r in range(P)]
    return dim, O, opp.Balancer(O), parts, [capi.Balancer(p) for p in parts], comms


@pytest.mark.parametrize("which", [0, 1])
def test_balancer_partition_matches_oracle(ppo, synth, capi, opp, which):
    """ParticleBalancer::partition on an array of particles per element (test_lb.cpp:77-133): sbars and plan
    equal the oracle's, the destinations carry exactly the plan's amounts, imbalance <= 1.3 afterwards"""
    dim, O, ob, parts, bals, comms = _lb_setup(ppo, synth, capi, opp, which)
    P = len(parts)
    for r, (p, b) in enumerate(zip(parts, bals)):
        assert b.sbars().tolist() == ob.masks
        assert np.array_equal(b.sbar_ids(), ob.part_index[r])
    ppe = [np.full(p.nents[dim], (r + 1) * 50, dtype=np.int32) for r, p in enumerate(parts)]
    plan_o, W_o, w_o = ob.partition_counts(ppe, 1.05)
    for b, x in zip(bals, ppe):
        b.partition_begin(x)
    procs = [b.partition_end(1.05) for b in bals]
    arriving = np.zeros(P, dtype=np.int64)
    for r, (b, pr) in enumerate(zip(bals, procs)):
        plan, W = b.last_plan()
        assert plan == plan_o[r] and W.tolist() == W_o
        sent = np.bincount(pr[pr != r], minlength=P)
        want = np.zeros(P, dtype=np.int64)
        for _, q, t in plan:
            want[q] += t
        assert np.array_equal(sent, want)
        # a particle only goes where its element is safe
        elem_of = np.repeat(np.arange(parts[r].nents[dim]), ppe[r])
        moved = pr != r
        m = np.asarray(ob.masks, dtype=np.uint64)[ob.part_index[r][elem_of[moved]]]
        assert np.all((m >> pr[moved].astype(np.uint64)) & np.uint64(1))
        arriving += np.bincount(pr, minlength=P)
    assert arriving.max() * P / arriving.sum() <= 1.3
    for c in comms:
        c.destroy()


@pytest.mark.parametrize("which", [0, 1])
def test_balancer_repartition_then_migrate(ppo, synth, capi, opp, which):
    """testBalancePS (test_lb.cpp:135-213): 100 particles per element on the even ranks only; two rounds of
    repartition + migrate; imbalance <= 1.5 at the end, no particle lost, every particle in an element that
    is safe on (or owned by) the rank that holds it; weights and plan equal the oracle's each round"""
    dim, O, ob, parts, bals, comms = _lb_setup(ppo, synth, capi, opp, which)
    P = len(parts)
    members = [(np.int32, 1)]
    structs = []
    for r, p in enumerate(parts):
        ne = p.nents[dim]
        ppe = np.full(ne, 100 if r % 2 == 0 else 0, dtype=np.int32)
        elem = np.repeat(np.arange(ne, dtype=np.int32), ppe)
        ids = (np.arange(len(elem), dtype=np.int32) + r * 10_000_000)[None, :]
        structs.append(capi.PS.scs(members, ne, ppe, C_=32, gids=p.array(capi.PART_GIDS, dim),
                                   particle_elements=elem, particle_info=[ids]))
    total = sum(s.nPtcls() for s in structs)
    g2l = []
    for p in parts:  # gid -> element of the part (the reference's UnorderedMap, SCS_migrate.h:181-187)
        t = np.full(O.mesh.nelems, -1, dtype=np.int32)
        t[p.array(capi.PART_GIDS, dim)] = np.arange(p.nents[dim], dtype=np.int32)
        g2l.append(capi.DevArray.from_host(t))
    for rnd in range(2):
        ne_d, np_d, host = [], [], []
        for r, (p, s) in enumerate(zip(parts, structs)):
            se, mk = s.slot_info()
            safe = p.array(capi.PART_SAFE).astype(bool)
            own = p.array(capi.PART_OWNERS, dim)
            live = mk.astype(bool)
            ne_h = np.where(live, se, -1).astype(np.int32)
            np_h = np.full(len(se), r, dtype=np.int32)
            np_h[live] = np.where(safe[se[live]], r, own[se[live]])  # balancePtcls' setValues (:215-243)
            ne_d.append(capi.DevArray.from_host(ne_h))
            np_d.append(capi.DevArray.from_host(np_h))
            host.append((ne_h[live], np_h[live]))
        w_o, f_o = ob.weights([h[0] for h in host], [h[1] for h in host])
        plan_o, W_o = ob.plan(w_o, f_o, 1.05)
        for b, s, a, c in zip(bals, structs, ne_d, np_d):
            b.repartition_begin(s, a, c)
        for b in bals:
            b.repartition_end(1.05)
        for r, b in enumerate(bals):
            plan, W = b.last_plan()
            assert plan == plan_o[r] and W.tolist() == W_o
            after = np_d[r].to_host()
            live = structs[r].slot_info()[1].astype(bool)
            sent = np.bincount(after[live][after[live] != r], minlength=P)
            want = np.bincount(host[r][1][host[r][1] != r], minlength=P).astype(np.int64)
            for _, q, t in plan:
                want[q] += t
            assert np.array_equal(sent, want)
        for r, (s, a, c, cm) in enumerate(zip(structs, ne_d, np_d, comms)):
            capi.migrate_begin(s, a, c, cm, gid2lid=g2l[r])
        for s, cm in zip(structs, comms):
            capi.migrate_end(s, cm)
    counts = np.array([s.nPtcls() for s in structs])
    assert counts.sum() == total
    assert counts.max() * P / counts.sum() <= 1.5
    for r, (p, s) in enumerate(zip(parts, structs)):
        se, mk = s.slot_info()
        e = se[mk.astype(bool)]
        assert np.all(p.array(capi.PART_SAFE).astype(bool)[e] | (p.array(capi.PART_OWNERS, dim)[e] == r))
    for c in comms:
        c.destroy()


# ---------------------------------------------------------------- the time step on BFS parts (no full-mesh replica)
@pytest.mark.parametrize("dim", [2, 3])
def test_step_on_bfs_parts_matches_the_full_mesh_run(ppo, synth, capi, opp, dim):
    """Push + search + migrate on PICparts with a 1-layer buffer and a 1- / 0-layer safe zone -- every rank holds
    only its part's mesh; elements travel as global ids (globalIds(dim) of the part, a gid -> local table on
    the receiver).  The union of the ranks equals the single-structure run on the full mesh, particle by
    particle (element, position, phase bit-exact), and every particle sits in an element that is safe on its
    rank.  Vertex counts summed through the owners (reduceCommArray) equal the full-mesh histogram."""
    import common
    from test_gpu_comm import _oracle_run, _population, _snapshot, _step, H, K, D, NSTEPS
    pop = _population(synth, dim)
    P = 5  # (a buffer is made of whole parts: with 3 slabs the middle one would hold the full mesh)
    owner = slab_owners(pop["coords"], pop["e2v"], P, axis=0 if dim == 2 else 2)  # (the push turns in the R-z plane)
    ne_full = len(pop["e2v"])
    ref = _oracle_run(ppo, pop, NSTEPS)
    mg = capi.Mesh(dim, pop["coords"], pop["e2v"], pop["cls"])
    comms = capi.Comm.local(P)
    # (the coarse torus: one vertex layer around a slab is most of the mesh -- its safe zone is the core)
    parts = [capi.PicPart(mg, owner, comms[r], capi.PART_BFS, capi.PART_BFS, 0, 1, 1 if dim == 2 else 0)
             for r in range(P)]
    assert sum(p.nents[dim] < ne_full for p in parts) >= 4 and not any(p.is_full_mesh for p in parts)
    structs, tables, safes, owners_d, full_ids = [], [], [], [], []
    for r, p in enumerate(parts):
        ent = p.array(capi.PART_ENT_IDS, dim)
        mine = owner[pop["elem"]] == r
        elem = ent[pop["elem"][mine]].astype(np.int32)
        assert elem.size == 0 or elem.min() >= 0
        info = [np.ascontiguousarray(a[..., mine]) for a in pop["info"]]
        gids = p.array(capi.PART_GIDS, dim)
        structs.append(capi.PS.scs(capi.PARTICLE_XGCM, p.nents[dim], np.bincount(elem, minlength=p.nents[dim]).astype(np.int32),
                                   gids=gids, particle_elements=elem, particle_info=info))
        t = np.full(ne_full, -1, dtype=np.int32)
        t[gids] = np.arange(p.nents[dim], dtype=np.int32)
        tables.append(capi.DevArray.from_host(t))
        safes.append(capi.DevArray.from_host(p.array(capi.PART_SAFE)))
        owners_d.append(capi.DevArray.from_host(p.array(capi.PART_OWNERS, dim)))
        full_ids.append(p.array(capi.PART_FULL_IDS, dim))
    moved = 0
    for step in range(NSTEPS):
        keep = []
        for r, (p, ps) in enumerate(zip(parts, structs)):
            ids = capi.DevArray.from_host(np.full(max(ps.capacity(), 1), -1, dtype=np.int32))
            _step(capi, p.mesh, ps, dim, 6.0, ids)
            ne_d, np_d = capi.set_unsafe_procs(ps, ids, safes[r], owners_d[r], r)
            # 2-D: the caller's gid -> local table; tets: the structure's own (built from its element gids)
            capi.migrate_begin(ps, ne_d, np_d, comms[r], commit=True, gid2lid=tables[r] if dim == 2 else None)
            keep.append((ids, ne_d, np_d))
        for r, ps in enumerate(structs):
            ns, nr = capi.migrate_end(ps, comms[r])
            moved += ns
    assert moved > 0
    snaps = []
    for r, (p, ps) in enumerate(zip(parts, structs)):
        pid, se, x, ph = _snapshot(ps)
        assert np.all(p.array(capi.PART_SAFE).astype(bool)[se])
        snaps.append((pid, full_ids[r][se], x, ph))
    io, eo, xo, pho, _ = ref
    ids = np.concatenate([s[0] for s in snaps])
    order = np.argsort(ids)
    assert np.array_equal(ids[order], io)
    assert np.array_equal(np.concatenate([s[1] for s in snaps])[order], eo)
    assert np.array_equal(np.concatenate([s[2] for s in snaps], axis=1)[:, order], xo)
    assert np.array_equal(np.concatenate([s[3] for s in snaps])[order], pho)
    # per-vertex particle counts: every rank adds its own particles on its part, SUM through the owners
    cnt_full = np.zeros(mg.nverts)
    np.add.at(cnt_full, pop["e2v"][eo].ravel(), 1.0)
    arrs = []
    for r, (p, ps) in enumerate(zip(parts, structs)):
        se = _snapshot(ps)[1]
        w = np.zeros(p.nents[0])
        np.add.at(w, p.mesh.array(1).reshape(-1, dim + 1)[se].ravel(), 1.0)
        arrs.append(capi.DevArray.from_host(w))
    capi.picpart_reduce_all(parts, 0, capi.OP_SUM, arrs)
    for p, a in zip(parts, arrs):
        assert np.array_equal(a.to_host(), cnt_full[p.array(capi.PART_FULL_IDS, 0)])
    for c in comms:
        c.destroy()
