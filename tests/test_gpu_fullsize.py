"""BASELINE.json's configurations at FULL size on the GPU, checked through size-independent
properties (the oracle needs minutes at these sizes): every reported element really contains its
particle (independent numpy barycentric test on a sample), the two fused kernel variants agree on
all 10 M ids, rebuild conserves the population (count, id sum, per-element histogram), scatter
conserves mass, ps_combo160's pseudo-push/redistribute/rebuild loop keeps every particle."""
import os

import numpy as np
import pytest

import bench
import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi(pp):
    from pumipic_amd import capi as c
    c.init(0)
    return c


def _tet_bcc(coords, e2v, elems, pts):
    """barycentric coordinates (sum 1) of pts (n,3) in tets `elems`, plain numpy, own formula"""
    v = coords.reshape(-1, 3)[e2v.reshape(-1, 4)[elems]]  # n,4,3
    T = np.stack([v[:, 1] - v[:, 0], v[:, 2] - v[:, 0], v[:, 3] - v[:, 0]], axis=2)
    lam = np.linalg.solve(T, (pts - v[:, 0])[..., None])[..., 0]
    return np.concatenate([1 - lam.sum(1, keepdims=True), lam], axis=1)


def test_c2_full_size_properties(pp, capi, monkeypatch):
    """configs[1]: 100 800 tets, 10 M particles, push+search only."""
    w = bench.build_workload(pp, capi, "c2", 10_000_000, 0, 1, 0.5)
    s = pp.synth
    mesh, ps = w["mesh"], w["ps"]
    cap = ps.capacity()
    ids = capi.DevArray.from_host(np.full(cap, -1, dtype=np.int32))
    mask = ps.slot_info()[1].astype(bool)
    assert int(mask.sum()) == 10_000_000
    for step in range(3):
        found = capi.push_search(mesh, ps, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, ids, seeded=(step > 0),
                                 looplimit=200)
        assert found
        ps.swap_members(0, 1)
    got = ids.to_host()[:cap]
    x = ps.member(0)[:, :cap]  # after the swap member 0 is the searched position
    live = np.flatnonzero(mask)
    assert (got[live] >= -1).all() and (got[live] < w["ne"]).all()
    inside = live[got[live] >= 0]
    assert len(inside) > 0.99 * len(live)
    rng = np.random.default_rng(5)
    samp = rng.choice(inside, size=200_000, replace=False)
    lam = _tet_bcc(w["coords"], w["e2v"], got[samp], x[:, samp].T)
    assert lam.min() > -1e-9, lam.min()  # each sampled particle lies in the element it was given
    after = got
    # the two row-tiled kernels (walk in the loop / deferred walk) agree on every id
    res = {}
    state = [ps.member(m).copy() for m in range(5)]
    for q in ("0", "1"):
        monkeypatch.setenv("PP_WALK_QUEUE", q)
        for m in range(5):
            ps.set_member(m, state[m])
        idq = capi.DevArray.from_host(after.copy())
        capi.push_search(mesh, ps, s.XGC_H, s.XGC_K, s.XGC_D, 1.5, idq, seeded=True, looplimit=200)
        res[q] = (idq.to_host()[:cap][live], ps.member(1)[:, :cap][:, live])
    assert np.array_equal(res["0"][0], res["1"][0])
    assert np.array_equal(res["0"][1], res["1"][1])
    assert (res["0"][0] != after[live]).mean() > 0.05  # the comparison covered real walks


def test_c2_full_size_equals_oracle(pp, ppo, capi):
    """configs[1] at BASELINE's size against the oracle, particle by particle: 100 800 tets, 10 M particles, six steps of
    push + search with the ids re-used as seeds and x <-> x_tgt swapped in between (no rebuild: by the last step
    a third of the particles has left the element of its row and walks from its own seed record).  After every
    step the element id of every particle equals the oracle's; after the last, so do x, x_tgt and phi."""
    w = bench.build_workload(pp, capi, "c2", 10_000_000, 0, 1, 0.5)
    s = pp.synth
    mg, pg = w["mesh"], w["ps"]
    mo = ppo.Mesh(3, w["coords"], w["e2v"], w["cls"])
    po = ppo.PS.scs(ppo.PARTICLE_XGCM, w["ne"], w["ppe"], C_max=64, sigma=2**31 - 1, V=1024, pad_strat=0,
                    shuffle_padding=0.1, extra_padding=0.0, particle_elements=w["elem"], particle_info=w["info"])
    capg, capo = pg.capacity(), po.capacity()
    ids_g = capi.DevArray.from_host(np.full(capg, -1, dtype=np.int32))
    idg, mkg = pg.member(2)[0, :capg].copy(), pg.slot_info()[1]
    ido, mko = po.member(2)[0, :capo].copy(), po.slot_info()[1]
    ids_o = None
    # The libm link of the parity chain, at this size (round 5): a SECOND oracle structure takes the same six steps
    # with trig=0 -- glibc sin/cos, what the reference's Kokkos::Serial build calls -- and is compared with the HIP
    # path after every step: element ids of all 10 M particles, then x, x_tgt, phi within 1e-12 relative.
    pl = ppo.PS.scs(ppo.PARTICLE_XGCM, w["ne"], w["ppe"], C_max=64, sigma=2**31 - 1, V=1024, pad_strat=0,
                    shuffle_padding=0.1, extra_padding=0.0, particle_elements=w["elem"], particle_info=w["info"])
    ids_l = None
    libm_diff = []
    ppo.set_threads(ppo.max_threads())
    try:
        for step in range(6):
            capi.push_search(mg, pg, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, ids_g, seeded=(step > 0), looplimit=200)
            ppo.toroidal_push(po, mo, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, trig=1)
            ids_o = ppo.search_mesh(mo, po, elem_ids=ids_o, looplimit=200)["elem_ids"]
            ppo.toroidal_push(pl, mo, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, trig=0)
            ids_l = ppo.search_mesh(mo, pl, elem_ids=ids_l, looplimit=200)["elem_ids"]
            _, eg = common.by_id(idg, mkg, ids_g.to_host()[:capg])
            _, eo = common.by_id(ido, mko, ids_o[:capo])
            _, el = common.by_id(ido, mko, ids_l[:capo])
            assert np.array_equal(eg, eo), step
            libm_diff.append(int((eg != el).sum()))
            if step < 5:
                pg.swap_members(0, 1)
                for q in (po, pl):
                    a, b = q.member(0), q.member(1)
                    tmp = a.copy()
                    a[:] = b
                    b[:] = tmp
    finally:
        ppo.set_threads(1)
    assert (eo != common.by_id(ido, mko, po.slot_info()[0])[1]).mean() > 0.25  # (a third has left its row's element)
    for m in (0, 1, 4):
        _, a = common.by_id(ido, mko, po.member(m)[:, :capo])
        _, b = common.by_id(idg, mkg, pg.member(m)[:, :capg])
        assert np.array_equal(a, b), m
        _, c = common.by_id(ido, mko, pl.member(m)[:, :capo])
        c = c.astype(np.float64)  # (relative to the particle's own scale: the largest component of the member)
        rel = np.abs(c - b) / np.maximum(np.abs(c).max(axis=0, keepdims=True), 1e-300)
        assert rel.max() <= 1e-12, (m, rel.max())
    print("HIP path vs oracle(libm), 10 M particles: particles in another element after steps 1-6: %s" % libm_diff)
    assert max(libm_diff) == 0, libm_diff


def test_c3_full_size_properties(pp, capi):
    """configs[2] (2-D literal): push + search + updatePtclPositions + rebuild + gyroScatter."""
    w = bench.build_workload(pp, capi, "2dc3", 10_000_000, 0, 1, 0.5)
    s = pp.synth
    mesh, ps = w["mesh"], w["ps"]
    fwd, _ = capi.create_gyro_ring_mappings(mesh)
    n0 = ps.nPtcls()
    id_sum0 = int(np.arange(n0, dtype=np.int64).sum())
    for step in range(3):
        cap = ps.capacity()
        ids = capi.DevArray.from_host(np.full(cap, -1, dtype=np.int32))
        capi.push_search(mesh, ps, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, ids, seeded=True, looplimit=200)
        want = ids.to_host()[:cap]
        mask = ps.slot_info()[1].astype(bool)
        pid_before = ps.member(2)[0, :cap][mask]
        elem_before = want[mask]
        xt_before = ps.member(1)[:2, :cap][:, mask]
        ps.rebuild_commit(ids, 0, 1)
        se, mk = ps.slot_info()
        mk = mk.astype(bool)
        cap2 = ps.capacity()
        pid_after = ps.member(2)[0, :cap2][mk]
        kept = elem_before >= 0
        assert ps.nPtcls() == int(kept.sum()) == int(mk.sum())
        # every particle sits in the row of the element the search gave it
        order_b = np.argsort(pid_before[kept])
        order_a = np.argsort(pid_after)
        assert np.array_equal(pid_before[kept][order_b], pid_after[order_a])
        assert np.array_equal(elem_before[kept][order_b], se[:cap2][mk][order_a])
        # updatePtclPositions fused into the move: x <- x_tgt, x_tgt <- 0 (pseudoXGCm.cpp:92-108)
        assert np.array_equal(ps.member(0)[:2, :cap2][:, mk][:, order_a], xt_before[:, kept][:, order_b])
        assert not ps.member(1)[:, :cap2][:, mk].any()
        wsum = capi.gyro_scatter(mesh, ps, fwd).to_host()
        assert np.isfinite(wsum).all() and wsum.min() >= 0
    if ps.nPtcls() == n0:
        assert int(ps.member(2)[0, :ps.capacity()][ps.slot_info()[1].astype(bool)].astype(np.int64).sum()) == id_sum0


@pytest.mark.parametrize("kind", ["scs", "csr"])
def test_c4_ps_combo160_full_size(pp, capi, kind):
    """configs[3]: ps_combo160 'largeE_smallP' stress point, 1 M elements / 1 M particles, uniform
    distribution: pseudo-push then redistribute(p=0.5) + rebuild (performance_tests/ps_combo160.cpp:
    134-232), 3 rounds."""
    ne = npt = 1_000_000
    rng = np.random.default_rng(0)
    elems = rng.integers(0, ne, size=npt).astype(np.int32)
    ppe = np.bincount(elems, minlength=ne).astype(np.int32)
    order = np.argsort(elems, kind="stable").astype(np.int32)
    members = capi.PERF160
    info = [np.zeros((17, npt)), np.zeros((4, npt), dtype=np.int32), np.arange(npt, dtype=np.int64)[None, :]]
    if kind == "scs":
        ps = capi.PS.scs(members, ne, ppe, C_=64, sigma=ne, V=1024, particle_elements=elems,
                         particle_info=info)
    else:
        ps = capi.PS.csr(members, ne, ppe, particle_elements=elems, particle_info=info)
    parent = capi.DevArray.from_host(np.sqrt(np.arange(ne, dtype=np.float64)) * np.arange(ne))
    assert ps.nPtcls() == npt
    for rnd in range(3):
        capi.pseudo_push160(ps, parent)
        cap = ps.capacity()
        se, mk = ps.slot_info()
        mk = mk.astype(bool)
        d = ps.member(0)[:, :cap]
        nums = ps.member(1)[:, :cap]
        lint = ps.member(2)[0, :cap]
        slots = np.flatnonzero(mk)
        assert np.array_equal(lint[slots], slots)                   # lint(p) = p
        assert np.array_equal(nums[:, slots], 4 * slots[None, :] + np.arange(4)[:, None])
        e_of = se[:cap][slots].astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            expect = 10.3 * 10.3 * 10.3 / np.sqrt(slots.astype(np.float64)) / np.sqrt(e_of) + np.sqrt(e_of) * e_of
        ok = np.isfinite(expect)
        assert np.allclose(d[:, slots][:, ok], expect[ok][None, :], rtol=1e-14, atol=0)
        if kind == "scs":  # padded lanes run the functor with mask == 0; CSR has no padded lanes
            assert (d[:, ~mk] == 0).all() and (nums[:, ~mk] == -1).all()
        # redistribute: half of the particles draw a new element (uniform), the rest stay
        new_elem = se[:cap].copy()
        move = mk & (rng.random(cap) < 0.5)
        new_elem[move] = rng.integers(0, ne, size=int(move.sum()))
        new_elem[~mk] = -1
        # tag every particle with its slot so the move can be followed
        tag = np.zeros((1, ps.info().stride), dtype=np.int64)
        tag[0, slots] = slots
        ps.set_member(2, tag)
        ps.rebuild(new_elem)
        assert ps.nPtcls() == npt
        se2, mk2 = ps.slot_info()
        mk2 = mk2.astype(bool)
        cap2 = ps.capacity()
        tags = ps.member(2)[0, :cap2][mk2]
        assert np.array_equal(np.sort(tags), slots)                  # nobody lost or duplicated
        assert np.array_equal(se2[:cap2][mk2], new_elem[tags])       # everybody in the requested row


def test_config1_restated_size_matches_oracle(pp, ppo, capi):
    """BASELINE configs[0] at its restated size: Kuhn split of a 16^3 box = 24 576 tets,
    100 000 particles seeded on the y = 0 model face, push (-0.5, 0.8, 0) * 1/20 per step, legacy
    search_mesh (loop limit 100), updatePtclPositions + rebuild every step, 30 steps -- every step's
    element ids, wall faces, wall points and surviving population equal the oracle's."""
    import common
    synth = pp.synth
    pop = common.population_box(synth, n=16, num_ptcls=100_000)
    assert len(pop["e2v"]) == 24576
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=64)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_PUSH, C=64)
    ppo.set_threads(ppo.max_threads())
    try:
        steps = 0
        for it in range(30):
            if po.nPtcls() == 0:
                break
            ppo.linear_push(po, 1.0 / 20, -0.5, 0.8, 0.0)
            capi.linear_push(pg, 1.0 / 20, -0.5, 0.8, 0.0)
            ro = ppo.search_mesh_legacy3d(mo, po, looplimit=100)
            rg = capi.search_mesh_legacy3d(mg, pg, looplimit=100)
            assert ro["found"] == rg["found"]
            capo, capg = po.capacity(), pg.capacity()
            mko, mkg = po.slot_info()[1], pg.slot_info()[1]
            ido, idg = po.member(2)[0, :capo], pg.member(2)[0, :capg]
            for key, width in (("elem_ids", 1), ("xface", 1), ("xpoints", 3)):
                a = np.asarray(ro[key])
                b = rg[key].to_host()
                if width == 3:
                    a, b = a[:3 * capo].reshape(capo, 3).T, b[:3 * capg].reshape(capg, 3).T
                else:
                    a, b = a[:capo], b[:capg]
                io, vo = common.by_id(ido, mko, a)
                ig, vg = common.by_id(idg, mkg, b)
                assert np.array_equal(io, ig) and np.array_equal(vo, vg), (it, key)
            ppo.update_positions(po)
            po.rebuild(ro["elem_ids"])
            pg.rebuild_commit(rg["elem_ids"], 0, 1)
            assert po.nPtcls() == pg.nPtcls()
            steps += 1
        assert steps >= 10
    finally:
        ppo.set_threads(1)


def test_c2_intersection_mode_full_size(pp, ppo, capi):
    """configs[1] in the reference's intersection mode (search_mesh with requireIntersection,
    adjacency.tpp:284-361, through the packed-record walk k_search_mt3): 100 800 tets, 10 M particles.
    Size-independent properties with an independent numpy evaluation: every moved particle's ray ends on an
    EXPOSED face of the element it reports (the rays are followed to the boundary), the intersection point lies
    in that face's plane, inside the face, on the ray beyond the origin, and in the reported element; then a
    ALL 10 M particles (PP_TEST_MT_SAMPLE for fewer) are searched by the oracle from the same positions, on all host
    cores, and must agree bit for bit (parents, exit faces, intersection points)."""
    w = bench.build_workload(pp, capi, "c2mt", 10_000_000, 0, 1, 0.5)
    s = pp.synth
    mesh, ps = w["mesh"], w["ps"]
    cap = ps.capacity()
    capi.toroidal_push(ps, mesh, s.XGC_H, s.XGC_K, s.XGC_D, 0.5)
    r = capi.search_mesh(mesh, ps, require_intersection=True, looplimit=2000)
    assert r["found"] and r["not_in_elem"] == 0
    se, mk = ps.slot_info()
    live = np.flatnonzero(mk)
    assert len(live) == 10_000_000
    ids, faces = r["elem_ids"].to_host()[:cap], r["inter_faces"].to_host()[:cap]
    pts = r["inter_points"].to_host()[:3 * cap].reshape(cap, 3)
    steps = capi.search_walk_steps()
    if os.environ.get("PP_MT_PACKED") != "0":  # (the fallback walk on the Omega_h-style arrays does not count its steps)
        assert steps > 20 * len(live)  # tens of elements per ray
    assert (faces[~mk.astype(bool)] == -1).all() and not pts[~mk.astype(bool)].any()
    assert (ids[live] >= 0).all() and (faces[live] >= 0).all()  # every ray of this push reaches the wall
    exposed = mesh.array(capi.MESH_SIDE_EXPOSED)
    assert exposed[faces[live]].all()
    e2s = mesh.array(capi.MESH_ELEM2SIDES).reshape(-1, 4)
    rng = np.random.default_rng(11)
    samp = rng.choice(live, size=200_000, replace=False)
    assert (e2s[ids[samp]] == faces[samp][:, None]).any(axis=1).all()  # the face belongs to the reported element
    s2v = mesh.array(capi.MESH_SIDE2VERTS).reshape(-1, 3)
    coords = np.asarray(w["coords"]).reshape(-1, 3)
    tri = coords[s2v[faces[samp]]]  # n,3,3
    x0 = ps.member(0)[:, :cap][:, samp].T
    x1 = ps.member(1)[:, :cap][:, samp].T
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    scale = np.abs(coords).max()
    assert np.abs(np.einsum("ij,ij->i", pts[samp] - tri[:, 0], n)).max() < 1e-9 * scale        # in the plane
    d = x1 - x0
    t = np.einsum("ij,ij->i", pts[samp] - x0, d) / np.einsum("ij,ij->i", d, d)
    assert t.min() > 0                                                                           # ahead of the origin
    assert np.abs(pts[samp] - (x0 + t[:, None] * d)).max() < 1e-9 * scale                        # on the ray
    A = np.stack([tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]], axis=2)                         # n,3,2
    uv = np.stack([np.linalg.lstsq(A[i], pts[samp][i] - tri[i, 0], rcond=None)[0] for i in range(0, len(samp), 40)])
    assert uv.min() > -1e-6 and (uv.sum(1) < 1 + 1e-6).all()                                     # inside the face
    lam = _tet_bcc(w["coords"], w["e2v"], ids[samp], pts[samp])
    assert lam.min() > -1e-6                                                                     # in the element
    # ---- oracle on a sample of the same particles (same origins, same pushed targets)
    nsub = min(int(os.environ.get("PP_TEST_MT_SAMPLE", "10000000")), len(live))
    sub = np.sort(rng.choice(live, size=nsub, replace=False))
    elem = se[sub].astype(np.int32)
    order = np.argsort(elem, kind="stable")
    sub, elem = sub[order], elem[order]
    info = [ps.member(m)[:, :cap][:, sub] for m in range(5)]
    mo = ppo.Mesh(3, w["coords"], w["e2v"], w["cls"])
    po = ppo.PS.scs(ppo.PARTICLE_XGCM, w["ne"], np.bincount(elem, minlength=w["ne"]).astype(np.int32), C_max=64,
                    particle_elements=elem, particle_info=info)
    ppo.set_threads(ppo.max_threads())  # (the rays of different particles are independent)
    try:
        ro = ppo.search_mesh(mo, po, require_intersection=True, looplimit=2000)
    finally:
        ppo.set_threads(1)
    so, mko = po.slot_info()
    lo = np.flatnonzero(mko)
    pid_o = po.member(2)[0, :po.capacity()][lo]
    pid_g = ps.member(2)[0, :cap]
    slot_of = np.full(int(pid_g[live].max()) + 1, -1, dtype=np.int64)
    slot_of[pid_g[sub]] = sub
    gs = slot_of[pid_o]
    assert (gs >= 0).all()
    assert np.array_equal(ro["elem_ids"][lo], ids[gs])
    assert np.array_equal(ro["inter_faces"][lo], faces[gs])
    assert np.array_equal(ro["inter_points"].reshape(-1, 3)[lo], pts[gs])


def test_tet_c3_full_size_properties(pp, capi):
    """configs[2] on tets (the bench's default workload): 100 800 tets, 10 M particles, fused push + walk,
    rebuild with the commit and both gyroScatter calls in one entry point.  Size-independent checks:
    the population is conserved by particle id, every particle sits in the row of the element the walk
    gave it (verified with an independent numpy barycentric test on a sample), x <- x_tgt / x_tgt <- 0,
    and the scatter fields carry 2 rings x 4 vertices x mapped fraction of every particle."""
    _tet_c3_properties(pp, capi, 10_000_000, "100k", 3)


@pytest.mark.skipif(not os.environ.get("PP_TEST_BIG"), reason="opt-in (PP_TEST_BIG=<particles>): minutes of host work")
def test_tet_c3_byte_offsets_beyond_int32(pp, capi):
    """The same properties with a population whose member arrays are longer than 2^31 BYTES (one GPU, 288 GB of HBM:
    PP_TEST_BIG=300000000 on the 998 400-tet mesh -- 8-byte members end at byte 2.4e9 of their array, the 64-byte
    records at 1.9e10): every slot -> address product in the kernels has to be 64-bit."""
    n = int(os.environ["PP_TEST_BIG"])
    assert n * 8 > 2**31
    _tet_c3_properties(pp, capi, n, "1m", 2)


def _tet_c3_properties(pp, capi, nparticles, mesh_name, steps):
    w = bench.build_workload(pp, capi, "c3", nparticles, 0, 1, 0.5, "last", mesh_name)
    s = pp.synth
    mesh, ps = w["mesh"], w["ps"]
    fwd, bkwd = capi.create_gyro_ring_mappings(mesh)
    wf, wb = capi.DevArray(mesh.nverts, np.float64), capi.DevArray(mesh.nverts, np.float64)
    n0 = ps.nPtcls()
    rng = np.random.default_rng(9)
    for step in range(steps):
        cap = ps.capacity()
        ids = capi.DevArray(cap + cap // 10, np.int32)
        capi.push_search(mesh, ps, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, ids, seeded=False, looplimit=200)
        # the mode bench.py times: from the second step on the origins are trusted (pp_ps_set_origin_trust);
        # the counter proves that no particle finished without a containment test that would have failed
        assert capi.push_search_counters() == (0, 0, 0)
        ps.set_origin_trust(True)
        want = ids.to_host()[:cap]
        mask = ps.slot_info()[1].astype(bool)
        pid_before = ps.member(2)[0, :cap][mask]
        elem_before = want[mask]
        xt_before = ps.member(1)[:, :cap][:, mask]
        capi.rebuild_scatter(ps, mesh, ids, [fwd, bkwd], [wf, wb])
        se, mk = ps.slot_info()
        mk = mk.astype(bool)
        cap2 = ps.capacity()
        pid_after = ps.member(2)[0, :cap2][mk]
        kept = elem_before >= 0
        assert ps.nPtcls() == int(kept.sum()) == int(mk.sum())
        order_b, order_a = np.argsort(pid_before[kept]), np.argsort(pid_after)
        assert np.array_equal(pid_before[kept][order_b], pid_after[order_a])
        assert np.array_equal(elem_before[kept][order_b], se[:cap2][mk][order_a])
        x_after = ps.member(0)[:, :cap2][:, mk]
        assert np.array_equal(x_after[:, order_a], xt_before[:, kept][:, order_b])
        assert not ps.member(1)[:, :cap2][:, mk].any()          # x_tgt <- 0 (materialised on access)
        samp = rng.choice(int(mk.sum()), size=100_000, replace=False)
        lam = _tet_bcc(w["coords"], w["e2v"], se[:cap2][mk][samp], x_after[:, samp].T)
        assert lam.min() > -1e-9, lam.min()
        f, b = wf.to_host(), wb.to_host()
        assert np.array_equal(f, b) and np.isfinite(f).all() and f.min() >= 0
        # every particle adds 1 to two rings of its 4 vertices; each ring value is spread as value/8 over
        # 8 points x 4 mapped vertices (points outside the domain drop out)
        assert 0.8 * 32 * ps.nPtcls() <= f.sum() <= 32 * ps.nPtcls()
    assert ps.nPtcls() > 0.99 * n0


@pytest.mark.parametrize("name", ["c3", "2dc3"])
def test_c3_full_size_equals_oracle(pp, ppo, capi, name):
    """configs[2] at BASELINE's size AGAINST THE ORACLE, particle by particle (tets: the north-star variant; 2dc3:
    the 2-D literal of pseudoXGCm, 100 352 triangles, elliptical push + search_mesh_2d): 100 800 tets, 10 M particles (the
    literal population: one element holds 60 000), three steps of what bench.py times -- pp_push_search without
    seeds, pp_ps_rebuild_scatter with the commit and both ring maps -- in the steady state of the record-fed
    loop (nothing reads a member between the steps: row-major records, two columns per fetch, the over-full row's
    own blocks all run).  After every step the per-element particle counts, the layout arrays and both scatter
    fields equal the oracle's; after the last, every member of every particle equals the oracle's by particle id,
    bit for bit.  The oracle runs its per-particle loops on all host cores (results do not depend on the thread
    count); its structure has the same C, sigma, V and padding."""
    w = bench.build_workload(pp, capi, name, 10_000_000, 0, 1, 0.5)
    s = pp.synth
    mg, pg = w["mesh"], w["ps"]
    dim = w["dim"]
    mo = ppo.Mesh(dim, w["coords"], w["e2v"], w["cls"])
    po = ppo.PS.scs(ppo.PARTICLE_XGCM, w["ne"], w["ppe"], C_max=64, sigma=2**31 - 1, V=1024, pad_strat=0,
                    shuffle_padding=0.1, extra_padding=0.0, particle_elements=w["elem"], particle_info=w["info"])
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg)
    wf, wb = capi.DevArray(mg.nverts, np.float64), capi.DevArray(mg.nverts, np.float64)
    # The libm link of the parity chain, at this size (round 5): a SECOND oracle structure takes the same steps with
    # trig=0 (glibc sin/cos: the reference's Kokkos::Serial build).  After every step its per-particle elements (by
    # particle id) and its scatter field are compared with the shared-sincos oracle, which the HIP path equals bit
    # for bit (the assertions below); at the end positions and phi within 1e-12 relative (north_star's tolerance).
    pl = ppo.PS.scs(ppo.PARTICLE_XGCM, w["ne"], w["ppe"], C_max=64, sigma=2**31 - 1, V=1024, pad_strat=0,
                    shuffle_padding=0.1, extra_padding=0.0, particle_elements=w["elem"], particle_info=w["info"])
    libm_diff = []
    ppo.set_threads(ppo.max_threads())
    try:
        for step in range(3):
            if dim == 3:
                ppo.toroidal_push(po, mo, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, trig=1)
                ids_o = ppo.search_mesh(mo, po, looplimit=200)["elem_ids"]
                ppo.toroidal_push(pl, mo, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, trig=0)
                ids_l = ppo.search_mesh(mo, pl, looplimit=200)["elem_ids"]
            else:
                ppo.elliptical_push(po, mo, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, trig=1)
                _, ids_o, _ = ppo.search_mesh_2d(mo, po, looplimit=200)
                ppo.elliptical_push(pl, mo, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, trig=0)
                _, ids_l, _ = ppo.search_mesh_2d(mo, pl, looplimit=200)
            ppo.update_positions(pl)
            pl.rebuild(ids_l)
            cap = pg.capacity()
            ids_g = capi.DevArray(cap + cap // 10, np.int32)
            capi.push_search(mg, pg, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, ids_g, seeded=False, looplimit=200)
            if dim == 3:
                assert capi.push_search_counters() == (0, 0, 0)
            ppo.update_positions(po)
            po.rebuild(ids_o)
            capi.rebuild_scatter(pg, mg, ids_g, [fg, bg], [wf, wb])
            assert po.nPtcls() == pg.nPtcls()
            lo, lg = po.layout(), pg.layout()
            for k in ("offsets", "slice_to_chunk", "row_to_element", "mask"):
                assert np.array_equal(lo[k], lg[k]), (step, k)
            assert np.array_equal(ppo.gyro_scatter(mo, po, fo), wf.to_host()), step
            assert np.array_equal(ppo.gyro_scatter(mo, po, bo), wb.to_host()), step
            # libm oracle against the shared-sincos oracle (== the HIP path): same particles in the same elements
            sl, mkl = pl.slot_info()
            so_, mko_ = po.slot_info()
            il, el = common.by_id(pl.member(2)[0, :pl.capacity()], mkl, sl)
            io_, eo_ = common.by_id(po.member(2)[0, :po.capacity()], mko_, so_)
            assert np.array_equal(il, io_), step
            libm_diff.append(int((el != eo_).sum()))
            if libm_diff[-1] == 0:
                assert np.array_equal(ppo.gyro_scatter(mo, pl, fo), wf.to_host()), step
    finally:
        ppo.set_threads(1)
    so, mko = po.slot_info()
    sg, mkg = pg.slot_info()
    capo, capg = po.capacity(), pg.capacity()
    ido, idg = po.member(2)[0, :capo], pg.member(2)[0, :capg]
    io, eo = common.by_id(ido, mko, so)
    ig, eg = common.by_id(idg, mkg, sg)
    assert np.array_equal(io, ig) and np.array_equal(eo, eg)
    sl, mkl = pl.slot_info()
    idl = pl.member(2)[0, :pl.capacity()]
    for m in range(len(ppo.PARTICLE_XGCM)):
        _, a = common.by_id(ido, mko, po.member(m)[:, :capo])
        _, b = common.by_id(idg, mkg, pg.member(m)[:, :capg])
        assert np.array_equal(a, b), m
        _, c = common.by_id(idl, mkl, pl.member(m)[:, :pl.capacity()])
        c = c.astype(np.float64)  # (relative to the particle's own scale: the largest component of the member)
        rel = np.abs(c - b) / np.maximum(np.abs(c).max(axis=0, keepdims=True), 1e-300)
        assert rel.max() <= 1e-12, (m, rel.max())  # HIP path vs oracle(libm): north_star's 1e-12 relative
    print("%s: HIP path vs oracle(libm), 10 M particles: particles in another element after steps 1-3: %s"
          % (name, libm_diff))
    assert max(libm_diff) == 0, libm_diff


@pytest.mark.parametrize("world,per_rank", [(2, 16_000_000), (8, 2_000_000)])
def test_config5_share_two_virtual_ranks_full_size(pp, ppo, capi, world, per_rank):
    """BASELINE configs[4] at the size one GPU carries: the 998 400-tet mesh, 32 M particles -- here as TWO
    virtual ranks of 16 M on a `local` communicator, so the exchange really moves particles between two
    structures.  Three steps of what bench.py's c5 step calls (pp_push_search with trusted origins from the
    second step on, pp_migrate_ptcls_begin / pp_ps_migrate_end with commit + both gyroScatter maps,
    gyroSync).  Size-independent checks: the population is conserved BY PARTICLE ID across the ranks (no
    loss, no duplicate), every particle sits on the rank that owns its element and in the row of the element
    the walk gave it, its position is the x_tgt the push wrote (x <- x_tgt, x_tgt <- 0), a sample passes an
    independent numpy barycentric test, and the synced scatter fields carry 2 rings x 4 vertices x the mapped
    fraction of every particle on every rank."""
    from pumipic_amd import dist as ppdist
    # (world 2: the 32 M particles one GPU of configs[4] carries; world 8: the eight owner blocks of the node, 16 M)
    s = pp.synth
    ws = [bench.build_workload(pp, capi, "c5", per_rank, r, world, 0.5, mesh_size="1m") for r in range(world)]
    ne = ws[0]["ne"]
    assert ne == 998_400
    mesh = ws[0]["mesh"]  # (the ranks of one node hold the full mesh each; here they share one copy)
    owners = ppdist.element_block_owners(ne, world)
    owners_d = capi.DevArray.from_host(owners)
    safes = [capi.DevArray.from_host((owners == r).astype(np.uint8)) for r in range(world)]
    comms = capi.Comm.local(world)
    fwd, bkwd = capi.create_gyro_ring_mappings(mesh)
    for r, w in enumerate(ws):  # particle ids unique across the ranks
        tag = np.zeros((1, w["ps"].info().stride), dtype=np.int32)
        se, mk = w["ps"].slot_info()
        slots = np.flatnonzero(mk)
        assert len(slots) == per_rank and np.all(owners[se[slots]] == r)
        # (rank r's k-th particle becomes k + r * per_rank: the numbering of the oracle run at the end of the test)
        tag[0, slots] = w["ps"].member(2)[0, slots] + r * per_rank
        w["ps"].set_member(2, tag)
    rng = np.random.default_rng(11)
    total = world * per_rank
    moved = 0
    for step in range(3):
        before, fields = [], []
        for r, w in enumerate(ws):
            ps = w["ps"]
            cap = ps.capacity()
            ids = capi.DevArray(cap + cap // 10, np.int32)
            capi.push_search(mesh, ps, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, ids, seeded=False, looplimit=200)
            nf, nie, unm = capi.push_search_counters()
            assert nf == 0 and nie == 0 and unm == 0
            ps.set_origin_trust(True)  # from the second step on the origins are what the last walk accepted
            if step == 2:  # what the exchange must deliver, by particle id
                mk = ps.slot_info()[1].astype(bool)
                before.append((ps.member(2)[0, :cap][mk], ids.to_host()[:cap][mk], ps.member(1)[:, :cap][:, mk]))
            wf, wb = capi.DevArray(mesh.nverts, np.float64), capi.DevArray(mesh.nverts, np.float64)
            capi.migrate_ptcls_begin(ps, ids, safes[r], owners_d, comms[r], commit=True,
                                     scatter=(mesh, [fwd, bkwd], [wf, wb]))
            fields.append((wf, wb, ids))
        for r, w in enumerate(ws):
            ns, nr = capi.migrate_end(w["ps"], comms[r])
            moved += ns
        packed = [capi.gyro_sync_pack(mesh.nverts, wf, wb) for wf, wb, _ in fields]
        for r in range(world):
            comms[r].allreduce_sum(packed[r])
        alive = sum(w["ps"].nPtcls() for w in ws)
        assert 0.999 * total <= alive <= total
        for r, w in enumerate(ws):  # ownership: a rank holds only elements of its block
            se, mk = w["ps"].slot_info()
            assert np.all(owners[se[mk.astype(bool)]] == r)
        g0 = packed[0].to_host()
        for r in range(1, world):
            assert np.array_equal(g0, packed[r].to_host())  # every rank has the same synced fields
        f = g0[0::2]
        assert np.array_equal(f, g0[1::2]) and np.isfinite(f).all() and f.min() >= 0
        assert 0.8 * 32 * alive <= f.sum() <= 32 * alive
    assert moved > 1000  # particles really crossed the block boundary
    # ---- last step: identity of every particle
    pid_b = np.concatenate([b[0] for b in before])
    elem_b = np.concatenate([b[1] for b in before])
    xt_b = np.concatenate([b[2] for b in before], axis=1)
    kept = elem_b >= 0
    pid_a, elem_a, x_a = [], [], []
    for w in ws:
        ps = w["ps"]
        cap = ps.capacity()
        se, mk = ps.slot_info()
        mk = mk.astype(bool)
        pid_a.append(ps.member(2)[0, :cap][mk])
        elem_a.append(se[:cap][mk])
        x_a.append(ps.member(0)[:, :cap][:, mk])
        assert not ps.member(1)[:, :cap][:, mk].any()  # x_tgt <- 0
    pid_a, elem_a, x_a = np.concatenate(pid_a), np.concatenate(elem_a), np.concatenate(x_a, axis=1)
    ob, oa = np.argsort(pid_b[kept]), np.argsort(pid_a)
    assert np.array_equal(pid_b[kept][ob], pid_a[oa])            # nobody lost, nobody duplicated
    assert len(np.unique(pid_a)) == len(pid_a)
    assert np.array_equal(elem_b[kept][ob], elem_a[oa])           # everybody in the row of its new element
    assert np.array_equal(xt_b[:, kept][:, ob], x_a[:, oa])       # x <- x_tgt
    samp = rng.choice(len(pid_a), size=200_000, replace=False)
    lam = _tet_bcc(ws[0]["coords"], ws[0]["e2v"], elem_a[samp], x_a[:, samp].T)
    assert lam.min() > -1e-9, lam.min()
    # ---- and against the ORACLE: ONE structure holding both ranks' particles, the same three steps without any
    # exchange (per-particle loops on all host cores).  The union of the two ranks equals it particle by particle
    # -- element, position, b, phi, bit for bit -- and the synced scatter field is the single structure's.
    b_a = np.concatenate([w["ps"].member(3)[0, :w["ps"].capacity()][w["ps"].slot_info()[1].astype(bool)] for w in ws])
    phi_a = np.concatenate([w["ps"].member(4)[0, :w["ps"].capacity()][w["ps"].slot_info()[1].astype(bool)] for w in ws])
    mo = ppo.Mesh(3, ws[0]["coords"], ws[0]["e2v"], ws[0]["cls"])
    info = [np.concatenate([w["info"][m] for w in ws], axis=-1) for m in range(5)]
    info[2] = np.concatenate([np.arange(per_rank, dtype=np.int32) + r * per_rank for r in range(world)])
    po = ppo.PS.scs(ppo.PARTICLE_XGCM, ne, sum(w["ppe"] for w in ws), C_max=64, sigma=2**31 - 1, V=1024, pad_strat=0,
                    shuffle_padding=0.1, extra_padding=0.0, particle_elements=np.concatenate([w["elem"] for w in ws]),
                    particle_info=info)
    fo, _ = ppo.create_gyro_ring_mappings(mo, trig=1)
    ppo.set_threads(ppo.max_threads())
    try:
        for step in range(3):
            ppo.toroidal_push(po, mo, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, trig=1)
            ids_o = ppo.search_mesh(mo, po, looplimit=200)["elem_ids"]
            ppo.update_positions(po)
            po.rebuild(ids_o)
    finally:
        ppo.set_threads(1)
    so, mko = po.slot_info()
    capo = po.capacity()
    ido = po.member(2)[0, :capo]
    io, eo = common.by_id(ido, mko, so)
    assert np.array_equal(io, pid_a[oa]) and np.array_equal(eo, elem_a[oa])
    assert np.array_equal(common.by_id(ido, mko, po.member(0)[:, :capo])[1], x_a[:, oa])
    assert np.array_equal(common.by_id(ido, mko, po.member(3)[:, :capo])[1][0], b_a[oa])
    assert np.array_equal(common.by_id(ido, mko, po.member(4)[:, :capo])[1][0], phi_a[oa])
    assert np.array_equal(ppo.gyro_scatter(mo, po, fo), f)        # SUM over the ranks == the one structure's field
    for c in comms:
        c.destroy()


@pytest.mark.skipif(not os.environ.get("PP_TEST_C5_FULL"), reason="opt-in (PP_TEST_C5_FULL=1): 256 M particles, minutes of host work")
def test_config5_full_size_eight_virtual_ranks(pp, capi):
    """BASELINE configs[4] at its FULL size on one GPU: the 998 400-tet mesh, 8 x 32 M = 256 M particles as the node's
    eight element-block owners on a `local` communicator (~47 GB of device memory), three steps of what
    `bench.py --workload c5 --virtual-ranks 8` times (pp_push_search, pp_migrate_ptcls_begin / pp_ps_migrate_end with
    the commit and both gyroScatter maps, gyroSync).  Size-independent properties: the population is conserved BY
    PARTICLE ID across the ranks (nobody lost, nobody duplicated; particles that leave the domain are the only ones
    to go), every particle sits on the rank that owns its element and in the row of the element the walk gave it, its
    position is the x_tgt the push wrote (x <- x_tgt, x_tgt <- 0) -- checked on a sample of a million -- a sample
    passes an independent numpy barycentric test, and every rank holds the same synced scatter field, which carries
    2 rings x 4 vertices x the mapped fraction of every particle."""
    from pumipic_amd import dist as ppdist
    s = pp.synth
    world, per_rank = 8, 32_000_000
    ws = []
    for r in range(world):
        w = bench.build_workload(pp, capi, "c5", per_rank, r, world, 0.5, mesh_size="1m")
        for k in ("elem", "info", "ppe"):
            w.pop(k, None)
        ws.append(w)
    ne = ws[0]["ne"]
    assert ne == 998_400
    mesh = ws[0]["mesh"]
    owners = ppdist.element_block_owners(ne, world)
    owners_d = capi.DevArray.from_host(owners)
    safes = [capi.DevArray.from_host((owners == r).astype(np.uint8)) for r in range(world)]
    comms = capi.Comm.local(world)
    fwd, bkwd = capi.create_gyro_ring_mappings(mesh)
    for r, w in enumerate(ws):  # particle ids unique across the ranks: k-th particle of rank r -> k + r * per_rank
        ps = w["ps"]
        tag = np.zeros((1, ps.info().stride), dtype=np.int32)
        se, mk = ps.slot_info()
        slots = np.flatnonzero(mk)
        assert len(slots) == per_rank and np.all(owners[se[slots]] == r)
        tag[0, slots] = ps.member(2)[0, slots] + r * per_rank
        ps.set_member(2, tag)
        del tag, se, mk, slots
    total = world * per_rank
    rng = np.random.default_rng(13)
    sample = np.sort(rng.choice(total, size=1_000_000, replace=False))
    moved = 0
    want = None
    for step in range(3):
        fields = []
        if step == 2:
            want = {"elem": np.full(len(sample), -2, dtype=np.int64), "xt": np.zeros((3, len(sample)))}
        for r, w in enumerate(ws):
            ps = w["ps"]
            cap = ps.capacity()
            ids = capi.DevArray(cap + cap // 10, np.int32)
            capi.push_search(mesh, ps, s.XGC_H, s.XGC_K, s.XGC_D, 0.5, ids, seeded=False, looplimit=200)
            assert capi.push_search_counters() == (0, 0, 0)
            if step == 2:  # what the exchange must deliver for the sampled ids
                mk = ps.slot_info()[1].astype(bool)
                pid = ps.member(2)[0, :cap]
                pos = np.searchsorted(sample, pid[mk])
                pos[pos >= len(sample)] = 0
                hit = sample[pos] == pid[mk]
                slots = np.flatnonzero(mk)[hit]
                want["elem"][pos[hit]] = ids.to_host()[:cap][slots]
                want["xt"][:, pos[hit]] = ps.member(1)[:, :cap][:, slots]
                del mk, pid, pos, hit, slots
            wf, wb = capi.DevArray(mesh.nverts, np.float64), capi.DevArray(mesh.nverts, np.float64)
            capi.migrate_ptcls_begin(ps, ids, safes[r], owners_d, comms[r], commit=True,
                                     scatter=(mesh, [fwd, bkwd], [wf, wb]))
            fields.append((wf, wb, ids))
        for r, w in enumerate(ws):
            ns, nr = capi.migrate_end(w["ps"], comms[r])
            moved += ns
        packed = [capi.gyro_sync_pack(mesh.nverts, wf, wb) for wf, wb, _ in fields]
        for r in range(world):
            comms[r].allreduce_sum(packed[r])
        alive = sum(w["ps"].nPtcls() for w in ws)
        assert 0.999 * total <= alive <= total
        g0 = packed[0].to_host()
        for r in range(1, world):
            assert np.array_equal(g0, packed[r].to_host())
        f = g0[0::2]
        assert np.array_equal(f, g0[1::2]) and np.isfinite(f).all() and f.min() >= 0
        assert 0.8 * 32 * alive <= f.sum() <= 32 * alive
    assert moved > 100_000  # particles really crossed the block boundaries
    # (a sampled id that is not seen before the last exchange left the domain in an earlier step: ~2 in 10 000)
    there = want["elem"] != -2
    assert there.mean() > 0.999
    # ---- after the last step: identity of every particle, rank by rank
    seen = np.zeros(total, dtype=bool)
    got_elem = np.full(len(sample), -2, dtype=np.int64)
    got_x = np.zeros((3, len(sample)))
    n_alive = 0
    for r, w in enumerate(ws):
        ps = w["ps"]
        cap = ps.capacity()
        se, mk = ps.slot_info()
        mk = mk.astype(bool)
        pid = ps.member(2)[0, :cap][mk]
        assert not seen[pid].any() and len(np.unique(pid)) == len(pid)  # nobody duplicated
        seen[pid] = True
        n_alive += len(pid)
        assert np.all(owners[se[:cap][mk]] == r)  # ownership
        x = ps.member(0)[:, :cap][:, mk]
        assert not ps.member(1)[:, :cap][:, mk].any()  # x_tgt <- 0
        pos = np.searchsorted(sample, pid)
        pos[pos >= len(sample)] = 0
        hit = sample[pos] == pid
        got_elem[pos[hit]] = se[:cap][mk][hit]
        got_x[:, pos[hit]] = x[:, hit]
        samp = rng.choice(len(pid), size=50_000, replace=False)
        lam = _tet_bcc(ws[0]["coords"], ws[0]["e2v"], se[:cap][mk][samp], x[:, samp].T)
        assert lam.min() > -1e-9, lam.min()
        del se, mk, pid, x, pos, hit
    assert n_alive == sum(w["ps"].nPtcls() for w in ws)
    kept = want["elem"] >= 0  # (-1: left the domain in the last step -> deleted; -2: gone before)
    assert np.array_equal(got_elem[kept], want["elem"][kept])      # everybody in the row of its new element
    assert (got_elem[~kept] == -2).all()                            # the deleted ones are gone from every rank
    assert np.array_equal(got_x[:, kept], want["xt"][:, kept])     # x <- x_tgt, across the exchange
    lost = total - n_alive
    print("configs[4] full size: %d particles on 8 virtual ranks, %d migrated in 3 steps, %d left the domain" % (n_alive, moved, lost))
    for c in comms:
        c.destroy()
