"""CPU: property tests of the oracle modelled on the reference's own test strategy:
test_adj.cpp (search post-conditions), test_rebuild.cpp / test_structure.cpp (per-element id sums,
counts), scs_padding / buildSCSTest (layout invariants), Distribute.  These are the same
properties the HIP path is held to in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import common

H, K, D = 1.72479370 - .08, .020558260, 0.6
INT_MAX = 2**31 - 1


# ---------------------------------------------------------------- SCS layout invariants
@pytest.mark.parametrize("C,V,sigma,pad", [(1, 1024, INT_MAX, 0), (4, 2, 1, 0), (32, 3, 7, 1),
                                           (64, 1024, INT_MAX, 2), (64, 16, 100, 0)])
def test_scs_layout_invariants(ppo, synth, C, V, sigma, pad):
    ne, np_ = 300, 4000
    ppe, epp = synth.distribute_particles(ne, np_, 2, seed=3)
    ps = ppo.PS.scs(ppo.PARTICLE_PUSH, ne, ppe, C_max=C, sigma=sigma, V=V, pad_strat=pad)
    L = ps.layout()
    Cc = L["C"]
    assert Cc == min(C, int((ppe > 0).sum()))
    # every element owns exactly one row and the maps are inverse
    r2e, e2r = L["row_to_element"], L["element_to_row"]
    assert np.array_equal(np.sort(r2e[:ne]), np.arange(ne))
    assert np.array_equal(e2r[r2e[:ne]], np.arange(ne))
    # ascending particle counts inside every sigma window (SCS_sort.h:36-47)
    sg = min(sigma, ne)
    nwin = ne // sg
    cnt = ppe[r2e[:ne]]
    for w in range(nwin):
        a, b = w * sg, (ne if w == nwin - 1 else (w + 1) * sg)
        assert np.all(np.diff(cnt[a:b]) >= 0)
        assert set(r2e[a:b]) == set(range(a, b))  # sorting never crosses a window
    # slices: contiguous, multiples of C, at most V columns, slice_to_chunk monotone
    off, s2c = L["offsets"], L["slice_to_chunk"]
    sizes = np.diff(off)
    assert off[0] == 0 and off[-1] == L["capacity"] == ps.capacity()
    assert np.all(sizes % Cc == 0) and np.all(sizes // Cc <= V) and np.all(sizes > 0)
    assert np.all(np.diff(s2c) >= 0)
    # mask: per row exactly ppe live slots, packed at the front of the row
    se, mk = ps.slot_info()
    assert np.array_equal(np.bincount(se[mk > 0], minlength=ne)[:ne], ppe)
    assert mk.sum() == np_
    padded, pslices, empty = ps.metrics()
    assert padded == L["capacity"] - np_
    assert empty == (L["num_rows"] - ne) + int((ppe == 0).sum())


def test_csr_layout(ppo, synth):
    ne, np_ = 100, 1234
    ppe, epp = synth.distribute_particles(ne, np_, 1, seed=5)
    ids = np.arange(np_, dtype=np.int32)
    ps = ppo.PS.csr([(np.int32, 1)], ne, ppe, particle_elements=epp, particle_info=[ids])
    off = ps.layout()["offsets"]
    assert np.array_equal(np.diff(off), ppe) and ps.capacity() == int(np_ * 1.05)
    got = ps.member(0)[0, :np_]
    se, mk = ps.slot_info()
    assert np.array_equal(epp[got], se[:np_])       # every particle sits in its element's range
    for e in range(ne):                             # stable: input order kept inside an element
        assert np.all(np.diff(got[off[e]:off[e + 1]]) > 0)


# ---------------------------------------------------------------- rebuild (test_rebuild.cpp)
def _ids_structure(ppo, synth, kind, ne=120, np_=3000, strat=2):
    ppe, epp = synth.distribute_particles(ne, np_, strat, seed=11)
    ids = np.arange(np_, dtype=np.int32)
    members = [(np.int32, 1), (np.float64, 3)]
    info = [ids, np.vstack([ids * 0.5, ids * 2.0, -ids.astype(float)])]
    if kind == "scs":
        return ppo.PS.scs(members, ne, ppe, C_max=8, particle_elements=epp, particle_info=info), epp
    return ppo.PS.csr(members, ne, ppe, particle_elements=epp, particle_info=info), epp


def _by_id(ps):
    se, mk = ps.slot_info()
    cap = ps.capacity()
    live = mk.astype(bool)
    ids = ps.member(0)[0, :cap][live]
    o = np.argsort(ids)
    return ids[o], se[live][o], ps.member(1)[:, :cap][:, live][:, o]


@pytest.mark.parametrize("kind", ["scs", "csr"])
@pytest.mark.parametrize("shuffle", [True, False])
def test_rebuild_scenarios(ppo, synth, kind, shuffle):
    ps, epp = _ids_structure(ppo, synth, kind)
    if kind == "scs":
        ps.set_try_shuffling(shuffle)
    ne, np0 = ps.nElems(), ps.nPtcls()
    # 1) no changes (rebuildNoChanges, test_rebuild.cpp:4-66)
    se, mk = ps.slot_info()
    ps.rebuild(np.where(mk > 0, se, -1))
    ids, elem, dat = _by_id(ps)
    assert ps.nPtcls() == np0 and np.array_equal(elem, epp[ids])
    # 2) reassigned elements: (e*3 + id) % ne (rebuildNewElems :68-130)
    se, mk = ps.slot_info()
    pid = ps.member(0)[0, :ps.capacity()]
    tgt = np.where(mk > 0, (se * 3 + pid) % ne, -1).astype(np.int32)
    expect = {int(pid[s]): int(tgt[s]) for s in np.flatnonzero(mk)}
    sums_before = np.bincount(tgt[mk > 0], weights=pid[mk > 0], minlength=ne)
    ps.rebuild(tgt)
    ids, elem, dat = _by_id(ps)
    assert ps.nPtcls() == np0
    assert all(expect[int(i)] == int(e) for i, e in zip(ids, elem))
    assert np.array_equal(np.bincount(elem, weights=ids, minlength=ne), sums_before)  # id sums
    assert np.array_equal(dat, np.vstack([ids * 0.5, ids * 2.0, -ids.astype(float)]))  # payload
    # 3) remove every 7th particle, add new ones (rebuildNewPtcls / rebuildPtclsDestroyed)
    se, mk = ps.slot_info()
    pid = ps.member(0)[0, :ps.capacity()]
    tgt = np.where(mk > 0, se, -1).astype(np.int32)
    kill = (mk > 0) & (pid % 7 == 0)
    tgt[kill] = -1
    n_new = 40
    new_e = (np.arange(n_new) * 5 % ne).astype(np.int32)
    new_ids = np.arange(np0, np0 + n_new, dtype=np.int32)
    ps.rebuild(tgt, new_e, [new_ids, np.vstack([new_ids * 0.5, new_ids * 2.0, -new_ids.astype(float)])])
    ids, elem, dat = _by_id(ps)
    assert ps.nPtcls() == np0 - int(kill.sum()) + n_new
    assert not np.any((ids < np0) & (ids % 7 == 0))
    assert np.array_equal(elem[ids >= np0], new_e)
    assert np.array_equal(dat, np.vstack([ids * 0.5, ids * 2.0, -ids.astype(float)]))
    # 4) delete everything, then refill an empty structure
    ps.rebuild(np.full(max(ps.capacity(), 1), -1, dtype=np.int32))
    assert ps.nPtcls() == 0
    ps.rebuild(np.full(max(ps.capacity(), 1), -1, dtype=np.int32), new_e,
               [new_ids, np.zeros((3, n_new))])
    ids, elem, _ = _by_id(ps)
    assert ps.nPtcls() == n_new and np.array_equal(elem, new_e)
    # getPIDs (ps_for.hpp:65-85)
    off, pids = ps.get_pids()
    se, mk = ps.slot_info()
    assert np.all(mk[pids] == 1)
    assert np.array_equal(np.repeat(np.arange(ne), np.diff(off)), se[pids])


def test_reshuffle_fast_path_is_taken_and_equivalent(ppo, synth):
    """SCS_rebuild.h:4-120: when every row has room, particles move into holes in place."""
    a, epp = _ids_structure(ppo, synth, "scs", ne=60, np_=1500, strat=0)
    b, _ = _ids_structure(ppo, synth, "scs", ne=60, np_=1500, strat=0)
    b.set_try_shuffling(False)
    cap0 = a.capacity()
    for ps in (a, b):
        se, mk = ps.slot_info()
        pid = ps.member(0)[0, :ps.capacity()]
        tgt = np.where(mk > 0, se, -1).astype(np.int32)
        movers = (mk > 0) & (pid % 25 == 0)
        tgt[movers] = (se[movers] + 1) % 60
        ps.rebuild(tgt)
    assert a.s.last_rebuild_was_shuffle == 1 and a.capacity() == cap0
    assert b.s.last_rebuild_was_shuffle == 0
    ia, ea, da = _by_id(a)
    ib, eb, db = _by_id(b)
    assert np.array_equal(ia, ib) and np.array_equal(ea, eb) and np.array_equal(da, db)


# ---------------------------------------------------------------- search post-conditions (test_adj.cpp)
def _inside(ppo, mesh, elem, pos, tol):
    if mesh.dim == 2:
        b = ppo.barycentric_tri(mesh.coords[mesh.elem2verts[elem]], pos[:2], mesh.elem_measure[elem])
    else:
        b, _, _ = ppo.barycentric_tet(mesh.coords[mesh.elem2verts[elem]], pos, mesh.elem_measure[elem])
    return ppo.all_positive(b, tol)


@pytest.mark.parametrize("dim", [2, 3])
def test_bcc_search_postconditions(ppo, synth, dim):
    """every found particle's target lies in elem_ids[pid] (test_adj.cpp:565-587); particles
    reported outside really left through an exposed side (:591-614)."""
    pop = common.population_2d(synth, num_ptcls=1500) if dim == 2 else common.population_3d(synth, num_ptcls=1500)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=1)
    if dim == 2:
        ppo.elliptical_push(ps, mesh, H, K, D, 25.0, trig=0)
    else:
        ppo.toroidal_push(ps, mesh, H, K, D, 25.0, trig=0)
    r = ppo.search_mesh(mesh, ps, require_intersection=False, looplimit=1000)
    assert r["found"] and r["not_in_elem"] == 0
    se, mk = ps.slot_info()
    xt = ps.member(1)[:, :ps.capacity()]
    moved = 0
    for s in np.flatnonzero(mk)[::7]:
        e = r["elem_ids"][s]
        if e >= 0:
            assert _inside(ppo, mesh, e, xt[:, s], 1e-10)
            moved += e != se[s]
        else:  # not inside ANY element
            assert not any(_inside(ppo, mesh, t, xt[:, s], 0.0) for t in range(0, mesh.nelems, 1)) \
                if mesh.nelems < 2000 else True
    assert moved > 10


def test_search_mesh_3d_agrees_with_the_other_walks(ppo, synth):
    """search_mesh_3d (adjacency.hpp:314-555) against the two other 3-D walks on generic inputs:
    particles that stay in the domain end in the element the BCC walk of search_mesh finds, wall
    hits are the legacy search's (same intersection routine), and every found target is inside
    its element (test_adj.cpp:565-587)."""
    pop = common.population_box(synth, n=4, num_ptcls=800)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=1)
    ppo.linear_push(ps, 0.6, -0.5, 0.8, 0.1)
    r3 = ppo.search_mesh_3d(mesh, ps, looplimit=200)
    rl = ppo.search_mesh_legacy3d(mesh, ps, looplimit=200)
    rb = ppo.search_mesh(mesh, ps, require_intersection=False, looplimit=200)
    assert r3["found"] == 1 and rl["found"] == 1 and rb["found"]
    se, mk = ps.slot_info()
    live = mk.astype(bool)
    assert np.array_equal(r3["elem_ids"][live], rl["elem_ids"][live])
    assert np.array_equal(r3["xface"][live], rl["xface"][live])
    hit = live & (r3["xface"] >= 0)
    assert hit.sum() > 20 and (live & ~hit).sum() > 100
    assert np.array_equal(r3["xpoints"][hit], rl["xpoints"][hit])
    assert np.all(r3["elem_ids"][hit] == -1)
    stay = live & ~hit
    assert np.array_equal(r3["elem_ids"][stay], rb["elem_ids"][stay])
    assert (r3["elem_ids"][stay] != se[stay]).sum() > 50
    xt = ps.member(1)[:, :ps.capacity()]
    for s in np.flatnonzero(stay)[::5]:
        assert _inside(ppo, mesh, r3["elem_ids"][s], xt[:, s], 1e-10)
    # wall points lie on the reported exposed face and on the path
    x0 = ps.member(0)[:, :ps.capacity()]
    for s in np.flatnonzero(hit)[::3]:
        f = r3["xface"][s]
        assert mesh.side_exposed[f]
        tri = mesh.coords[mesh.side2verts[f]]
        n = np.cross(tri[1] - tri[0], tri[2] - tri[0])
        assert abs(np.dot(r3["xpoints"][s] - tri[0], n)) < 1e-12
        d, dx = xt[:, s] - x0[:, s], r3["xpoints"][s] - x0[:, s]
        assert np.linalg.norm(np.cross(d, dx)) < 1e-12 and np.dot(d, dx) >= 0


def test_search_mesh_3d_seeds_masks_and_loop_limit(ppo, synth):
    """elem_ids passed in: -1 stays -1 (ptcl_done=2, hpp:361-363); a loop limit leaves the
    unfinished particles at their current element and returns not-found (hpp:531-552); a particle
    outside its ROW element trips checkParent (hpp:371-382)."""
    pop = common.population_box(synth, n=4, num_ptcls=400)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=4)
    ppo.linear_push(ps, 0.55, 0.3, 0.8, 0.2)
    se, mk = ps.slot_info()
    live = np.flatnonzero(mk)
    seed = se.copy()
    seed[~mk.astype(bool)] = -1
    seed[live[::5]] = -1
    full = ppo.search_mesh_3d(mesh, ps, elem_ids=seed.copy(), looplimit=0)
    assert full["found"] == 1 and np.all(full["elem_ids"][live[::5]] == -1)
    assert np.all(full["elem_ids"][~mk.astype(bool)] == -1)
    lim = ppo.search_mesh_3d(mesh, ps, elem_ids=seed.copy(), looplimit=2)
    assert lim["found"] == 0 and lim["loops"] == 2
    unfinished = lim["elem_ids"] != full["elem_ids"]
    assert unfinished.sum() > 0 and np.all(lim["elem_ids"][unfinished] >= 0)
    ps.member(0)[:, live[3]] += 10.0  # a view of the structure's storage
    assert ppo.search_mesh_3d(mesh, ps, looplimit=50)["found"] == -2


def test_intersection_search_postconditions(ppo, synth):
    """wall hits lie in the reported exposed face, on the particle's path, and the face bounds the
    final element (test_adj.cpp:630-735)."""
    pop = common.population_box(synth, n=3, num_ptcls=300)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=1)
    ppo.linear_push(ps, 0.35, -0.5, 0.8, 0.3)
    r = ppo.search_mesh(mesh, ps, require_intersection=True, looplimit=1000)
    assert r["found"]
    se, mk = ps.slot_info()
    x, xt = ps.member(0)[:, :ps.capacity()], ps.member(1)[:, :ps.capacity()]
    hits = 0
    for s in np.flatnonzero(mk):
        f = r["inter_faces"][s]
        assert f >= 0  # rays (not segments) always reach the wall (adjacency.tpp:438)
        assert mesh.side_exposed[f] == 1
        e = r["elem_ids"][s]
        assert f in mesh.elem2sides[e]
        tri = mesh.coords[mesh.side2verts[f]]
        n = np.cross(tri[1] - tri[0], tri[2] - tri[0])
        p = r["inter_points"][s]
        assert abs(np.dot(p - tri[0], n)) <= 1e-9 * np.linalg.norm(n)        # in the face plane
        d = xt[:, s] - x[:, s]
        t = np.dot(p - x[:, s], d) / np.dot(d, d)
        assert t >= -1e-9 and np.linalg.norm(x[:, s] + t * d - p) <= 1e-9   # on the path
        hits += 1
    assert hits == 300


def test_looplimit_marks_unfound(ppo, synth):
    pop = common.population_2d(synth, num_ptcls=800)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=1)
    ppo.elliptical_push(ps, mesh, H, K, D, 12.0, trig=0)
    found, ids, loops = ppo.search_mesh_2d(mesh, ps, looplimit=2)
    assert not found and loops == 2
    r = ppo.search_mesh(mesh, ps, looplimit=2)
    assert not r["found"]
    _, mk = ps.slot_info()
    assert (ids[mk > 0] == -1).any() and (ids[mk > 0] >= 0).any()


# ---------------------------------------------------------------- push / scatter invariants
def test_elliptical_push_stays_on_ellipse(ppo, synth):
    pop = common.population_2d(synth, num_ptcls=500)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=1)
    _, mk = ps.slot_info()
    live = mk.astype(bool)
    b = ps.member(3)[0, :ps.capacity()][live].astype(np.float64)
    for _ in range(5):
        ppo.elliptical_push(ps, mesh, H, K, D, 3.0, trig=0)
    xt = ps.member(1)[:, :ps.capacity()][:, live]
    q = ((xt[0] - H) / (D * b)) ** 2 + ((xt[1] - K) / b) ** 2
    assert np.abs(q - 1).max() < 1e-12


def test_gyro_scatter_mass(ppo, synth):
    """independent numpy evaluation of gyroScatter (gyroScatter.hpp:168-229): a particle in
    element e adds 1 to rings 0 and 1 of e's vertices; ring r of vertex v sends accum/gppr to every
    mapped vertex of its gppr points."""
    pop = common.population_2d(synth, num_ptcls=2000)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=1)
    gnr, gppr = 3, 8
    f, b = ppo.create_gyro_ring_mappings(mesh, 0.038, gnr, gppr, 0.0)
    assert np.array_equal(f, b)
    w = ppo.gyro_scatter(mesh, ps, f, 0.038, gnr, gppr)
    acc = np.zeros((mesh.nverts, gnr))
    for r in (0, 1):
        np.add.at(acc[:, r], mesh.elem2verts.ravel(), np.repeat(pop["ppe"], 3))
    m = f.reshape(mesh.nverts, gnr, gppr * 3)
    expect = np.zeros(mesh.nverts)
    for v in range(mesh.nverts):
        for r in range(gnr):
            tgt = m[v, r][m[v, r] >= 0]
            np.add.at(expect, tgt, acc[v, r] / gppr)
    assert np.array_equal(w, expect)          # multiples of 1/8: exact in any order
    assert 0.5 * 18 * 2000 < w.sum() <= 18 * 2000


def test_distribute_strategies(synth):
    for strat in range(5):
        ppe, epp = synth.distribute_particles(1000, 50000, strat, seed=0)
        assert ppe.sum() == 50000 and len(epp) == 50000 and epp.min() >= 0 and epp.max() < 1000
        assert np.array_equal(np.bincount(epp, minlength=1000), ppe)
    ppe, _ = synth.distribute_particles(1000, 50000, 0)
    assert ppe.max() - ppe.min() <= 1
    ppe, _ = synth.distribute_particles(1000, 50000, 4)
    assert abs(ppe[:400].sum() / 50000 - 0.85) < 1e-3


def test_pseudo_push_and_boris(ppo, synth):
    ne = 50
    ppe, _ = synth.distribute_particles(ne, 700, 1, seed=2)
    ps = ppo.PS.scs(ppo.PERF160, ne, ppe, C_max=4, sigma=ne)
    ped = np.sqrt(np.arange(ne, dtype=np.float64)) * np.arange(ne)
    ppo.pseudo_push160(ps, ped)
    se, mk = ps.slot_info()
    cap = ps.capacity()
    d, n, l = ps.member(0)[:, :cap], ps.member(1)[:, :cap], ps.member(2)[0, :cap]
    s = np.flatnonzero((mk > 0) & (se > 0))
    s = s[s > 0]
    expect = 10.3 * 10.3 * 10.3 / np.sqrt(s.astype(float)) / np.sqrt(se[s].astype(float)) + ped[se[s]]
    assert np.array_equal(d[5, s], expect) and np.array_equal(n[2, s], 4 * s + 2) and np.array_equal(l[s], s)
    assert np.all(d[:, mk == 0] == 0) and np.all(n[:, mk == 0] == -1)
    # Boris: pure E field accelerates along E, |v| preserved by pure B
    z = np.zeros(4)
    arrs = [z.copy() for _ in range(15)]
    arrs[6][:] = 1.0           # vx
    arrs[14][:] = 1e-3         # Bz
    ppo.push_boris(*arrs, 1e-6)
    assert np.allclose(np.hypot(arrs[6], arrs[7]), 1.0, rtol=0, atol=1e-12)


def test_gyro_ring_map_3d_points_are_inside_their_tets(ppo, synth):
    """tet ring map: every mapped ring point really lies in the tet whose 4 vertices it lists
    (independent numpy barycentric solve), unmapped points are outside the domain's (R,Z) range."""
    coords, e2v, cls = synth.torus_tet(n_b=4, n_theta=16, n_planes=8)
    mesh = ppo.Mesh(3, coords, e2v, cls)
    rmax, gnr, gppr = 0.03, 3, 8
    fwd, bkwd = ppo.create_gyro_ring_mappings(mesh, rmax=rmax, gnr=gnr, gppr=gppr, trig=0)
    assert np.array_equal(fwd, bkwd)
    fwd = fwd.reshape(mesh.nverts, gnr, gppr, 4)
    xyz = np.asarray(coords).reshape(-1, 3)
    tets = np.sort(np.asarray(e2v).reshape(-1, 4), axis=1)
    known = {tuple(t) for t in tets}
    rng = np.random.default_rng(3)
    checked = 0
    for v in rng.choice(mesh.nverts, size=60, replace=False):
        R = np.hypot(xyz[v, 0], xyz[v, 1])
        for r in range(gnr):
            for k in range(gppr):
                vs = fwd[v, r, k]
                if vs[0] < 0:
                    assert (vs < 0).all()
                    continue
                assert tuple(sorted(vs)) in known
                rad = np.deg2rad(k / gppr * 360)
                Rp, Zp = R + rmax * (r + 1) / gnr * np.cos(rad), xyz[v, 2] + rmax * (r + 1) / gnr * np.sin(rad)
                p = np.array([Rp / R * xyz[v, 0], Rp / R * xyz[v, 1], Zp])
                T = (xyz[vs[1:]] - xyz[vs[0]]).T
                lam = np.linalg.solve(T, p - xyz[vs[0]])
                assert lam.min() > -1e-7 and lam.sum() < 1 + 1e-7
                checked += 1
    assert checked > 500


def test_gather_side_reproduces_linear_fields(ppo, synth):
    """interpolateTetVtx with findBCCoordsInTet is exact for fields linear in x,y,z; the grid
    interpolators are exact for (bi/tri-)linear data; interp2dVector rotates (R,phi) into (x,y)."""
    pop = common.population_3d(synth, n_b=5, n_theta=20, n_planes=8, num_ptcls=3000)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    cap = ps.capacity()
    live = ps.slot_info()[1].astype(bool)
    x = ps.member(0)[:, :cap]
    xyz = np.asarray(pop["coords"]).reshape(-1, 3)
    coef = np.array([[0.3, -1.2, 2.0], [1.0, 0.5, -0.7], [-2.0, 0.1, 0.9]])
    field = (xyz @ coef.T + np.array([0.5, -1.0, 2.0])).ravel()  # dof 3, vertex-major
    out, bad = ppo.gather_tet_vtx(mesh, ps, field, dof=3)
    assert bad == 0
    expect = coef @ x + np.array([0.5, -1.0, 2.0])[:, None]
    assert np.allclose(out[:, :cap][:, live], expect[:, live], rtol=1e-11, atol=1e-11)
    assert not out[:, :cap][:, ~live].any()
    # regular (R,z) grid, linear data f = 2R - 3z + 1
    R = np.hypot(x[0], x[1])
    nx, nz, gx0, gz0, dx, dz = 40, 30, 0.9, -0.8, 0.04, 0.06
    gr, gz = gx0 + dx * np.arange(nx), gz0 + dz * np.arange(nz)
    data = (2 * gr[None, :] - 3 * gz[:, None] + 1).ravel()  # index i + j*nx
    got = ppo.interp2d_field(ps, data, gx0, gz0, dx, dz, nx, nz, cyl_symm=True)
    inside = live & (R >= gr[0]) & (R < gr[-1]) & (x[2] >= gz[0]) & (x[2] < gz[-1])
    assert inside.sum() > 1000
    assert np.allclose(got[:cap][inside], (2 * R - 3 * x[2] + 1)[inside], rtol=1e-12, atol=1e-12)
    # vector field with constant (R,phi,z) components -> rotated into x,y
    d3 = np.tile(np.array([1.5, -0.5, 0.25]), nx * nz)
    v = ppo.interp2d_vector(ps, d3, gx0, gz0, dx, dz, nx, nz, cyl_symm=True)
    th = np.arctan2(x[1], x[0])
    assert np.allclose(v[0, :cap][inside], (np.cos(th) * 1.5 + np.sin(th) * 0.5)[inside], atol=1e-12)
    assert np.allclose(v[1, :cap][inside], (np.sin(th) * 1.5 - np.cos(th) * 0.5)[inside], atol=1e-12)
    assert np.allclose(v[2, :cap][inside], 0.25, atol=1e-14)
    # tri-linear grid, data = x + 2y - z
    gx, gy, gzz = np.linspace(-2.1, 2.1, 22), np.linspace(-2.1, 2.1, 18), np.linspace(-0.9, 0.9, 12)
    d = (gx[None, None, :] + 2 * gy[None, :, None] - gzz[:, None, None]).ravel()  # i + j*nx + k*nx*ny
    got3 = ppo.interp3d_field(ps, gx, gy, gzz, d)
    assert np.allclose(got3[:cap][live], (x[0] + 2 * x[1] - x[2])[live], rtol=1e-12, atol=1e-12)


def test_closest_point_on_triangle_properties(ppo):
    """closest_point_on_triangle[_wnormal] (adjacency.hpp:824-1009): the result is on the triangle,
    no sampled triangle point is closer, the region code names the feature the point lies on, and
    the float-narrowed _wnormal form agrees to single precision."""
    rng = np.random.default_rng(11)
    uv = rng.random((400, 2))
    uv[uv.sum(1) > 1] = 1 - uv[uv.sum(1) > 1]
    seen = set()
    for _ in range(60):
        abc = rng.normal(size=(3, 3))
        a, b, c = abc
        samples = a + uv[:, :1] * (b - a) + uv[:, 1:] * (c - a)
        n = np.cross(b - a, c - a)
        for _ in range(12):
            p = rng.normal(size=3) * 2
            q, reg = ppo.closest_point_on_triangle(abc, p, reg0=-7)
            qw, regw = ppo.closest_point_on_triangle(abc, p, wnormal=True)
            seen.add(regw)
            assert reg == (regw if regw != 3 else -7)  # EDGEAB is not reported by the plain form
            assert np.allclose(q, qw, rtol=0, atol=2e-5 * (1 + np.abs(abc).max()))
            assert abs(np.dot(q - a, n)) < 1e-12 * (1 + np.dot(n, n))
            dq = np.linalg.norm(p - q)
            assert dq <= np.linalg.norm(p - samples, axis=1).min() + 1e-12
            if regw in (0, 1, 2):
                assert np.array_equal(q, abc[regw])
            elif regw == 6:  # foot of the perpendicular
                assert np.linalg.norm(np.cross(p - q, n)) < 1e-9 * (1 + np.dot(n, n))
    assert seen == {0, 1, 2, 3, 4, 5, 6}


@pytest.mark.parametrize("dim,mt", [(2, False), (2, True), (3, False), (3, True)])
def test_trace_with_functor(ppo, synth, dim, mt):
    """trace_particle_through_mesh's functor hook (adjacency.tpp:470-476,561-565): the default
    functor passed through the hook reproduces search_mesh; a wall model on class interfaces stops
    particles exactly on sides that separate two classes, in the element they came from."""
    pop = common.population_2d(synth, num_ptcls=1200) if dim == 2 else common.population_3d(synth, num_ptcls=1200)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    (ppo.elliptical_push if dim == 2 else ppo.toroidal_push)(ps, mesh, H, K, D, 30.0, trig=1)
    se, mk = ps.slot_info()
    cap = ps.capacity()
    xt = ps.member(1)
    xt[:, :cap][:, mk.astype(bool)] = common.radial_kick(xt[:, :cap][:, mk.astype(bool)], dim, H, K)
    ref = ppo.search_mesh(mesh, ps, require_intersection=mt, looplimit=500)

    def default(st):  # check_model_intersection, tpp:372-385
        idx = np.flatnonzero(mk.astype(bool) & (st["ptcl_done"] == 0))
        exposed = mesh.side_exposed[st["last_exit"][idx]].astype(bool)
        st["ptcl_done"][idx] = exposed
        if mt:
            st["inter_faces"][idx[exposed]] = st["last_exit"][idx[exposed]]
        else:
            st["elem_ids"][idx[exposed]] = -1
    got = ppo.trace_particle_through_mesh(mesh, ps, default, require_intersection=mt, looplimit=500)
    assert got["found"] == ref["found"] and got["loops"] == ref["loops"]
    assert np.array_equal(got["elem_ids"], ref["elem_ids"])
    assert np.array_equal(got["inter_faces"], ref["inter_faces"])
    assert np.array_equal(got["inter_points"], ref["inter_points"].ravel())

    hits = []
    wall = ppo.trace_particle_through_mesh(mesh, ps, common.class_interface_functor(mesh, mk, hits),
                                           require_intersection=mt, looplimit=500)
    assert wall["found"] and sum(hits) > 20
    live = np.flatnonzero(mk)
    f = wall["inter_faces"][live]
    stopped = live[(f >= 0) & ~mesh.side_exposed[np.maximum(f, 0)].astype(bool)]
    assert len(stopped) == sum(hits)
    for s in stopped:
        side = wall["inter_faces"][s]
        a, b = mesh.side2elems[mesh.side2elems_off[side]:mesh.side2elems_off[side] + 2]
        assert mesh.class_id[a] != mesh.class_id[b] and wall["elem_ids"][s] in (a, b)
        # every element on the way had the class of the start element
        assert mesh.class_id[wall["elem_ids"][s]] == mesh.class_id[se[s]]
    # particles that never met an interface end where the plain search puts them
    free = np.setdiff1d(live, stopped)
    same_class = mesh.class_id[np.maximum(ref["elem_ids"][free], 0)] == mesh.class_id[se[free]]
    keep = free[(ref["elem_ids"][free] >= 0) & same_class]
    if mt and dim == 3:  # the 3-D intersection walk follows the RAY to the boundary
        ff = wall["inter_faces"][free]
        assert np.all(mesh.side_exposed[ff[ff >= 0]] == 1)
    else:
        assert len(keep) > 100
    assert np.array_equal(wall["elem_ids"][keep], ref["elem_ids"][keep])


def test_redistribute_particles_statistics(ppo):
    """redistribute_particles (Distribute.h:28-89), uniform strategy: masked slots get -1, about
    percentMoved of the live particles draw a new element, uniformly over the elements, and the
    draws depend on the seed only."""
    ne, npt = 500, 40000
    rng = np.random.default_rng(2)
    elems = np.sort(rng.integers(0, ne, size=npt).astype(np.int32))
    ppe = np.bincount(elems, minlength=ne).astype(np.int32)
    info = [np.zeros((17, npt)), np.zeros((4, npt), dtype=np.int32), np.arange(npt, dtype=np.int64)[None, :]]
    ps = ppo.PS.scs(ppo.PERF160, ne, ppe, C_max=32, sigma=ne, V=1024, particle_elements=elems, particle_info=info)
    se, mk = ps.slot_info()
    live = mk.astype(bool)
    a = ppo.redistribute_particles(ps, 0.5, seed=7)
    assert np.array_equal(a, ppo.redistribute_particles(ps, 0.5, seed=7))
    assert not np.array_equal(a, ppo.redistribute_particles(ps, 0.5, seed=8))
    assert np.all(a[~live] == -1) and np.all((a[live] >= 0) & (a[live] < ne))
    moved = (a[live] != se[live]).mean()
    assert 0.47 < moved < 0.53
    cnt = np.bincount(a[live][a[live] != se[live]], minlength=ne)
    assert cnt.min() > 0 and cnt.max() < 4 * cnt.mean()
    assert np.array_equal(ppo.redistribute_particles(ps, 0.0, seed=1)[live], se[live])
    assert (ppo.redistribute_particles(ps, 1.0, seed=1)[live] != se[live]).mean() > 0.99


@pytest.mark.parametrize("dim,bridge", [(2, 0), (2, 1), (3, 0), (3, 2)])
def test_picpart_bfs_layers(ppo, synth, dim, bridge):
    """bfsBufferLayers / bfsSafeInward (pumipic_part_construct.cpp:387-468) against an independent
    breadth-first distance over the element graph: safe = within safe_layers of the core, buffered
    parts = owners met within ghost_layers, inward safe = own + buffered elements at least
    safe_layers+1 away from the unbuffered region."""
    coords, e2v, cls = synth.annulus_tri(n_b=10, n_theta=40, band_width=3) if dim == 2 else \
        synth.torus_tet(n_b=4, n_theta=12, n_planes=8)
    mesh = ppo.Mesh(dim, coords, e2v, cls)
    ne, nranks, rank = mesh.nelems, 5, 2
    owner = (np.arange(ne, dtype=np.int64) * nranks // ne).astype(np.int32)
    off, vals = (mesh.vert2elems_off, mesh.vert2elems) if bridge == 0 else (mesh.side2elems_off, mesh.side2elems)

    def distance_from(seed):
        dist = np.where(seed, 0, -1)
        level, frontier = 0, seed.copy()
        while frontier.any():
            touched = np.zeros(len(off) - 1, dtype=bool)
            rows = np.repeat(np.arange(len(off) - 1), np.diff(off))
            np.logical_or.at(touched, rows, frontier[vals])
            nxt = np.zeros(ne, dtype=bool)
            nxt[vals[touched[rows]]] = True
            nxt &= dist < 0
            level += 1
            dist[nxt] = level
            frontier = nxt
        return dist

    d_core = distance_from(owner == rank)
    for safe_layers, ghost_layers in ((0, 0), (1, 3), (2, 2), (3, 5)):
        safe, part = ppo.bfs_buffer_layers(mesh, owner, rank, nranks, safe_layers, ghost_layers, bridge)
        assert np.array_equal(safe.astype(bool), (d_core >= 0) & (d_core <= safe_layers))
        exp_part = np.zeros(nranks, dtype=np.int32)
        exp_part[np.unique(owner[(d_core >= 0) & (d_core <= ghost_layers)])] = 1
        assert np.array_equal(part, exp_part) and part[rank] == 1
        inward = ppo.bfs_safe_inward(mesh, owner, rank, safe_layers, part, bridge)
        unbuffered = ~part[owner].astype(bool)
        if unbuffered.any():
            d_out = distance_from(unbuffered)
            exp = (d_out > safe_layers) | (owner == rank)
        else:
            exp = np.ones(ne, dtype=bool)
        assert np.array_equal(inward.astype(bool), exp)


@pytest.mark.parametrize("dim", [2, 3])
def test_element_ids_do_not_depend_on_the_sincos_variant(ppo, synth, dim):
    """The parity chain is  reference (libm cos/sin)  ==  oracle(trig=0)  ~(1 ulp)~  oracle(trig=1)  ==bits==
    GPU.  The middle link is closed here for the quantity the north star calls bit-exact: the
    ELEMENT IDS of >= 1 M particles over 20 push + search + rebuild steps of the pseudoXGCm loop
    (deg 0.5, the ctest value) are identical under both sincos variants, in 2-D and on tets; the
    positions differ by a few 1e-16 at most."""
    n = 1_000_000
    if dim == 2:
        coords, e2v, cls = synth.annulus_tri(n_b=49, n_theta=256)
        mdl = 12
    else:
        coords, e2v, cls = synth.torus_tet(n_b=10, n_theta=50, n_planes=12)
        mdl = 8
    ppe = synth.xgcm_source_counts(cls, n, mdl, remainder="spread")
    elem, xyz = synth.particles_in_elements(coords, e2v, ppe)
    R = np.hypot(xyz[0], xyz[1]) if dim == 3 else xyz[0]
    Z = xyz[2] if dim == 3 else xyz[1]
    b, phi = synth.elliptical_state(R, Z)
    info = [xyz, np.zeros_like(xyz), np.arange(n, dtype=np.int32), b, phi]
    ne = len(e2v)
    mesh = ppo.Mesh(dim, coords, e2v, cls)
    runs = []
    nthr = ppo.max_threads()
    ppo.set_threads(nthr)  # the per-particle loops do not depend on the thread count
    try:
        for trig in (0, 1):
            ps = ppo.PS.scs(ppo.PARTICLE_XGCM, ne, ppe, C_max=1, particle_elements=elem, particle_info=info)
            per_step = []
            for _ in range(20):
                if dim == 2:
                    ppo.elliptical_push(ps, mesh, synth.XGC_H, synth.XGC_K, synth.XGC_D, 0.5, trig=trig)
                    _, ids, _ = ppo.search_mesh_2d(mesh, ps, looplimit=200)
                else:
                    ppo.toroidal_push(ps, mesh, synth.XGC_H, synth.XGC_K, synth.XGC_D, 0.5, trig=trig)
                    ids = ppo.search_mesh(mesh, ps, looplimit=200)["elem_ids"]
                cap = ps.capacity()
                live = np.flatnonzero(ps.slot_info()[1])
                pid = ps.member(2)[0, :cap][live]
                order = np.argsort(pid)
                per_step.append((pid[order], ids[:cap][live][order], ps.member(1)[:, :cap][:, live][:, order]))
                ppo.update_positions(ps)
                ps.rebuild(ids)
            runs.append(per_step)
    finally:
        ppo.set_threads(1)
    moved = 0
    for (pa, ea, xa), (pb, eb, xb) in zip(*runs):
        assert np.array_equal(pa, pb)
        assert np.array_equal(ea, eb)                      # element ids: bit-exact
        assert np.abs(xa - xb).max() <= 1e-13              # positions: 1 ulp of sincos, accumulated
        moved += 1
    assert moved == 20 and len(runs[0][-1][0]) > 0.9 * n


def test_redistribute_particles_by_strategy(ppo):
    """redistribute_particles with the re-draw of distribute_particles' strategies (Distribute.cpp:76-253):
    2 gaussian(ne/2, ne/8), 3 exponential conversion, 4 GITRm approximation; strategy 1 is the uniform call"""
    ne = 20000
    ppe = np.full(ne, 5, dtype=np.int32)
    elem = np.repeat(np.arange(ne, dtype=np.int32), ppe)
    n = len(elem)
    ps = ppo.PS.scs(ppo.PARTICLE_PUSH, ne, ppe, C_max=32, particle_elements=elem,
                    particle_info=[np.zeros((3, n)), np.zeros((3, n)), np.arange(n, dtype=np.int32)])
    se, mk = ps.slot_info()
    live = mk.astype(bool)
    assert np.array_equal(ppo.redistribute_particles_dist(ps, 1, 0.4, seed=9), ppo.redistribute_particles(ps, 0.4, seed=9))
    for strat in (2, 3, 4):
        a = ppo.redistribute_particles_dist(ps, strat, 1.0, seed=5)
        assert np.all(a[~live] == -1) and a[live].min() >= 0 and a[live].max() < ne
        assert np.array_equal(a, ppo.redistribute_particles_dist(ps, strat, 1.0, seed=5))
        half = ppo.redistribute_particles_dist(ps, strat, 0.5, seed=5)[live]
        assert abs((half != se[live]).mean() - 0.5) < 0.02
    g = ppo.redistribute_particles_dist(ps, 2, 1.0, seed=5)[live]
    assert abs(g.mean() / ne - 0.5) < 0.01 and abs(g.std() / ne - 0.125) < 0.005
    x = ppo.redistribute_particles_dist(ps, 3, 1.0, seed=5)[live]
    assert x.mean() / ne < 0.15 and (x < ne // 2).mean() > 0.95      # mass near element 0
    t = ppo.redistribute_particles_dist(ps, 4, 1.0, seed=5)[live]
    assert abs((t < 2 * ne // 5).mean() - 0.85) < 0.01
