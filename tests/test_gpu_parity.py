"""GPU parity: every HIP entry point, called through the C-ABI, against the CPU oracle on the
same seeded inputs.  Bar: bit-exact for ids / indices / integer work AND for positions that go
through the shared deterministic sincos; float32-ulp tolerance only where the device libm is used
(ellipticalPush::setup, a one-time initialisation)."""
import os

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

H, K, D = 1.72479370 - .08, .020558260, 0.6


@pytest.fixture(scope="module")
def capi(pp):
    from pumipic_amd import capi as c
    c.init(0)
    return c


def _live(ps_o, ps_g):
    so, mo = ps_o.slot_info()
    sg, mg = ps_g.slot_info()
    return so, mo, sg, mg


# ---------------------------------------------------------------- mesh derivation
@pytest.mark.parametrize("which", ["plate", "box", "annulus", "torus"])
def test_mesh_arrays_match_oracle(ppo, synth, capi, which):
    if which == "plate":
        dim, (c, e, cl) = 2, synth.plate_tri8_pardiag()
    elif which == "box":
        dim, (c, e, cl) = 3, synth.kuhn_box(3)
    elif which == "annulus":
        dim, (c, e, cl) = 2, synth.annulus_tri(n_b=6, n_theta=24, band_width=2)
    else:
        dim, (c, e, cl) = 3, synth.torus_tet(n_b=4, n_theta=12, n_planes=6)
    mo = ppo.Mesh(dim, c, e, cl)
    mg = capi.Mesh(dim, c, e, cl)
    assert (mg.nsides, mg.nelems, mg.nverts) == (mo.nsides, mo.nelems, mo.nverts)
    pairs = [(capi.MESH_ELEM2SIDES, mo.elem2sides), (capi.MESH_SIDE2VERTS, mo.side2verts),
             (capi.MESH_SIDE2ELEMS_OFF, mo.side2elems_off), (capi.MESH_SIDE2ELEMS, mo.side2elems),
             (capi.MESH_SIDE_EXPOSED, mo.side_exposed), (capi.MESH_ELEM_MEASURE, mo.elem_measure),
             (capi.MESH_DUAL_OFF, mo.dual_off), (capi.MESH_DUAL_ELEMS, mo.dual_elems),
             (capi.MESH_VERT2ELEMS_OFF, mo.vert2elems_off), (capi.MESH_VERT2ELEMS, mo.vert2elems)]
    for wid, ref in pairs:
        assert np.array_equal(mg.array(wid), np.asarray(ref).ravel()), wid
    assert mg.tolerance() == mo.tolerance()


# ---------------------------------------------------------------- structure construction
@pytest.mark.parametrize("C,V,sigma", [(64, 1024, 2**31 - 1), (4, 2, 1), (32, 3, 7), (1, 1024, 2**31 - 1)])
@pytest.mark.parametrize("pad", [0, 1, 2])
def test_scs_layout_matches_oracle(ppo, synth, capi, C, V, sigma, pad):
    pop = common.population_2d(synth, num_ptcls=1500)
    ne = len(pop["e2v"])
    po = ppo.PS.scs(ppo.PARTICLE_XGCM, ne, pop["ppe"], C_max=C, sigma=sigma, V=V, pad_strat=pad,
                    particle_elements=pop["elem"], particle_info=pop["info"])
    pg = capi.PS.scs(capi.PARTICLE_XGCM, ne, pop["ppe"], C_=C, sigma=sigma, V=V, pad_strat=pad,
                     particle_elements=pop["elem"], particle_info=pop["info"])
    lo, lg = po.layout(), pg.layout()
    for k in ("C", "num_chunks", "num_slices", "capacity", "num_rows"):
        assert lo[k] == lg[k], k
    for k in ("offsets", "slice_to_chunk", "row_to_element", "element_to_row", "mask"):
        assert np.array_equal(lo[k], lg[k]), k
    so, mo = po.slot_info()
    assert np.array_equal(so, lg["slot_elem"])
    cap = lo["capacity"]
    for m in range(5):
        a, b = po.member(m)[:, :cap], pg.member(m)[:, :cap]
        live = mo.astype(bool)
        assert np.array_equal(a[:, live], b[:, live]), m
    assert po.metrics() == pg.metrics()


def test_csr_layout_matches_oracle(ppo, synth, capi):
    pop = common.population_2d(synth, num_ptcls=1500)
    ne = len(pop["e2v"])
    po = ppo.PS.csr(ppo.PARTICLE_XGCM, ne, pop["ppe"], particle_elements=pop["elem"],
                    particle_info=pop["info"])
    pg = capi.PS.csr(capi.PARTICLE_XGCM, ne, pop["ppe"], particle_elements=pop["elem"],
                     particle_info=pop["info"])
    assert po.capacity() == pg.capacity() and po.nPtcls() == pg.nPtcls()
    assert np.array_equal(po.layout()["offsets"], pg.layout()["offsets"])
    n = po.nPtcls()
    for m in range(5):
        assert np.array_equal(po.member(m)[:, :n], pg.member(m)[:, :n])


def test_empty_structure(capi, synth):
    c, e, cl = synth.plate_tri8_pardiag()
    mesh = capi.Mesh(2, c, e, cl)
    ps = capi.PS.scs(capi.PARTICLE_XGCM, 8, np.zeros(8, np.int32))
    assert ps.nPtcls() == 0 and ps.capacity() == 0
    capi.elliptical_push(ps, mesh, H, K, D, 0.5)
    found, _ = capi.search_mesh_2d(mesh, ps, looplimit=10)
    assert found


# ---------------------------------------------------------------- pushes
@pytest.mark.parametrize("kind", ["scs", "csr"])
def test_elliptical_push_bitwise(ppo, synth, capi, kind):
    pop = common.population_2d(synth)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, kind)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, kind)
    for _ in range(3):
        ppo.elliptical_push(po, mo, H, K, D, 0.5, trig=1)
        capi.elliptical_push(pg, mg, H, K, D, 0.5)
    cap = po.capacity()
    _, mask = po.slot_info()
    live = mask.astype(bool)
    assert np.array_equal(po.member(1)[:, :cap][:, live], pg.member(1)[:, :cap][:, live])
    assert np.array_equal(po.member(4)[:, :cap][:, live], pg.member(4)[:, :cap][:, live])
    # and the shared sincos stays within 1e-15 relative of the literal libm push
    mo2, po2 = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, kind)
    for _ in range(3):
        ppo.elliptical_push(po2, mo2, H, K, D, 0.5, trig=0)
    a, b = po.member(1)[:2, :cap][:, live], po2.member(1)[:2, :cap][:, live]
    assert np.abs(a - b).max() <= 1e-15 * np.abs(b).max() * 4


def test_elliptical_setup_float_tolerance(ppo, synth, capi):
    pop = common.population_2d(synth)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    ppo.elliptical_setup(po, H, K, D)
    capi.elliptical_setup(pg, H, K, D)
    cap = po.capacity()
    live = po.slot_info()[1].astype(bool)
    for m in (3, 4):
        a, b = po.member(m)[0, :cap][live], pg.member(m)[0, :cap][live]
        # device atan2/sin are libm-grade (not bit-identical): one float32 ulp
        assert np.all(np.abs(a - b) <= np.spacing(np.abs(a).astype(np.float32))), m


def test_toroidal_linear_update_bitwise(ppo, synth, capi):
    pop = common.population_3d(synth)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    cap = po.capacity()
    live = po.slot_info()[1].astype(bool)
    ppo.toroidal_push(po, mo, H, K, D, 2.0, trig=1)
    capi.toroidal_push(pg, mg, H, K, D, 2.0)
    assert np.array_equal(po.member(1)[:, :cap][:, live], pg.member(1)[:, :cap][:, live])
    assert np.array_equal(po.member(4)[:, :cap][:, live], pg.member(4)[:, :cap][:, live])
    ppo.update_positions(po)
    capi.update_positions(pg)
    assert np.array_equal(po.member(0)[:, :cap], pg.member(0)[:, :cap])
    assert np.array_equal(po.member(1)[:, :cap], pg.member(1)[:, :cap])
    ppo.linear_push(po, 0.05, -0.5, 0.8, 0.1)
    capi.linear_push(pg, 0.05, -0.5, 0.8, 0.1)
    assert np.array_equal(po.member(1)[:, :cap][:, live], pg.member(1)[:, :cap][:, live])


def test_boris_bitwise(ppo, capi):
    rng = np.random.default_rng(3)
    n = 1000
    host = [rng.normal(size=n) for _ in range(15)]
    host[12:] = [h * 1e-6 for h in host[12:]]
    ref = [h.copy() for h in host]
    ppo.push_boris(*ref, 1e-9)
    dev = [capi.DevArray.from_host(h) for h in host]
    capi.push_boris(dev, 1e-9)
    for i in range(9):
        assert np.array_equal(dev[i].to_host(), ref[i]), i


def test_pseudo_push160_bitwise(ppo, synth, capi):
    ne, np_ = 200, 5000
    ppe, epp = synth.distribute_particles(ne, np_, 2, seed=0)
    po = ppo.PS.scs(ppo.PERF160, ne, ppe, C_max=64, sigma=ne, V=1024)
    pg = capi.PS.scs(capi.PERF160, ne, ppe, C_=64, sigma=ne, V=1024)
    ped = np.sqrt(np.arange(ne, dtype=np.float64)) * np.arange(ne)
    ppo.pseudo_push160(po, ped)
    capi.pseudo_push160(pg, capi.DevArray.from_host(ped))
    cap = po.capacity()
    for m in range(3):
        assert np.array_equal(po.member(m)[:, :cap], pg.member(m)[:, :cap], equal_nan=True), m


# ---------------------------------------------------------------- searches
def _pushed_2d(ppo, capi, synth, kind="scs", deg=3.0, steps=1):
    pop = common.population_2d(synth)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, kind)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, kind)
    for _ in range(steps):
        ppo.elliptical_push(po, mo, H, K, D, deg, trig=1)
        capi.elliptical_push(pg, mg, H, K, D, deg)
    return pop, mo, po, mg, pg


@pytest.mark.parametrize("kind", ["scs", "csr"])
def test_search_mesh_2d_exact(ppo, synth, capi, kind):
    pop, mo, po, mg, pg = _pushed_2d(ppo, capi, synth, kind, deg=8.0)
    found_o, ids_o, loops = ppo.search_mesh_2d(mo, po, looplimit=200)
    found_g, ids_g = capi.search_mesh_2d(mg, pg, looplimit=200)
    assert found_o == found_g and loops > 1
    assert np.array_equal(ids_o, ids_g.to_host()[:po.capacity()])
    live = po.slot_info()[1].astype(bool)
    assert (ids_o[live] != po.slot_info()[0][live]).mean() > 0.05  # particles really moved


def test_search_mesh_2d_looplimit(ppo, synth, capi):
    pop, mo, po, mg, pg = _pushed_2d(ppo, capi, synth, deg=40.0)
    found_o, ids_o, _ = ppo.search_mesh_2d(mo, po, looplimit=3)
    found_g, ids_g = capi.search_mesh_2d(mg, pg, looplimit=3)
    assert (not found_o) and (not found_g)
    assert np.array_equal(ids_o, ids_g.to_host()[:po.capacity()])


@pytest.mark.parametrize("mt", [False, True])
def test_search_mesh_tpp_2d_exact(ppo, synth, capi, mt):
    pop, mo, po, mg, pg = _pushed_2d(ppo, capi, synth, deg=6.0)
    # tpp search needs x_orig inside the start element: positions are the seeded ones
    ro = ppo.search_mesh(mo, po, require_intersection=mt, looplimit=500)
    rg = capi.search_mesh(mg, pg, require_intersection=mt, looplimit=500)
    cap = po.capacity()
    assert ro["found"] == rg["found"] and ro["not_in_elem"] == rg["not_in_elem"]
    assert np.array_equal(ro["elem_ids"], rg["elem_ids"].to_host()[:cap])
    if mt:
        assert np.array_equal(ro["inter_faces"], rg["inter_faces"].to_host()[:cap])
        assert np.array_equal(ro["inter_points"].ravel(), rg["inter_points"].to_host()[:cap * 2])
        assert (ro["inter_faces"] >= 0).any()


@pytest.mark.parametrize("mt", [False, True])
def test_search_mesh_tpp_3d_exact(ppo, synth, capi, mt):
    pop = common.population_3d(synth)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    ppo.toroidal_push(po, mo, H, K, D, 12.0, trig=1)
    capi.toroidal_push(pg, mg, H, K, D, 12.0)
    ro = ppo.search_mesh(mo, po, require_intersection=mt, looplimit=500)
    rg = capi.search_mesh(mg, pg, require_intersection=mt, looplimit=500)
    cap = po.capacity()
    assert ro["found"] == rg["found"] and ro["not_in_elem"] == rg["not_in_elem"] == 0
    assert np.array_equal(ro["elem_ids"], rg["elem_ids"].to_host()[:cap])
    live = po.slot_info()[1].astype(bool)
    assert (ro["elem_ids"][live] != po.slot_info()[0][live]).mean() > 0.05
    if mt:
        assert np.array_equal(ro["inter_faces"], rg["inter_faces"].to_host()[:cap])
        assert np.array_equal(ro["inter_points"].ravel(), rg["inter_points"].to_host()[:cap * 3])


@pytest.mark.parametrize("kind", ["scs", "csr"])
@pytest.mark.parametrize("looplimit,deg", [(2000, 12.0), (3, 12.0), (2000, 0.0), (40, 30.0)])
def test_search_mesh_intersection_mode_packed_walk(ppo, synth, capi, kind, looplimit, deg):
    """search_mesh with requireIntersection on tets (adjacency.tpp:284-361: Moeller-Trumbore per face, the ray
    followed to the domain boundary) through the packed-record walk with lane refill (k_search_mt3): parents,
    exit faces and intersection points bit-equal to the oracle -- unseeded and seeded (with wrong parents and
    deleted particles, tpp:516-522, 72-145), loop limits that cut rays off (tpp:584-606), a push of zero
    degrees (every particle 'unmoved', tpp:525-533), SCS and CSR (tail slots).  (tools/gpu_test_matrix.sh
    runs the suite with PP_MT_PACKED=0 too: the one-thread-per-slot form on the Omega_h-style arrays.)"""
    pop = common.population_3d(synth, n_b=6, n_theta=24, n_planes=10, num_ptcls=4000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, kind=kind)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, kind=kind)
    ppo.toroidal_push(po, mo, H, K, D, deg, trig=1)
    capi.toroidal_push(pg, mg, H, K, D, deg)
    cap = po.capacity()
    slot_e, mask = po.slot_info()
    live = np.flatnonzero(mask)
    seed = slot_e.copy().astype(np.int32)
    seed[~mask.astype(bool)] = -1
    seed[live[::9]] = (seed[live[::9]] + 17) % mo.nelems  # wrong parents -> deleted by check_initial_parents
    seed[live[::11]] = -1                                   # already deleted: stay -1, untouched outputs
    for seeded in (False, True):
        ro = ppo.search_mesh(mo, po, elem_ids=seed.copy() if seeded else None, require_intersection=True,
                             looplimit=looplimit)
        ncap = max(pg.capacity(), 1)
        rg = capi.search_mesh(mg, pg, elem_ids=capi.DevArray.from_host(np.resize(seed, ncap)) if seeded else None,
                              require_intersection=True, looplimit=looplimit)
        assert ro["found"] == rg["found"], (seeded, ro["found"])
        assert ro["not_in_elem"] == rg["not_in_elem"]
        assert np.array_equal(ro["elem_ids"][:cap], rg["elem_ids"].to_host()[:cap])
        assert np.array_equal(ro["inter_faces"][:cap], rg["inter_faces"].to_host()[:cap])
        assert np.array_equal(ro["inter_points"].ravel()[:cap * 3], rg["inter_points"].to_host()[:cap * 3])
        if deg > 0 and looplimit >= 2000:
            assert (ro["inter_faces"][:cap][mask.astype(bool)] >= 0).mean() > 0.5  # the rays do reach the wall
            if not os.environ.get("PP_MT_PACKED") == "0":
                assert capi.search_walk_steps() > 3 * len(live)  # ... through several elements each


def test_mt_face_codes_cover_every_stored_side_order(ppo, synth, capi):
    """The packed walk reads per (tet, face) which tet-local vertex is faceVerts[0], faceVerts[2 - flip],
    faceVerts[flip + 1] of the STORED side (bridgeVerts order + isFaceFlipped, adjacency.tpp:322-331): on a
    mesh whose sides are seen first from either of their two elements both flip values and all three choices
    of the first vertex must occur, or the test above would not exercise them."""
    pop = common.population_3d(synth, n_b=6, n_theta=24, n_planes=10, num_ptcls=10)
    mg, _ = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    e2v = np.asarray(pop["e2v"])
    e2s = mg.array(capi.MESH_ELEM2SIDES).reshape(-1, 4)
    s2v = mg.array(capi.MESH_SIDE2VERTS).reshape(-1, 3)
    tmpl = [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)]
    firsts, flips = set(), set()
    for e in range(0, len(e2v), 7):
        for fi in range(4):
            fv = s2v[e2s[e, fi]]
            loc = [int(np.flatnonzero(e2v[e] == v)[0]) for v in fv]
            assert sorted(loc) == sorted(tmpl[fi])
            firsts.add(tmpl[fi].index(loc[0]))
            k = tmpl[fi].index(loc[0])
            flips.add(loc[1] != tmpl[fi][(k + 1) % 3])
    assert firsts == {0, 1, 2} and flips == {False, True}


def test_search_mesh_tpp_seeded_and_origin_check(ppo, synth, capi):
    """elem_ids passed in (tpp:516-522) + particles whose origin is not in the seed element are
    deleted (check_initial_parents, tpp:72-145)."""
    pop = common.population_3d(synth)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    ppo.toroidal_push(po, mo, H, K, D, 5.0, trig=1)
    capi.toroidal_push(pg, mg, H, K, D, 5.0)
    slot_e, mask = po.slot_info()
    seed = slot_e.copy()
    seed[~mask.astype(bool)] = -1
    live = np.flatnonzero(mask)
    seed[live[::7]] = (seed[live[::7]] + 11) % mo.nelems  # wrong parents -> deleted
    seed[live[::13]] = -1                                   # already-deleted particles stay -1
    ro = ppo.search_mesh(mo, po, elem_ids=seed.copy(), looplimit=300)
    rg = capi.search_mesh(mg, pg, elem_ids=capi.DevArray.from_host(seed), looplimit=300)
    assert ro["not_in_elem"] == rg["not_in_elem"] > 0
    assert np.array_equal(ro["elem_ids"], rg["elem_ids"].to_host()[:po.capacity()])


@pytest.mark.parametrize("C", [1, 64])
def test_search_mesh_3d_exact(ppo, synth, capi, C):
    """search_mesh_3d (adjacency.hpp:314-555) in the pseudoPushAndSearch loop: ids, wall faces and
    wall points bit-identical by particle id, step after step."""
    pop = common.population_box(synth, n=4, num_ptcls=900)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=C)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_PUSH, C=C)
    hits = 0
    for step in range(8):
        ppo.linear_push(po, 0.11, -0.5, 0.8, 0.07 * step)
        capi.linear_push(pg, 0.11, -0.5, 0.8, 0.07 * step)
        ro = ppo.search_mesh_3d(mo, po, looplimit=100)
        rg = capi.search_mesh_3d(mg, pg, looplimit=100)
        assert ro["found"] == rg["found"] == 1
        capo, capg = po.capacity(), pg.capacity()
        ids_g = rg["elem_ids"].to_host()[:capg]
        mko, mkg = po.slot_info()[1], pg.slot_info()[1]
        pido, pidg = po.member(2)[0, :capo], pg.member(2)[0, :capg]
        io, eo = common.by_id(pido, mko, ro["elem_ids"])
        ig, eg = common.by_id(pidg, mkg, ids_g)
        assert np.array_equal(io, ig) and np.array_equal(eo, eg), step
        _, fo = common.by_id(pido, mko, ro["xface"])
        _, fg = common.by_id(pidg, mkg, rg["xface"].to_host()[:capg])
        assert np.array_equal(fo, fg)
        _, xo = common.by_id(pido, mko, ro["xpoints"].T)
        _, xg = common.by_id(pidg, mkg, rg["xpoints"].to_host().reshape(-1, 3)[:capg].T)
        assert np.array_equal(xo[:, fo >= 0], xg[:, fg >= 0])
        hits += int((fo >= 0).sum())
        ppo.update_positions(po)
        capi.update_positions(pg)
        po.rebuild(ro["elem_ids"])
        pg.rebuild(ids_g)
        assert po.nPtcls() == pg.nPtcls()
        if po.nPtcls() == 0:
            break
    assert hits > 0


def test_search_mesh_3d_seeded_limit_and_abort(ppo, synth, capi):
    """seed ids with -1 entries, a loop limit that leaves walks unfinished (hpp:531-552) and the
    checkParent failure (hpp:371-382) report the same as the oracle; torus population."""
    pop = common.population_3d(synth, num_ptcls=4000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    ppo.toroidal_push(po, mo, H, K, D, 14.0, trig=1)
    capi.toroidal_push(pg, mg, H, K, D, 14.0)
    se, mk = po.slot_info()
    cap = po.capacity()
    seed = se.copy()
    seed[~mk.astype(bool)] = -1
    seed[np.flatnonzero(mk)[::9]] = -1
    for limit in (0, 3):
        ro = ppo.search_mesh_3d(mo, po, elem_ids=seed.copy(), looplimit=limit)
        rg = capi.search_mesh_3d(mg, pg, elem_ids=capi.DevArray.from_host(seed), looplimit=limit)
        assert ro["found"] == rg["found"] == (1 if limit == 0 else 0)
        assert np.array_equal(ro["elem_ids"], rg["elem_ids"].to_host()[:cap])
        assert np.array_equal(ro["xface"], rg["xface"].to_host()[:cap])
        assert np.array_equal(ro["xpoints"].ravel(), rg["xpoints"].to_host()[:cap * 3])
    s = np.flatnonzero(mk)[5]
    po.member(0)[:, s] += 7.0
    xg = pg.member(0)
    xg[:, s] += 7.0
    pg.set_member(0, xg)
    assert ppo.search_mesh_3d(mo, po, looplimit=50)["found"] == -2
    assert capi.search_mesh_3d(mg, pg, looplimit=50)["found"] == -2


def test_search_mesh_legacy3d_exact(ppo, synth, capi):
    """pseudoPushAndSearch loop (test/pseudoPushAndSearch.cpp:513-542): push, legacy search,
    rebuild.  After a rebuild slot order inside a row is free, so compare by particle id."""
    pop = common.population_box(synth, n=4, num_ptcls=600)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_PUSH)
    hits = 0
    for step in range(8):
        ppo.linear_push(po, 1.0 / 20, -0.5, 0.8, 0.0)
        capi.linear_push(pg, 1.0 / 20, -0.5, 0.8, 0.0)
        ro = ppo.search_mesh_legacy3d(mo, po, looplimit=100)
        rg = capi.search_mesh_legacy3d(mg, pg, looplimit=100)
        assert ro["found"] == rg["found"] == 1
        capo, capg = po.capacity(), pg.capacity()
        ids_g = rg["elem_ids"].to_host()[:capg]
        mko, mkg = po.slot_info()[1], pg.slot_info()[1]
        pido, pidg = po.member(2)[0, :capo], pg.member(2)[0, :capg]
        io, eo = common.by_id(pido, mko, ro["elem_ids"])
        ig, eg = common.by_id(pidg, mkg, ids_g)
        assert np.array_equal(io, ig) and np.array_equal(eo, eg), step
        _, fo = common.by_id(pido, mko, ro["xface"])
        _, fg = common.by_id(pidg, mkg, rg["xface"].to_host()[:capg])
        assert np.array_equal(fo, fg)
        _, xo = common.by_id(pido, mko, ro["xpoints"].T)
        _, xg = common.by_id(pidg, mkg, rg["xpoints"].to_host().reshape(-1, 3)[:capg].T)
        assert np.array_equal(xo[:, fo >= 0], xg[:, fg >= 0])
        hits += int((fo >= 0).sum())
        ppo.update_positions(po)
        capi.update_positions(pg)
        po.rebuild(ro["elem_ids"])
        pg.rebuild(ids_g)
        assert po.nPtcls() == pg.nPtcls()
        if po.nPtcls() == 0:
            break
    assert hits > 0  # some particles reached the wall


# ---------------------------------------------------------------- fused hot path
@pytest.mark.parametrize("dim", [2, 3])
def test_fused_push_search_equals_unfused(ppo, synth, capi, dim):
    pop = common.population_2d(synth) if dim == 2 else common.population_3d(synth)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    cap = po.capacity()
    ids_g = capi.DevArray.from_host(np.full(cap, -1, dtype=np.int32))
    ids_o = None
    for step in range(4):
        if dim == 2:
            ppo.elliptical_push(po, mo, H, K, D, 4.0, trig=1)
            _, ids_o, _ = ppo.search_mesh_2d(mo, po, elem_ids=ids_o, looplimit=200)
            capi.push_search(mg, pg, H, K, D, 4.0, ids_g, seeded=True, looplimit=200)
        else:
            ppo.toroidal_push(po, mo, H, K, D, 6.0, trig=1)
            r = ppo.search_mesh(mo, po, elem_ids=ids_o, looplimit=200)
            ids_o = r["elem_ids"]
            capi.push_search(mg, pg, H, K, D, 6.0, ids_g, seeded=(step > 0), looplimit=200)
        assert np.array_equal(ids_o, ids_g.to_host()[:cap]), step
        live = po.slot_info()[1].astype(bool)
        assert np.array_equal(po.member(1)[:, :cap][:, live], pg.member(1)[:, :cap][:, live])
        assert np.array_equal(po.member(4)[0, :cap][live], pg.member(4)[0, :cap][live])
        if dim == 3:  # no rebuild: ping-pong x <-> x_tgt like BASELINE config 2
            a, b = po.member(0), po.member(1)
            tmp = a.copy()
            a[:] = b
            b[:] = tmp
            pg.swap_members(0, 1)


# ---------------------------------------------------------------- scatter
def test_gyro_maps_and_scatter(ppo, synth, capi):
    pop = common.population_2d(synth, num_ptcls=4000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    fo, bo = ppo.create_gyro_ring_mappings(mo, 0.038, 3, 8, 0.0, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg, 0.038, 3, 8, 0.0)
    assert np.array_equal(fo, fg.to_host()) and np.array_equal(bo, bg.to_host())
    assert (fo >= 0).mean() > 0.8
    wo = ppo.gyro_scatter(mo, po, fo, 0.038, 3, 8)
    wg = capi.gyro_scatter(mg, pg, fg, 0.038, 3, 8).to_host()
    assert wo.sum() > 0
    assert np.array_equal(wo, wg)  # multiples of 1/8: exact in any order
    packed = capi.gyro_sync_pack(mg.nverts, capi.DevArray.from_host(wo),
                                 capi.DevArray.from_host(2 * wo)).to_host()
    assert np.array_equal(packed[0::2], wo) and np.array_equal(packed[1::2], 2 * wo)


def test_gyro_scatter_reference_kat(synth, capi):
    """test/pseudoXGCm_scatter.cpp:115-178 through the HIP path."""
    c, e, cl = synth.plate_tri8_pardiag()
    mesh = capi.Mesh(2, c, e, cl)
    fwd, _ = capi.create_gyro_ring_mappings(mesh, .2, 2, 6, 15)
    m = fwd.to_host().reshape(9, 2 * 6 * 3).copy()
    keep = m[3].copy()
    m[:] = 2
    m[3] = keep
    ppe = np.zeros(8, np.int32)
    ppe[0] = 1
    ps = capi.PS.scs(capi.PARTICLE_XGCM, 8, ppe, C_=64, V=32)
    w = capi.gyro_scatter(mesh, ps, capi.DevArray.from_host(m.reshape(-1)), .2, 2, 6).to_host()
    for i in range(9):
        expect = {3: 2.0, 2: 12.0, 8: 0.0}.get(i, 2.0 / 3.0)
        assert abs(w[i] - expect) <= 1e-12 * max(1.0, expect), (i, w[i])


def test_avg_density(ppo, synth, capi):
    pop = common.population_box(synth, n=3, num_ptcls=400)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_PUSH)
    eo, vo = ppo.avg_ptcl_density(mo, po)
    eg, vg = capi.avg_ptcl_density(mg, pg)
    assert np.array_equal(eo, eg.to_host()) and np.array_equal(vo, vg.to_host())


# ---------------------------------------------------------------- rebuild
def _check_same_population(po, pg, members):
    """multiset equality keyed by the particle-id member (member 2)"""
    so, mo = po.slot_info()
    sg, mg = pg.slot_info()
    capo, capg = po.capacity(), pg.capacity()
    ido, idg = po.member(2)[0, :capo], pg.member(2)[0, :capg]
    io, eo = common.by_id(ido, mo, so)
    ig, eg = common.by_id(idg, mg, sg)
    assert np.array_equal(io, ig)
    assert np.array_equal(eo, eg)
    for m in range(len(members)):
        _, a = common.by_id(ido, mo, po.member(m)[:, :capo])
        _, b = common.by_id(idg, mg, pg.member(m)[:, :capg])
        assert np.array_equal(a, b), m


@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("kind,C,V,sigma,pad", [("scs", 64, 1024, 2**31 - 1, 0), ("scs", 8, 4, 16, 1),
                                                ("scs", 32, 16, 2**31 - 1, 2), ("csr", 0, 0, 0, 0)])
def test_rebuild_matches_oracle(ppo, synth, capi, kind, C, V, sigma, pad, shuffle):
    pop = common.population_2d(synth, num_ptcls=5000)
    ne = len(pop["e2v"])
    if kind == "scs":
        po = ppo.PS.scs(ppo.PARTICLE_XGCM, ne, pop["ppe"], C_max=C, sigma=sigma, V=V, pad_strat=pad,
                        particle_elements=pop["elem"], particle_info=pop["info"])
        pg = capi.PS.scs(capi.PARTICLE_XGCM, ne, pop["ppe"], C_=C, sigma=sigma, V=V, pad_strat=pad,
                         particle_elements=pop["elem"], particle_info=pop["info"])
        common.set_shuffling(po, pg, on=shuffle)
    else:
        if shuffle:
            pytest.skip("CSR has no in-place rebuild (csr/CSR_rebuild.hpp)")
        po = ppo.PS.csr(ppo.PARTICLE_XGCM, ne, pop["ppe"], particle_elements=pop["elem"],
                        particle_info=pop["info"])
        pg = capi.PS.csr(capi.PARTICLE_XGCM, ne, pop["ppe"], particle_elements=pop["elem"],
                         particle_info=pop["info"])
    rng = np.random.default_rng(7)
    next_id = 5000
    for it in range(4):
        so, mo = po.slot_info()
        sg, mg = pg.slot_info()
        ido = po.member(2)[0, :po.capacity()]
        idg = pg.member(2)[0, :pg.capacity()]
        # per-particle decision keyed by id so both structures apply the same moves
        dec = rng.integers(0, ne, size=next_id).astype(np.int32)
        stay = rng.random(next_id) < (0.5 if it % 2 == 0 or not shuffle else 0.97)
        dele = rng.random(next_id) < (0.1 if it % 2 == 0 or not shuffle else 0.01)
        def new_elems(slot_e, mask, ids):
            out = np.full(len(slot_e), -1, dtype=np.int32)
            live = mask.astype(bool)
            i = ids[live]
            ne_ = np.where(stay[i], slot_e[live], dec[i])
            ne_ = np.where(dele[i], -1, ne_)
            out[live] = ne_
            return out
        n_new = 37 if it % 2 == 0 else 0
        add_e = rng.integers(0, ne, size=n_new).astype(np.int32)
        add_info = None
        if n_new:
            add_info = [rng.random((3, n_new)), rng.random((3, n_new)),
                        np.arange(next_id, next_id + n_new, dtype=np.int32),
                        rng.random(n_new).astype(np.float32), rng.random(n_new).astype(np.float32)]
            next_id += n_new
        po.rebuild(new_elems(so, mo, ido), add_e if n_new else None, add_info)
        before = pg.rebuild_stats() if kind == "scs" else (0, 0)
        pg.rebuild(new_elems(sg, mg, idg), add_e if n_new else None, add_info)
        if kind == "scs":  # the two sides take the same reshuffle-or-rebuild decision
            assert bool(po.s.last_rebuild_was_shuffle) == (pg.rebuild_stats()[0] > before[0])
        assert po.nPtcls() == pg.nPtcls()
        _check_same_population(po, pg, ppo.PARTICLE_XGCM)
        if kind == "scs":
            lo, lg = po.layout(), pg.layout()
            for k in ("C", "num_chunks", "num_slices", "capacity", "num_rows"):
                assert lo[k] == lg[k], (k, lo[k], lg[k])
            for k in ("offsets", "slice_to_chunk", "row_to_element", "element_to_row"):
                assert np.array_equal(lo[k], lg[k]), k
            # mask: same number of live slots per row (slot order inside a row is free)
            s1, m1 = po.slot_info()
            s2, m2 = pg.slot_info()
            assert np.array_equal(np.bincount(s1[m1 > 0], minlength=ne),
                                  np.bincount(s2[m2 > 0], minlength=ne))
        else:
            assert np.array_equal(po.layout()["offsets"], pg.layout()["offsets"])
        # getPIDs property (ps_for.hpp:65-85)
        off, pids = pg.get_pids()
        se, ms = pg.slot_info()
        assert np.all(ms[pids] == 1)
        assert np.array_equal(np.repeat(np.arange(ne), np.diff(off)), se[pids])


@pytest.mark.parametrize("heavy,ne", [(6, 3000), (1100, 3000), (6, 140000), (1100, 140000)])
def test_rebuild_layout_sort_with_heavy_rows(ppo, capi, heavy, ne):
    """The one-pass layout sort (k_rs_pass_wide) keeps rows of 2047 and more particles in one digit and orders
    them in the layout kernel (up to 1024 rows; more than that: the 8-bit passes run instead).  Heavy rows with
    ties, light rows around them, rows that become heavy / light from one rebuild to the next: layout arrays
    and population equal the oracle's (stable ascending order, SCS_sort.h).  140 000 elements = 69 sort tiles:
    the digit table is prefixed by its own two kernels instead of being swept by every block."""
    rng = np.random.default_rng(heavy)
    n = heavy * 2060 + 80000 + 20 * ne

    def counts(step):
        idx = rng.permutation(ne)
        c = np.zeros(ne, dtype=np.int64)
        c[idx[:heavy]] = rng.integers(2040, 2060, size=heavy)  # around the digit boundary, many ties
        c[idx[:3]] = [5000, 5000, 2047] if step % 2 == 0 else [2046, 70000 // (step + 1), 2048]
        light = idx[heavy:]
        c[light] = rng.multinomial(n - int(c.sum()), np.full(len(light), 1.0 / len(light)))
        return c

    c0 = counts(0)
    elem = np.repeat(np.arange(ne, dtype=np.int32), c0)
    xyz = rng.random((3, n))
    info = [xyz, np.zeros_like(xyz), np.arange(n, dtype=np.int32), rng.random(n).astype(np.float32),
            rng.random(n).astype(np.float32)]
    kw = dict(sigma=2**31 - 1, V=1024, pad_strat=0, particle_elements=elem, particle_info=info)
    po = ppo.PS.scs(ppo.PARTICLE_XGCM, ne, c0.astype(np.int32), C_max=64, **kw)
    pg = capi.PS.scs(capi.PARTICLE_XGCM, ne, c0.astype(np.int32), C_=64, **kw)
    common.set_shuffling(po, pg, on=False)
    for step in range(1, 4):
        dest = np.repeat(np.arange(ne, dtype=np.int32), counts(step))[rng.permutation(n)]  # by particle id
        for ps in (po, pg):
            se, mk = ps.slot_info()
            ids = ps.member(2)[0, :ps.capacity()]
            ne_ = np.full(len(se), -1, dtype=np.int32)
            live = mk.astype(bool)
            ne_[live] = dest[ids[live]]
            ps.rebuild(ne_, None, None)
        assert po.nPtcls() == pg.nPtcls() == n
        lo, lg = po.layout(), pg.layout()
        for k in ("C", "num_chunks", "num_slices", "capacity", "num_rows"):
            assert lo[k] == lg[k], (k, lo[k], lg[k])
        for k in ("offsets", "slice_to_chunk", "row_to_element", "element_to_row"):
            assert np.array_equal(lo[k], lg[k]), (step, k)
        _check_same_population(po, pg, ppo.PARTICLE_XGCM)


def test_rebuild_commit_equals_update_then_rebuild(ppo, synth, capi):
    """pp_ps_rebuild_commit == updatePtclPositions + rebuild (pseudoXGCm.cpp:116-140), including
    after an O(1) member swap."""
    pop = common.population_2d(synth, num_ptcls=4000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    common.set_shuffling(po, pg)
    for step in range(3):
        ppo.elliptical_push(po, mo, H, K, D, 3.0, trig=1)
        _, ids_o, _ = ppo.search_mesh_2d(mo, po, looplimit=200)
        ids_g = capi.DevArray.from_host(np.full(max(pg.capacity(), 1), -1, dtype=np.int32))
        capi.push_search(mg, pg, H, K, D, 3.0, ids_g, seeded=True, looplimit=200)
        ppo.update_positions(po)
        po.rebuild(ids_o)
        if step == 1:  # exercise the permutation normalisation
            pg.swap_members(0, 1)
            pg.swap_members(0, 1)
        pg.rebuild_commit(ids_g)
        capi.sync()
        assert po.nPtcls() == pg.nPtcls()
        _check_same_population(po, pg, ppo.PARTICLE_XGCM)
        assert np.array_equal(ppo.gyro_scatter(mo, po, ppo.create_gyro_ring_mappings(mo, trig=1)[0]),
                              capi.gyro_scatter(mg, pg, capi.create_gyro_ring_mappings(mg)[0]).to_host())


def test_rebuild_delete_all_then_refill(ppo, synth, capi):
    pop = common.population_2d(synth, num_ptcls=800)
    ne = len(pop["e2v"])
    pg = capi.PS.scs(capi.PARTICLE_XGCM, ne, pop["ppe"], particle_elements=pop["elem"],
                     particle_info=pop["info"])
    pg.rebuild(np.full(pg.capacity(), -1, dtype=np.int32))
    assert pg.nPtcls() == 0
    # the reference means to clear the mask here (SCS_rebuild.h:168-176) but its resetMask runs after
    # num_ptcls = 0 and parallel_for returns at once; the library does clear it: no ghost particles for
    # callers that walk slots by mask
    assert not pg.slot_info()[1].any()
    n_new = 50
    info = [np.ones((3, n_new)), np.zeros((3, n_new)), np.arange(n_new, dtype=np.int32),
            np.ones(n_new, np.float32), np.ones(n_new, np.float32)]
    pg.rebuild(np.full(max(pg.capacity(), 1), -1, dtype=np.int32), np.arange(n_new) % ne, info)
    assert pg.nPtcls() == n_new
    se, ms = pg.slot_info()
    ids = pg.member(2)[0, :pg.capacity()]
    i, e = common.by_id(ids, ms, se)
    assert np.array_equal(i, np.arange(n_new)) and np.array_equal(e, np.arange(n_new) % ne)


def test_rebuild_rejects_inactive_new_particles(synth, capi):
    pop = common.population_2d(synth, num_ptcls=300)
    ne = len(pop["e2v"])
    pg = capi.PS.scs(capi.PARTICLE_XGCM, ne, pop["ppe"], particle_elements=pop["elem"],
                     particle_info=pop["info"])
    info = [np.ones((3, 1)), np.zeros((3, 1)), np.zeros(1, np.int32), np.ones(1, np.float32),
            np.ones(1, np.float32)]
    with pytest.raises(capi.PPError):
        pg.rebuild(pg.slot_info()[0], np.array([-1]), info)


# ---------------------------------------------------------------- end-to-end pseudoXGCm loop
@pytest.mark.parametrize("kind", ["scs", "csr"])
def test_pseudo_xgcm_steps_2d(ppo, synth, capi, kind):
    """push -> search_mesh_2d -> updatePtclPositions -> rebuild -> gyroScatter x2, 15 steps
    (test/pseudoXGCm.cpp:504-534); element ids by particle id and scatter sums bit-exact."""
    pop = common.population_2d(synth, num_ptcls=4000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, kind)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, kind)
    if kind == "scs":
        common.set_shuffling(po, pg)
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg)
    for step in range(15):
        ppo.elliptical_push(po, mo, H, K, D, 2.0, trig=1)
        _, ids_o, _ = ppo.search_mesh_2d(mo, po, looplimit=200)
        ids_g = capi.DevArray.from_host(np.full(max(pg.capacity(), 1), -1, dtype=np.int32))
        capi.push_search(mg, pg, H, K, D, 2.0, ids_g, seeded=True, looplimit=200)
        so, mko = po.slot_info()
        sg, mkg = pg.slot_info()
        io, eo = common.by_id(po.member(2)[0, :po.capacity()], mko, ids_o)
        ig, eg = common.by_id(pg.member(2)[0, :pg.capacity()], mkg, ids_g.to_host()[:pg.capacity()])
        assert np.array_equal(io, ig) and np.array_equal(eo, eg), step
        ppo.update_positions(po)
        capi.update_positions(pg)
        po.rebuild(ids_o)
        pg.rebuild(ids_g)
        assert po.nPtcls() == pg.nPtcls() > 0
        wo = ppo.gyro_scatter(mo, po, fo)
        wg = capi.gyro_scatter(mg, pg, fg).to_host()
        assert np.array_equal(wo, wg), step
    _check_same_population(po, pg, ppo.PARTICLE_XGCM)


# ---------------------------------------------------------------- migration glue
def test_unsafe_procs_and_pack(ppo, synth, capi):
    pop = common.population_2d(synth, num_ptcls=3000)
    ne = len(pop["e2v"])
    gids = np.arange(ne, dtype=np.int64) * 3 + 5
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    pg = capi.PS.scs(capi.PARTICLE_XGCM, ne, pop["ppe"], gids=gids, particle_elements=pop["elem"],
                     particle_info=pop["info"])
    nranks, rank = 4, 1
    owners = (np.arange(ne) * nranks // ne).astype(np.int32)
    safe = (owners == rank).astype(np.uint8)
    se, mk = pg.slot_info()
    rng = np.random.default_rng(11)
    elems = np.where(mk > 0, rng.integers(0, ne, size=len(se)), -1).astype(np.int32)
    ne_o, np_o = ppo.set_unsafe_procs(po, elems, safe, owners, rank)
    ne_g, np_g = capi.set_unsafe_procs(pg, capi.DevArray.from_host(elems),
                                       capi.DevArray.from_host(safe),
                                       capi.DevArray.from_host(owners), rank)
    cap = pg.capacity()
    assert np.array_equal(ne_o, ne_g.to_host()[:cap]) and np.array_equal(np_o, np_g.to_host()[:cap])
    counts = capi.migrate_count(pg, ne_g, np_g, rank, nranks)
    live = mk > 0
    expect = np.bincount(np_o[live & (np_o != rank)], minlength=nranks)
    assert np.array_equal(counts, expect) and counts[rank] == 0 and counts.sum() > 0
    ids_before = pg.member(2)[0, :cap].copy()
    x_before = pg.member(0)[:, :cap].copy()
    gid, bufs = capi.migrate_pack(pg, ne_g, np_g, rank, nranks, counts)
    total = int(counts.sum())
    sent_ids = bufs[2].to_host()[:total]
    sent_gid = gid.to_host()[:total]
    sent_x = bufs[0].to_host()[:3 * total].reshape(3, total)
    start = np.concatenate([[0], np.cumsum(counts)])
    slot_of = {int(i): s for s, i in enumerate(ids_before) if live[s]}
    for r in range(nranks):
        seg = slice(start[r], start[r + 1])
        for i, g, x in zip(sent_ids[seg], sent_gid[seg], sent_x[:, seg].T):
            s = slot_of[int(i)]
            assert np_o[s] == r and g == gids[ne_o[s]] and np.array_equal(x, x_before[:, s])
    after = ne_g.to_host()[:cap]
    assert np.all(after[live & (np_o != rank)] == -1)
    assert np.array_equal(after[live & (np_o == rank)], ne_o[live & (np_o == rank)])


# ---------------------------------------------------------------- C++ boundary (drivers/)
def test_cpp_driver_pseudoxgcm(synth, capi, tmp_path):
    """The pseudoXGCm driver, written against the particle_structs mirror headers with USER
    lambdas through ps::parallel_for, runs the reference step loop end to end."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "pumi-pic_amd", "drivers")
    subprocess.check_call(["make", "-C", drv, "-s"])
    c, e, cl = synth.annulus_tri(n_b=24, n_theta=96, band_width=3)
    mesh_file = str(tmp_path / "annulus.bin")
    synth.write_mesh_bin(mesh_file, 2, c, e, cl)
    npt = 20000
    out = subprocess.run([os.path.join(drv, "pseudoXGCm"), mesh_file, str(npt), "6", "10", "2.0", "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"RESULT particles (\d+) scatter_mass (\S+) touched_elements (\d+)", out.stdout)
    assert m, out.stdout[-2000:]
    particles, mass, touched = int(m.group(1)), float(m.group(2)), int(m.group(3))
    assert particles == npt                      # interior bands: nobody leaves the domain
    assert 0.9 * 18 * npt <= mass <= 18 * npt    # 2 rings x 3 verts x (8 pts x 3 mapped)/8 each
    assert touched > 0
    assert "Metrics 0, C 64" in out.stdout


@pytest.mark.parametrize("fused", [False, True])
def test_migration_records_two_virtual_ranks(ppo, synth, capi, fused):
    """pp_ps_migrate_pack_records + pp_ps_rebuild_records: two element-block 'ranks' living in one
    process exchange packed records through host memory (standing in for the all-to-all-v); the
    union of both structures must equal the single-rank oracle run, bit for bit.  fused: the
    position commit travels in the records and in the receiver's rebuild, and the two gyroScatter
    calls ride behind it (pp_ps_migrate_pack_records_commit, pp_ps_rebuild_records_scatter); the sum
    of the ranks' fields must equal the oracle's field."""
    pop = common.population_2d(synth, n_b=12, n_theta=48, num_ptcls=3000, mdl_face=3, band_width=3)
    ne = len(pop["e2v"])
    world = 2
    owners = (np.arange(ne) * world // ne).astype(np.int32)
    mesh = capi.Mesh(2, pop["coords"], pop["e2v"], pop["cls"])
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    common.set_shuffling(po)
    ranks = []
    for r in range(world):
        mine = owners[pop["elem"]] == r
        elem = pop["elem"][mine]
        info = [np.ascontiguousarray(a[..., mine]) for a in pop["info"]]
        ps = capi.PS.scs(capi.PARTICLE_XGCM, ne, np.bincount(elem, minlength=ne).astype(np.int32),
                         gids=np.arange(ne, dtype=np.int64), particle_elements=elem, particle_info=info)
        ranks.append(ps)
    recb = capi.migrate_record_bytes(ranks[0])
    assert recb == 80
    owners_d = capi.DevArray.from_host(owners)
    moved = 0
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mesh)
    for step in range(5):
        ppo.elliptical_push(po, mo, H, K, D, 6.0, trig=1)
        _, ids_o, _ = ppo.search_mesh_2d(mo, po, looplimit=200)
        ppo.update_positions(po)
        po.rebuild(ids_o)
        outbox, fields = [], []
        for r, ps in enumerate(ranks):
            ids = capi.DevArray.from_host(np.full(max(ps.capacity(), 1), -1, dtype=np.int32))
            capi.push_search(mesh, ps, H, K, D, 6.0, ids, seeded=True, looplimit=200)
            if not fused:
                capi.update_positions(ps)
            safe = capi.DevArray.from_host((owners == r).astype(np.uint8))
            ne_d, np_d = capi.set_unsafe_procs(ps, ids, safe, owners_d, r)
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
            rbuf = capi.DevArray.from_host(np.ascontiguousarray(recv).reshape(-1))
            if fused:
                wf = capi.DevArray(mesh.nverts, np.float64)
                wb = capi.DevArray(mesh.nverts, np.float64)
                capi.rebuild_records_scatter(ps, outbox[r][0], len(recv), rbuf.ptr, mesh, [fg, bg], [wf, wb])
                fields.append((wf.to_host(), wb.to_host()))
            else:
                capi.rebuild_records(ps, outbox[r][0], len(recv), rbuf.ptr)
            capi.sync()
        if fused:  # gyroSync: the ranks' fields add up to the single-rank field (exact integers / 8)
            assert np.array_equal(sum(f for f, _ in fields), ppo.gyro_scatter(mo, po, fo))
            assert np.array_equal(sum(b for _, b in fields), ppo.gyro_scatter(mo, po, bo))
    assert moved > 0
    ids_all, elem_all, x_all, phi_all = [], [], [], []
    for r, ps in enumerate(ranks):
        se, mk = ps.slot_info()
        cap = ps.capacity()
        live = mk.astype(bool)
        assert np.all(owners[se[live]] == r)
        ids_all.append(ps.member(2)[0, :cap][live])
        elem_all.append(se[live])
        x_all.append(ps.member(0)[:, :cap][:, live])
        phi_all.append(ps.member(4)[0, :cap][live])
    ids_all = np.concatenate(ids_all)
    order = np.argsort(ids_all)
    so, mko = po.slot_info()
    io, eo = common.by_id(po.member(2)[0, :po.capacity()], mko, so)
    assert np.array_equal(ids_all[order], io)
    assert np.array_equal(np.concatenate(elem_all)[order], eo)
    _, xo = common.by_id(po.member(2)[0, :po.capacity()], mko, po.member(0)[:, :po.capacity()])
    assert np.array_equal(np.concatenate(x_all, axis=1)[:, order], xo)


# ---------------------------------------------------------------- fused kernel variants
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("queue", ["0", "1"])
@pytest.mark.parametrize("C,looplimit,deg", [(64, 200, 6.0), (64, 1, 6.0), (64, 2, 12.0), (8, 200, 6.0),
                                             (48, 3, 25.0), (1, 200, 6.0)])
def test_fused_variants_match_oracle(ppo, synth, capi, monkeypatch, dim, queue, C, looplimit, deg):
    """Both row-tiled kernels (walk inside the column loop / deferred walk with the cooperative
    record fetch) on ragged layouts, tight loop limits (particles cut off -> -1 and found == 0)
    and pushes large enough to leave the domain."""
    monkeypatch.setenv("PP_WALK_QUEUE", queue)
    pop = common.population_2d(synth, num_ptcls=2500) if dim == 2 else common.population_3d(synth, num_ptcls=2500)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=C)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, C=C)
    cap = po.capacity()
    assert cap == pg.capacity()
    ids_g = capi.DevArray.from_host(np.full(cap, -1, dtype=np.int32))
    ids_o = None
    for step in range(3):
        if dim == 2:
            ppo.elliptical_push(po, mo, H, K, D, deg, trig=1)
            found_o, ids_o, _ = ppo.search_mesh_2d(mo, po, elem_ids=ids_o, looplimit=looplimit)
            found_g = capi.push_search(mg, pg, H, K, D, deg, ids_g, seeded=True, looplimit=looplimit)
        else:
            ppo.toroidal_push(po, mo, H, K, D, deg, trig=1)
            r = ppo.search_mesh(mo, po, elem_ids=ids_o, looplimit=looplimit)
            ids_o, found_o = r["elem_ids"], r["found"]
            found_g = capi.push_search(mg, pg, H, K, D, deg, ids_g, seeded=(step > 0), looplimit=looplimit)
        live = po.slot_info()[1].astype(bool)
        got = ids_g.to_host()[:cap]
        assert np.array_equal(ids_o[:cap][live], got[live]), (step, int((ids_o[:cap][live] != got[live]).sum()))
        assert bool(found_o) == bool(found_g), step
        assert np.array_equal(po.member(1)[:, :cap][:, live], pg.member(1)[:, :cap][:, live])
        if dim == 3:
            a, b = po.member(0), po.member(1)
            tmp = a.copy()
            a[:] = b
            b[:] = tmp
            pg.swap_members(0, 1)


def test_gyro_maps_and_scatter_3d(ppo, synth, capi):
    """tet variant (documented deviation): ring points in the vertex's poloidal
    half-plane, 4 vertices per ring point; map and scatter sums bit-exact vs the oracle."""
    pop = common.population_3d(synth, n_b=5, n_theta=20, n_planes=8, num_ptcls=3000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    fo, bo = ppo.create_gyro_ring_mappings(mo, rmax=0.03, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg, rmax=0.03)
    assert len(fo) == mo.nverts * 3 * 8 * 4
    assert np.array_equal(fo, fg.to_host()) and np.array_equal(bo, bg.to_host())
    assert (fo >= 0).mean() > 0.5 and (fo < 0).any()  # rings near the wall leave the domain
    wo = ppo.gyro_scatter(mo, po, fo, rmax=0.03)
    wg = capi.gyro_scatter(mg, pg, fg, rmax=0.03).to_host()
    assert np.array_equal(wo, wg) and wo.sum() > 0


@pytest.mark.parametrize("trust", [False, True])
def test_pseudo_xgcm_steps_3d(ppo, synth, capi, trust):
    """BASELINE configs[2] on tets: toroidal push -> search_mesh (BCC) -> updatePtclPositions ->
    rebuild -> gyroScatter (tet ring map) x2, 8 steps; element ids by particle id, positions and
    scatter sums bit-exact vs the oracle.  trust: from the second step on the caller vouches for the
    origins (pp_ps_set_origin_trust: they are the destinations the previous walk accepted), the fused
    push skips check_initial_parents -- the oracle still runs it, and nothing changes."""
    pop = common.population_3d(synth, n_b=5, n_theta=20, n_planes=8, num_ptcls=4000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    common.set_shuffling(po, pg)
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg)
    for step in range(8):
        ppo.toroidal_push(po, mo, H, K, D, 6.0, trig=1)
        ids_o = ppo.search_mesh(mo, po, looplimit=200)["elem_ids"]
        ids_g = capi.DevArray(max(pg.capacity(), 1), np.int32)
        capi.push_search(mg, pg, H, K, D, 6.0, ids_g, seeded=False, looplimit=200)
        if trust:
            assert capi.push_search_counters()[2] == 0  # nobody finished as "unmoved" without a test
            pg.set_origin_trust(True)
        io, eo = common.by_id(po.member(2)[0, :po.capacity()], po.slot_info()[1], ids_o[:po.capacity()])
        ig, eg = common.by_id(pg.member(2)[0, :pg.capacity()], pg.slot_info()[1],
                              ids_g.to_host()[:pg.capacity()])
        assert np.array_equal(io, ig) and np.array_equal(eo, eg), step
        ppo.update_positions(po)
        po.rebuild(ids_o)
        pg.rebuild_commit(ids_g, 0, 1)
        assert po.nPtcls() == pg.nPtcls() > 0
        for m_o, m_g in ((fo, fg), (bo, bg)):
            wo = ppo.gyro_scatter(mo, po, m_o)
            wg = capi.gyro_scatter(mg, pg, m_g).to_host()
            assert np.array_equal(wo, wg), step
    _check_same_population(po, pg, ppo.PARTICLE_XGCM)


@pytest.mark.parametrize("gap,slow", [(3000, True), (9000, True), (9000, False)])
def test_pseudo_xgcm_steps_3d_over_full_row(ppo, synth, capi, gap, slow):
    """The same loop on a population with ONE over-full element (pseudoXGCm's remainder rule, pseudoXGCm.cpp:167-222):
    with a full sort it is the last row of the last chunk, and in the steady state of the record-fed loop the
    histogram and the first pass of the re-layout visit its own columns through their own blocks (pp_ps::hot;
    PP_NO_HOT_ROW=1 is the A/B knob).  Layout arrays, element ids by particle id, every member and the scatter
    sums equal the oracle's after every step; particles leave the element and arrive in it on the way."""
    coords, e2v, cls = synth.torus_tet(n_b=5, n_theta=20, n_planes=8)
    # (a few hundred populated elements: with fewer than 64 the chunk height shrinks, SCS_buildFns.h:3-16)
    rng = np.random.default_rng(7)
    marked = np.flatnonzero(cls <= 5)
    ppe = np.zeros(len(e2v), dtype=np.int32)
    some = rng.choice(marked, size=min(400, len(marked)), replace=False)
    ppe[some] = rng.integers(5, 40, size=len(some))
    # slow: an element of class 1 (0.01 x the angle per push: its particles trickle out); else of the fastest class
    # (the cloud crosses into the next elements within a few steps: thousands of arrivals in one row)
    have = np.flatnonzero(ppe > 0)
    pick = have[cls[have] == (cls[have].min() if slow else cls[have].max())]
    big = int(pick[len(pick) // 2])
    ppe[big] += gap
    n = int(ppe.sum())
    elem, xyz = synth.particles_in_elements(coords, e2v, ppe)
    b, phi = synth.elliptical_state(np.hypot(xyz[0], xyz[1]), xyz[2])
    pop = dict(dim=3, coords=coords, e2v=e2v, cls=cls, ppe=ppe, elem=elem,
               info=[xyz, np.zeros_like(xyz), np.arange(n, dtype=np.int32), b, phi])
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    common.set_shuffling(po, pg)
    fo, _ = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, _ = capi.create_gyro_ring_mappings(mg)
    deg = 0.5 if slow else 6.0
    hot_steps = 0
    for step in range(6):
        ppo.toroidal_push(po, mo, H, K, D, deg, trig=1)
        ids_o = ppo.search_mesh(mo, po, looplimit=200)["elem_ids"]
        ids_g = capi.DevArray(max(pg.capacity(), 1), np.int32)
        capi.push_search(mg, pg, H, K, D, deg, ids_g, seeded=False, looplimit=200)
        io, eo = common.by_id(po.member(2)[0, :po.capacity()], po.slot_info()[1], ids_o[:po.capacity()])
        ig, eg = common.by_id(pg.member(2)[0, :pg.capacity()], pg.slot_info()[1],
                              ids_g.to_host()[:pg.capacity()]) if step == 0 else (None, None)
        if step == 0:  # (reading members materialises the records: only before the steady state begins)
            assert np.array_equal(io, ig) and np.array_equal(eo, eg)
        if step == 3:  # a few of the over-full element's particles are removed on the way
            kill = np.flatnonzero(ids_o[:po.capacity()] == big)[:50]
            kill_ids = po.member(2)[0, kill]
            ids_o = ids_o.copy()
            ids_o[kill] = -1
            hg = ids_g.to_host()
            live_g = pg.slot_info()[1].astype(bool)
            pid_g = pg.member(2)[0, :pg.capacity()]
            hg[:pg.capacity()][np.isin(pid_g, kill_ids) & live_g] = -1
            ids_g = capi.DevArray.from_host(hg)
        ppo.update_positions(po)
        po.rebuild(ids_o)
        pg.rebuild_commit(ids_g, 0, 1)
        assert po.nPtcls() == pg.nPtcls() > 0
        lo, lg = po.layout(), pg.layout()
        for k in ("offsets", "slice_to_chunk", "row_to_element"):
            assert np.array_equal(lo[k], lg[k]), (step, k)
        assert np.array_equal(ppo.gyro_scatter(mo, po, fo), capi.gyro_scatter(mg, pg, fg).to_host()), step
        cnt = np.sort(np.bincount(lg["slot_elem"][lg["mask"].astype(bool)], minlength=len(e2v)))
        hot_steps += int(cnt[-1] - cnt[-2] >= 2048 + 32)  # (the next rebuild goes through the over-full row's blocks)
    assert po.slot_info()[1].sum() == pg.slot_info()[1].sum()
    if slow:  # the over-full row was there to the end
        assert hot_steps == 6
    _check_same_population(po, pg, ppo.PARTICLE_XGCM)


def _driver(name):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "pumi-pic_amd", "drivers")
    subprocess.check_call(["make", "-C", drv, "-s"])
    return os.path.join(drv, name)


@pytest.mark.parametrize("structure", [0, 1])
@pytest.mark.parametrize("strat", [1, 2, 4])
def test_cpp_driver_ps_combo160(capi, structure, strat):
    """performance_tests/ps_combo160.cpp restated on the mirror headers: USER pseudo-push lambda
    through ps::parallel_for on the 160-byte particle, then redistribute + migrate rounds; no
    particle is lost and both timing rows are printed."""
    import re
    import subprocess
    out = subprocess.run([_driver("ps_combo160"), "20000", "200000", str(strat), str(structure),
                          "-i", "5", "-s", "64"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"RESULT structure (\S+) particles (\d+) rounds (\d+) particle_rounds (\d+)", out.stdout)
    assert m, out.stdout[-2000:]
    assert m.group(1) == ("Sell-64-ne" if structure == 0 else "CSR")
    assert int(m.group(2)) == 200000 and int(m.group(4)) == 5 * 200000
    assert "pseudo-push" in out.stderr and "migrate" in out.stderr


@pytest.mark.parametrize("n,npt", [(6, 3000), (20, 1_000_000)])
def test_cpp_driver_pseudo_push_and_search(ppo, synth, capi, tmp_path, n, npt):
    """test/pseudoPushAndSearch.cpp restated on the mirror headers; the same loop run with the
    oracle gives the same survivors, wall hits and touched elements (the tet twin of the 2-D drop-in driver test:
    1 M particles on 48 000 tets, 13 s of oracle on eight threads)."""
    import re
    import subprocess
    coords, e2v, cls = synth.kuhn_box(n)
    mesh_file = str(tmp_path / "box.bin")
    synth.write_mesh_bin(mesh_file, 3, coords, e2v, cls)
    out = subprocess.run([_driver("pseudoPushAndSearch"), mesh_file, str(npt), "-0.5", "0.8", "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"RESULT particles (\d+) wall_hits (\d+) touched_elements (\d+) iterations (\d+)", out.stdout)
    assert m, out.stdout[-2000:]
    # the oracle's version of the loop
    pop = common.population_box(synth, n=n, num_ptcls=npt)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=64)
    common.set_shuffling(po)
    ppo.set_threads(ppo.max_threads())
    touched = np.zeros(len(e2v), dtype=bool)
    touched[np.unique(po.slot_info()[0][po.slot_info()[1].astype(bool)])] = True
    hits, it = 0, 1
    while it <= 30 and po.nPtcls() > 0:
        ppo.linear_push(po, 1.0 / 20, -0.5, 0.8, 0.0)
        r = ppo.search_mesh_legacy3d(mo, po, looplimit=100)
        live = po.slot_info()[1].astype(bool)
        hits += int((r["xface"][:po.capacity()][live] >= 0).sum())
        ppo.update_positions(po)
        po.rebuild(r["elem_ids"])
        if po.nPtcls() == 0:
            break
        se, mk = po.slot_info()
        touched[np.unique(se[mk.astype(bool)])] = True
        it += 1
    ppo.set_threads(1)
    assert int(m.group(1)) == po.nPtcls()
    assert int(m.group(2)) == hits
    assert int(m.group(3)) == int(touched.sum())


def test_gather_side_matches_oracle(ppo, synth, capi):
    """pp_gather_tet_vtx / pp_interp2d_field / pp_interp2d_vector / pp_interp3d_field against the
    oracle: bit-exact where only +,-,*,/,sqrt,floor are involved, 1e-12 where atan2/cos/sin are."""
    pop = common.population_3d(synth, n_b=5, n_theta=20, n_planes=8, num_ptcls=3000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    cap = po.capacity()
    rng = np.random.default_rng(11)
    field = rng.standard_normal(mo.nverts * 3)
    oo, bo = ppo.gather_tet_vtx(mo, po, field, dof=3)
    og, bg = capi.gather_tet_vtx(mg, pg, field, dof=3)
    assert bo == bg == 0 and np.array_equal(oo[:, :cap], og[:, :cap])
    # with explicit element ids (some -1)
    ids = po.slot_info()[0].copy()
    ids[::7] = -1
    oo, _ = ppo.gather_tet_vtx(mo, po, field, dof=3, elem_ids=ids)
    og, _ = capi.gather_tet_vtx(mg, pg, field, dof=3, elem_ids=capi.DevArray.from_host(ids))
    assert np.array_equal(oo[:, :cap], og[:, :cap])
    nx, nz, gx0, gz0, dx, dz = 40, 30, 0.9, -0.8, 0.04, 0.06
    data = rng.standard_normal(nx * nz * 3)
    for comp in range(3):
        a = ppo.interp2d_field(po, data, gx0, gz0, dx, dz, nx, nz, True, 3, comp)
        b = capi.interp2d_field(pg, data, gx0, gz0, dx, dz, nx, nz, True, 3, comp)
        assert np.array_equal(a[:cap], b[:cap])
    a = ppo.interp2d_vector(po, data, gx0, gz0, dx, dz, nx, nz, cyl_symm=False)
    b = capi.interp2d_vector(pg, data, gx0, gz0, dx, dz, nx, nz, cyl_symm=False)
    assert np.array_equal(a[:, :cap], b[:, :cap])
    a = ppo.interp2d_vector(po, data, gx0, gz0, dx, dz, nx, nz, cyl_symm=True)
    b = capi.interp2d_vector(pg, data, gx0, gz0, dx, dz, nx, nz, cyl_symm=True)
    np.testing.assert_allclose(a[:, :cap], b[:, :cap], rtol=1e-12, atol=1e-12)
    gx, gy, gz = np.linspace(-2.1, 2.1, 22), np.linspace(-2.1, 2.1, 18), np.linspace(-0.9, 0.9, 12)
    d = rng.standard_normal(22 * 18 * 12)
    assert np.array_equal(ppo.interp3d_field(po, gx, gy, gz, d)[:cap], capi.interp3d_field(pg, gx, gy, gz, d)[:cap])


@pytest.mark.parametrize("queue", ["0", "1"])
def test_fused_2d_outside_sentinel_and_lost_seeds(ppo, synth, capi, monkeypatch, queue):
    """search_mesh_2d conventions in the fused kernels: a seed of -nelems marks a particle that is
    already outside (hpp:1051-1056: result -1, no walk), -1 means 'start from the own element'."""
    monkeypatch.setenv("PP_WALK_QUEUE", queue)
    pop = common.population_2d(synth, num_ptcls=3000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    cap = po.capacity()
    ne = len(pop["e2v"])
    seeds = np.full(cap, -1, dtype=np.int32)
    seeds[::5] = -ne
    se = po.slot_info()[0]
    seeds[1::5] = se[1::5]  # explicit own element
    ppo.elliptical_push(po, mo, H, K, D, 4.0, trig=1)
    ids_o = seeds.copy()
    _, ids_o, _ = ppo.search_mesh_2d(mo, po, elem_ids=ids_o, looplimit=200)
    ids_g = capi.DevArray.from_host(seeds.copy())
    capi.push_search(mg, pg, H, K, D, 4.0, ids_g, seeded=True, looplimit=200)
    live = po.slot_info()[1].astype(bool)
    got = ids_g.to_host()[:cap]
    assert np.array_equal(ids_o[:cap][live], got[live])
    assert (got[live][seeds[:cap][live] == -ne] == -1).all()


def test_push_search_on_an_all_deleted_structure(ppo, synth, capi):
    """rebuild with every particle deleted, then the fused kernels on the empty structure"""
    pop = common.population_3d(synth, num_ptcls=500)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    pg.rebuild(np.full(pg.capacity(), -1, dtype=np.int32))
    assert pg.nPtcls() == 0
    ids = capi.DevArray.from_host(np.full(max(pg.capacity(), 1), 7, dtype=np.int32))
    assert capi.push_search(mg, pg, H, K, D, 1.0, ids, seeded=False, looplimit=50)


@pytest.mark.parametrize("wnormal", [False, True])
def test_closest_point_on_triangle_exact(ppo, capi, wnormal):
    """closest_point_on_triangle[_wnormal] (adjacency.hpp:824-1009): points and region codes
    bit-identical to the oracle, per-point triangles and one shared triangle."""
    rng = np.random.default_rng(5)
    n = 5000
    tris = rng.normal(size=(n, 9))
    pts = rng.normal(size=(n, 3)) * 2
    q, reg = capi.closest_point_on_triangle(tris, pts, wnormal=wnormal, reg0=-7)
    exp = [ppo.closest_point_on_triangle(tris[i], pts[i], wnormal=wnormal, reg0=-7) for i in range(n)]
    assert np.array_equal(q, np.array([e[0] for e in exp]))
    assert np.array_equal(reg, np.array([e[1] for e in exp]))
    assert set(reg) == ({0, 1, 2, 3, 4, 5, 6} if wnormal else {0, 1, 2, -7, 4, 5, 6})
    q1, reg1 = capi.closest_point_on_triangle(tris[0], pts[:500], wnormal=wnormal, reg0=-7)
    exp1 = [ppo.closest_point_on_triangle(tris[0], pts[i], wnormal=wnormal, reg0=-7) for i in range(500)]
    assert np.array_equal(q1, np.array([e[0] for e in exp1])) and np.array_equal(reg1, [e[1] for e in exp1])


@pytest.mark.parametrize("dim,mt", [(2, False), (2, True), (3, False), (3, True)])
def test_trace_stepwise_and_functor(ppo, synth, capi, dim, mt):
    """The kernel-by-kernel walk (pp_trace_*, adjacency.tpp:460-615): with the default functor it
    equals the fused pp_search_mesh and the oracle; with a user functor between find_exit_face and
    set_new_element (tpp:561-565) it equals the oracle run of the same functor."""
    pop = common.population_2d(synth, num_ptcls=3000) if dim == 2 else common.population_3d(synth, num_ptcls=3000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    (ppo.elliptical_push if dim == 2 else ppo.toroidal_push)(po, mo, H, K, D, 30.0, trig=1)
    se, mk = po.slot_info()
    cap = po.capacity()
    live = mk.astype(bool)
    xt = po.member(1)
    xt[:, :cap][:, live] = common.radial_kick(xt[:, :cap][:, live], dim, H, K)
    xg = pg.member(1)
    xg[:, :cap] = xt[:, :cap]
    pg.set_member(1, xg)
    assert np.array_equal(pg.slot_info()[1], mk)
    ref = ppo.search_mesh(mo, po, require_intersection=mt, looplimit=400)
    fused = capi.search_mesh(mg, pg, require_intersection=mt, looplimit=400)
    step = capi.trace_particle_through_mesh(mg, pg, None, require_intersection=mt, looplimit=400)
    assert step["found"] == ref["found"] == fused["found"] and step["loops"] == ref["loops"]
    for k, w in (("elem_ids", 1), ("inter_faces", 1), ("inter_points", dim)):
        if k == "elem_ids" or mt:
            a = step[k].to_host()[:cap * w]
            assert np.array_equal(a, np.asarray(ref[k]).ravel()), k
            assert np.array_equal(a, fused[k].to_host()[:cap * w]), k
    hits_o, hits_g = [], []
    wo = ppo.trace_particle_through_mesh(mo, po, common.class_interface_functor(mo, mk, hits_o),
                                         require_intersection=mt, looplimit=400)
    wg = capi.trace_particle_through_mesh(
        mg, pg, common.on_device(common.class_interface_functor(mo, mk, hits_g)),
        require_intersection=mt, looplimit=400)
    assert hits_o == hits_g and sum(hits_o) > 50
    assert wo["found"] == wg["found"] and wo["loops"] == wg["loops"]
    assert np.array_equal(wo["elem_ids"], wg["elem_ids"].to_host()[:cap])
    assert np.array_equal(wo["inter_faces"][live], wg["inter_faces"].to_host()[:cap][live])
    if mt:
        assert np.array_equal(wo["inter_points"], wg["inter_points"].to_host()[:cap * dim])


def test_trace_stepwise_loop_limit_and_seed(ppo, synth, capi):
    """seeded ids with deleted particles, origin check failures and a loop limit through the
    stepwise entry points (tpp:516-522, 72-145, 583-600)."""
    pop = common.population_3d(synth, num_ptcls=3000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    ppo.toroidal_push(po, mo, H, K, D, 25.0, trig=1)
    capi.toroidal_push(pg, mg, H, K, D, 25.0)
    se, mk = po.slot_info()
    seed = se.copy()
    seed[~mk.astype(bool)] = -1
    lv = np.flatnonzero(mk)
    seed[lv[::7]] = (seed[lv[::7]] + 11) % mo.nelems
    seed[lv[::13]] = -1
    for limit in (0, 2):
        ro = ppo.search_mesh(mo, po, elem_ids=seed.copy(), looplimit=limit)
        rg = capi.trace_particle_through_mesh(mg, pg, None, elem_ids=capi.DevArray.from_host(seed),
                                              looplimit=limit)
        assert ro["found"] == rg["found"] and ro["loops"] == rg["loops"]
        assert ro["not_in_elem"] == rg["not_in_elem"] > 0
        assert np.array_equal(ro["elem_ids"], rg["elem_ids"].to_host()[:po.capacity()])


@pytest.mark.parametrize("mt", [0, 1])
@pytest.mark.parametrize("direction", [(-0.2, 0.9, 0.15), (0.45, 0.7, -0.3)])
def test_cpp_driver_trace_wall_model(ppo, synth, capi, tmp_path, mt, direction):
    """trace_particle_through_mesh with a USER functor compiled into the application
    (drivers/traceWallModel.cpp: device lambda through ps::parallel_for at the reference's hook,
    adjacency.tpp:470-476,563) equals the oracle's run of the same wall model; the default functor
    through the same template equals the fused search_mesh."""
    import re
    import subprocess
    coords, e2v, _ = synth.kuhn_box(6)
    cy = coords[e2v][:, :, 1].mean(axis=1)
    cls = (1 + np.floor(cy * 3)).astype(np.int32)  # three material slabs along y
    mesh_file = str(tmp_path / "slabs.bin")
    synth.write_mesh_bin(mesh_file, 3, coords, e2v, cls)
    npt, dist = 2000, 0.62
    out = subprocess.run([_driver("traceWallModel"), mesh_file, str(npt), str(dist)] +
                         [repr(d) for d in direction] + [str(mt)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"RESULT found (\d+) stopped (\d+) exposed_hits (\d+) left (\d+) elem_sum (-?\d+) "
                  r"face_sum (-?\d+) default_mismatch (\d+)", out.stdout)
    assert m, out.stdout[-2000:]
    found, stopped, exposed_hits, left, esum, fsum, mismatch = (int(g) for g in m.groups())
    assert mismatch == 0
    ppe, elem, xyz = synth.push_and_search_population(coords, e2v, npt)
    xt = xyz + dist * np.array(direction)[:, None]
    pop = dict(dim=3, coords=coords, e2v=e2v, cls=cls, ppe=ppe, elem=elem,
               info=[xyz, xt, np.arange(npt, dtype=np.int32)])
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH)
    se, mk = po.slot_info()
    hits = []
    w = ppo.trace_particle_through_mesh(mo, po, common.class_interface_functor(mo, mk, hits),
                                        require_intersection=bool(mt), looplimit=200)
    live = mk.astype(bool)
    f, e = w["inter_faces"][live], w["elem_ids"][live]
    exp = mo.side_exposed[np.maximum(f, 0)].astype(bool)
    assert found == int(w["found"])
    assert stopped == int(((f >= 0) & ~exp).sum()) == sum(hits) and stopped > 100
    assert exposed_hits == int(((f >= 0) & exp).sum())
    assert left == int((e < 0).sum())
    assert esum == int(e.astype(np.int64).sum()) and fsum == int(f.astype(np.int64).sum())


@pytest.mark.parametrize("kind", ["scs", "csr"])
def test_ps_combo160_rounds_exact(ppo, capi, kind):
    """ps_combo160 rounds (performance_tests/ps_combo160.cpp:134-232): pseudo-push, redistribute
    (Distribute.h:28-89, uniform) and rebuild; the redistributed ids equal the oracle's slot for slot
    on the initial layout, and after every rebuild the per-element id sets and the pushed 160-byte
    payload agree by particle id."""
    ne, npt = 3000, 50000
    rng = np.random.default_rng(4)
    elems = np.sort(rng.integers(0, ne, size=npt).astype(np.int32))
    ppe = np.bincount(elems, minlength=ne).astype(np.int32)
    info = [np.zeros((17, npt)), np.zeros((4, npt), dtype=np.int32), np.arange(npt, dtype=np.int64)[None, :]]
    if kind == "scs":
        po = ppo.PS.scs(ppo.PERF160, ne, ppe, C_max=64, sigma=ne, V=1024, particle_elements=elems, particle_info=info)
        pg = capi.PS.scs(capi.PERF160, ne, ppe, C_=64, sigma=ne, V=1024, particle_elements=elems, particle_info=info)
    else:
        po = ppo.PS.csr(ppo.PERF160, ne, ppe, particle_elements=elems, particle_info=info)
        pg = capi.PS.csr(capi.PERF160, ne, ppe, particle_elements=elems, particle_info=info)
    parent = np.sqrt(np.arange(ne, dtype=np.float64)) * np.arange(ne)
    dparent = capi.DevArray.from_host(parent)
    a = ppo.redistribute_particles(po, 0.5, seed=3)
    b = capi.redistribute_particles(pg, 0.5, seed=3).to_host()[:pg.capacity()]
    assert np.array_equal(a, b)  # same initial layout -> same draws slot for slot
    for rnd in range(3):
        ppo.pseudo_push160(po, parent)
        capi.pseudo_push160(pg, dparent)
        # the pseudo-push writes slot-dependent values: tag the particles so the move can be followed
        for ps_, cap in ((po, po.capacity()), (pg, pg.capacity())):
            se, mk = ps_.slot_info()
            ids = ps_.member(1)[0, :cap] // 4  # nums(p,0) = 4*slot
            assert np.array_equal(ids[mk.astype(bool)], np.flatnonzero(mk))
        no = ppo.redistribute_particles(po, 0.5, seed=10 + rnd)
        ng = capi.redistribute_particles(pg, 0.5, seed=10 + rnd)
        assert np.array_equal(no != -1, po.slot_info()[1].astype(bool))
        assert np.array_equal(ng.to_host()[:pg.capacity()] != -1, pg.slot_info()[1].astype(bool))
        po.rebuild(no)
        pg.rebuild(ng)
        assert po.nPtcls() == pg.nPtcls() == npt
        # slot order inside a structure is free after a rebuild: per-element populations must have
        # the same size distribution on both sides when both started from the same layout
        if rnd == 0:
            so, mo_ = po.slot_info()
            sg, mg_ = pg.slot_info()
            assert np.array_equal(np.bincount(so[mo_.astype(bool)], minlength=ne),
                                  np.bincount(sg[mg_.astype(bool)], minlength=ne))


@pytest.mark.parametrize("kind", ["scs", "csr"])
def test_consecutive_rebuilds_move_records_to_records(ppo, capi, kind):
    """The reference's ps_combo160 rebuild loop (performance_tests/ps_combo160.cpp:205-232): rebuild after rebuild
    with NO member access in between.  For the 160-byte particle the second pass of a re-layout is deferred and the
    next rebuild's first pass reads the 192-byte records (k_move_pack_rec; pp_ps_rebuild_stats counts them).  The
    destination of a particle is a function of its current ELEMENT and the round (what both sides can evaluate from
    their own layout without touching a member), half of the elements keep their particles, one element's particles
    are deleted; after seven rounds every member of every particle equals the oracle's by particle id, and a
    pseudo-push + further rounds + new particles leave the records / member arrays consistent."""
    ne, npt = 2500, 60000
    rng = np.random.default_rng(9)
    elems = np.sort(rng.integers(0, ne, size=npt).astype(np.int32))
    ppe = np.bincount(elems, minlength=ne).astype(np.int32)
    ids = np.arange(npt, dtype=np.int64)
    info = [np.stack([ids + 0.001 * c for c in range(17)]), np.stack([4 * ids + c for c in range(4)]).astype(np.int32),
            ids[None, :].copy()]
    if kind == "scs":
        po = ppo.PS.scs(ppo.PERF160, ne, ppe, C_max=64, sigma=ne, V=1024, particle_elements=elems, particle_info=info)
        pg = capi.PS.scs(capi.PERF160, ne, ppe, C_=64, sigma=ne, V=1024, particle_elements=elems, particle_info=info)
    else:
        po = ppo.PS.csr(ppo.PERF160, ne, ppe, particle_elements=elems, particle_info=info)
        pg = capi.PS.csr(capi.PERF160, ne, ppe, particle_elements=elems, particle_info=info)
    if kind == "scs":
        common.set_shuffling(po, pg, on=False)  # (every rebuild the full re-layout: the path under test)

    def rule(ps_, rnd):
        se, mk = ps_.slot_info()
        mk = mk.astype(bool)
        se = se.astype(np.int64)
        dest = np.where((se + rnd) % 2 == 0, se, (se * 7 + 13 * rnd + 1) % ne)
        dest = np.where(se == (17 * rnd + 3) % ne, -1, dest)  # one element's particles are deleted per round
        return np.where(mk, dest, -1).astype(np.int32)

    def compare():
        capo, capg = po.capacity(), pg.capacity()
        so, mko = po.slot_info()
        sg, mkg = pg.slot_info()
        ido, idg = po.member(2)[0, :capo], pg.member(2)[0, :capg]
        io, eo = common.by_id(ido, mko, so[:capo])
        ig, eg = common.by_id(idg, mkg, sg[:capg])
        assert np.array_equal(io, ig) and np.array_equal(eo, eg)
        for m in range(3):
            _, a = common.by_id(ido, mko, po.member(m)[:, :capo])
            _, b = common.by_id(idg, mkg, pg.member(m)[:, :capg])
            assert np.array_equal(a, b), m

    before = pg.rebuild_stats()
    for rnd in range(7):
        po.rebuild(rule(po, rnd))
        pg.rebuild(rule(pg, rnd))  # (slot_info reads the layout only: no member is touched)
        assert po.nPtcls() == pg.nPtcls()
    after = pg.rebuild_stats()
    # the first read the member arrays, six read records (laboratory build with PP_NO_LAZY_UNPACK=1: pass 2 runs at
    # once, nothing is ever fed by records)
    fed = 0 if os.environ.get("PP_NO_LAZY_UNPACK") and "lab" in os.environ.get("PUMIPIC_HIP_LIB", "") else 6
    assert after[1] - before[1] == 7 and after[2] - before[2] == fed
    compare()  # (materialises the GPU's member arrays from the records)
    assert 0 < pg.nPtcls() < npt
    # ... and on: a round from the member arrays again, one from records, new particles (their round runs both passes)
    po.rebuild(rule(po, 7))
    pg.rebuild(rule(pg, 7))
    po.rebuild(rule(po, 8))
    pg.rebuild(rule(pg, 8))
    n_new = 500
    new_ids = np.arange(npt, npt + n_new, dtype=np.int64)
    new_elems = rng.integers(0, ne, size=n_new).astype(np.int32)
    new_info = [np.stack([new_ids + 0.001 * c for c in range(17)]),
                np.stack([4 * new_ids + c for c in range(4)]).astype(np.int32), new_ids[None, :].copy()]
    po.rebuild(rule(po, 9), new_elems, new_info)
    pg.rebuild(rule(pg, 9), new_elems, new_info)
    po.rebuild(rule(po, 10))
    pg.rebuild(rule(pg, 10))
    compare()
    # a pseudo-push on live records gives them up (it overwrites every member); the rounds after it agree again
    po.rebuild(rule(po, 11))
    pg.rebuild(rule(pg, 11))
    parent = np.sqrt(np.arange(ne, dtype=np.float64)) * np.arange(ne)
    ppo.pseudo_push160(po, parent)
    capi.pseudo_push160(pg, capi.DevArray.from_host(parent))
    for ps_, cap in ((po, po.capacity()), (pg, pg.capacity())):
        se, mk = ps_.slot_info()
        mk = mk.astype(bool)
        assert np.array_equal(ps_.member(2)[0, :cap][mk], np.flatnonzero(mk))  # lint(p) = p
        assert np.array_equal(ps_.member(1)[0, :cap][mk], 4 * np.flatnonzero(mk))


@pytest.mark.parametrize("strat", [2, 3, 4])
def test_redistribute_by_strategy_matches_oracle(ppo, synth, capi, strat):
    """pp_redistribute_particles_dist: the re-draw of distribute_particles' strategies (Distribute.cpp:76-253) --
    gaussian, exponential conversion, GITRm approximation -- slot for slot equal to the oracle, on the
    initial layout of a population drawn with the same strategy, SCS and CSR"""
    ne, npt = 5000, 60000
    ppe, elems = synth.distribute_particles(ne, npt, strat, seed=1)
    elems = np.sort(elems)
    info = [np.zeros((17, npt)), np.zeros((4, npt), dtype=np.int32), np.arange(npt, dtype=np.int64)[None, :]]
    po = ppo.PS.scs(ppo.PERF160, ne, ppe, C_max=64, sigma=ne, V=1024, particle_elements=elems, particle_info=info)
    pg = capi.PS.scs(capi.PERF160, ne, ppe, C_=64, sigma=ne, V=1024, particle_elements=elems, particle_info=info)
    for pm, seed in ((0.5, 3), (1.0, 4), (0.0, 5)):
        a = ppo.redistribute_particles_dist(po, strat, pm, seed=seed)
        b = capi.redistribute_particles(pg, pm, seed=seed, strat=strat).to_host()[:pg.capacity()]
        assert np.array_equal(a, b)
    pc = capi.PS.csr(capi.PERF160, ne, ppe, particle_elements=elems, particle_info=info)
    oc = ppo.PS.csr(ppo.PERF160, ne, ppe, particle_elements=elems, particle_info=info)
    assert np.array_equal(ppo.redistribute_particles_dist(oc, strat, 0.5, seed=8),
                          capi.redistribute_particles(pc, 0.5, seed=8, strat=strat).to_host()[:pc.capacity()])
    # rounds of redistribute + rebuild keep every particle and follow the distribution
    for rnd in range(3):
        ng = capi.redistribute_particles(pg, 0.5, seed=20 + rnd, strat=strat)
        pg.rebuild(ng)
        assert pg.nPtcls() == npt
    with pytest.raises(capi.PPError):
        capi.redistribute_particles(pg, 0.5, strat=0)


@pytest.mark.parametrize("cyl", [False, True])
def test_boris_push_with_gathered_fields(ppo, synth, capi, cyl):
    """pp_boris_push_fields (gather + pushBoris in one pass): E from a 3-dof vertex
    field in the particle's tet, B from an (R,Z) grid, then pumipic_push.hpp:17-75.  Against the
    oracle's composition of the three steps: bit-exact without the cylindrical rotation, 1e-12
    relative with it (atan2/cos/sin); several steps so that x, x_prev and v all feed back."""
    pop = common.population_3d(synth, n_b=5, n_theta=20, n_planes=8, num_ptcls=4000)
    members = [(np.float64, 3), (np.float64, 3), (np.float64, 3), (np.int32, 1)]
    rng = np.random.default_rng(21)
    xyz = pop["info"][0]
    vel = rng.standard_normal(xyz.shape) * 1e3
    info = [xyz, xyz - vel * 1e-9, vel, np.arange(xyz.shape[1], dtype=np.int32)]
    pop = dict(pop, info=info)
    mo, po = common.oracle_pair(ppo, pop, members)
    mg, pg = common.gpu_pair(capi, pop, members)
    cap = po.capacity()
    efield = rng.standard_normal(mo.nverts * 3) * 50.0
    nx, nz, gx0, gz0, dx, dz = 40, 30, 0.9, -0.8, 0.04, 0.06
    bgrid = rng.standard_normal(nx * nz * 3)
    d_e, d_b = capi.DevArray.from_host(efield), capi.DevArray.from_host(bgrid)
    dt = 2e-9
    live = po.slot_info()[1].astype(bool)
    for step in range(3):
        E, bad = ppo.gather_tet_vtx(mo, po, efield, dof=3)
        B = ppo.interp2d_vector(po, bgrid, gx0, gz0, dx, dz, nx, nz, cyl_symm=cyl)
        x, xp, v = (po.member(m)[:, :cap] for m in range(3))
        cols = [np.ascontiguousarray(a[i, live]) for a in (x, xp, v) for i in range(3)]
        cols += [np.ascontiguousarray(E[i, :cap][live]) for i in range(3)]
        cols += [np.ascontiguousarray(B[i, :cap][live]) for i in range(3)]
        ppo.push_boris(*cols, dt)
        for a, k in ((x, 0), (xp, 3), (v, 6)):
            for i in range(3):
                a[i, live] = cols[k + i]
        badg = capi.boris_push_fields(mg, pg, d_e, d_b, gx0, gz0, dx, dz, nx, nz, dt, cyl=cyl)
        assert bad == badg == 0
        for m in range(3):
            a, b = po.member(m)[:, :cap][:, live], pg.member(m)[:, :cap][:, live]
            if cyl:
                np.testing.assert_allclose(a, b, rtol=1e-12, atol=0)
            else:
                assert np.array_equal(a, b), (step, m)


def test_cpp_driver_pseudoxgcm_reads_gmsh(synth, capi, tmp_path):
    """the pseudoXGCm driver takes a Gmsh .msh mesh like the reference's (pseudoXGCm.cpp:306-315):
    a format-2.2 file gives the same run as the binary container of the same mesh; a 4.1 file
    (elements regrouped by class) is read and keeps every particle."""
    import re
    import subprocess
    from pumipic_amd import meshio
    c, e, cl = synth.annulus_tri(n_b=24, n_theta=96, band_width=3)
    files = {"bin": str(tmp_path / "a.bin"), "2.2": str(tmp_path / "a22.msh"), "4.1": str(tmp_path / "a41.msh")}
    synth.write_mesh_bin(files["bin"], 2, c, e, cl)
    meshio.write_gmsh(files["2.2"], 2, c, e, cl, "2.2")
    meshio.write_gmsh(files["4.1"], 2, c, e, cl, "4.1")
    res = {}
    for k, f in files.items():
        out = subprocess.run([_driver("pseudoXGCm"), f, "20000", "6", "5", "2.0", "0"],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        m = re.search(r"RESULT particles (\d+) scatter_mass (\S+) touched_elements (\d+)", out.stdout)
        assert m, out.stdout[-2000:]
        res[k] = m.groups()
        if k != "bin":
            assert "reading gmsh mesh" in out.stdout
    assert res["2.2"] == res["bin"]
    assert res["4.1"][0] == "20000"


@pytest.mark.parametrize("dim", [2, 3])
def test_gyro_scatter_gather_and_atomic_forms(ppo, synth, capi, dim):
    """the per-vertex gather over the transposed ring map (maps created by the library) and the
    atomic form (any other map pointer) give the oracle's field; gppr = 6 makes the addends
    inexact, so the gather's fixed order is checked bit for bit and the atomic form to 1e-13.
    Overwriting a library map through the API drops its transpose."""
    pop = common.population_2d(synth, num_ptcls=5000) if dim == 2 else common.population_3d(synth, num_ptcls=5000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    for gnr, gppr in ((3, 8), (4, 6)):
        fo, _ = ppo.create_gyro_ring_mappings(mo, 0.03, gnr, gppr, 10.0, trig=1)
        fg, bg = capi.create_gyro_ring_mappings(mg, 0.03, gnr, gppr, 10.0)
        assert np.array_equal(fo, fg.to_host()[:len(fo)])
        wo = ppo.gyro_scatter(mo, po, fo, 0.03, gnr, gppr)
        w_gather = capi.gyro_scatter(mg, pg, fg, 0.03, gnr, gppr).to_host()
        w_bk = capi.gyro_scatter(mg, pg, bg, 0.03, gnr, gppr).to_host()
        copy = capi.DevArray.from_host(fo)
        w_atomic = capi.gyro_scatter(mg, pg, copy, 0.03, gnr, gppr).to_host()
        assert wo.sum() > 0
        if os.environ.get("PP_SCATTER_ATOMIC") is None:  # the gather form: fixed summation order
            assert np.array_equal(wo, w_gather) and np.array_equal(wo, w_bk)
        else:
            np.testing.assert_allclose(w_gather, wo, rtol=1e-13, atol=0)
            np.testing.assert_allclose(w_bk, wo, rtol=1e-13, atol=0)
        np.testing.assert_allclose(w_atomic, wo, rtol=1e-13, atol=0)
        if gppr == 8:
            assert np.array_equal(w_atomic, wo)
        # edit the library's map through the API: vertex 0's entries point nowhere
        edited = fo.copy()
        edited[edited == 0] = -1
        fg.upload(edited)
        we = ppo.gyro_scatter(mo, po, edited, 0.03, gnr, gppr)
        wg = capi.gyro_scatter(mg, pg, fg, 0.03, gnr, gppr).to_host()
        assert we[0] == 0 and wg[0] == 0
        np.testing.assert_allclose(wg, we, rtol=1e-13, atol=0)


@pytest.mark.parametrize("dim,bridge", [(2, 0), (2, 1), (3, 0), (3, 2)])
def test_picpart_bfs_layers_exact(ppo, synth, capi, dim, bridge):
    """pp_bfs_buffer_layers / pp_bfs_safe_inward equal the oracle's restatement of
    pumipic_part_construct.cpp:387-468 for every rank of an 8-way element-block partition."""
    coords, e2v, cls = synth.annulus_tri(n_b=12, n_theta=48, band_width=3) if dim == 2 else \
        synth.torus_tet(n_b=5, n_theta=16, n_planes=8)
    mo, mg = ppo.Mesh(dim, coords, e2v, cls), capi.Mesh(dim, coords, e2v, cls)
    ne, nranks = mo.nelems, 8
    owner = (np.arange(ne, dtype=np.int64) * nranks // ne).astype(np.int32)
    d_owner = capi.DevArray.from_host(owner)
    for rank in (0, 3, 7):
        for safe_layers, ghost_layers in ((0, 0), (1, 3), (3, 2), (2, 6)):
            so, po_ = ppo.bfs_buffer_layers(mo, owner, rank, nranks, safe_layers, ghost_layers, bridge)
            sg, pg_ = capi.bfs_buffer_layers(mg, d_owner, rank, nranks, safe_layers, ghost_layers, bridge)
            assert np.array_equal(so, sg.to_host()[:ne]) and np.array_equal(po_, pg_)
            io = ppo.bfs_safe_inward(mo, owner, rank, safe_layers, po_, bridge)
            ig = capi.bfs_safe_inward(mg, d_owner, rank, nranks, safe_layers, pg_, bridge)
            assert np.array_equal(io, ig.to_host()[:ne])


@pytest.mark.parametrize("kind,commit", [("scs", True), ("scs", False), ("csr", True)])
def test_rebuild_scatter_one_call(ppo, synth, capi, kind, commit):
    """pp_ps_rebuild_scatter = pp_ps_rebuild[_commit] + pp_gyro_scatter per map: the same fields, the
    same structure and the same particle data as the separate calls and as the oracle, over several
    steps (the first rebuilds take the checked path, later ones the speculative one)."""
    pop = common.population_2d(synth, num_ptcls=6000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, kind)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, kind)
    mg2, pg2 = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, kind)
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg)
    fg2, bg2 = capi.create_gyro_ring_mappings(mg2)
    for step in range(6):
        ppo.elliptical_push(po, mo, H, K, D, 3.0, trig=1)
        _, ido, _ = ppo.search_mesh_2d(mo, po, looplimit=200)
        ids1 = capi.DevArray.from_host(np.full(pg.capacity(), -1, dtype=np.int32))
        ids2 = capi.DevArray.from_host(np.full(pg2.capacity(), -1, dtype=np.int32))
        capi.push_search(mg, pg, H, K, D, 3.0, ids1, seeded=True, looplimit=200)
        capi.push_search(mg2, pg2, H, K, D, 3.0, ids2, seeded=True, looplimit=200)
        if commit:
            ppo.update_positions(po)
        po.rebuild(ido)
        wf, wb = capi.rebuild_scatter(pg, mg, ids1, [fg, bg], commit=commit)
        if commit:
            pg2.rebuild_commit(ids2)
        else:
            pg2.rebuild(ids2)
        wf2 = capi.gyro_scatter(mg2, pg2, fg2)
        wb2 = capi.gyro_scatter(mg2, pg2, bg2)
        assert np.array_equal(wf.to_host(), wf2.to_host()) and np.array_equal(wb.to_host(), wb2.to_host())
        assert np.array_equal(wf.to_host()[:mo.nverts], ppo.gyro_scatter(mo, po, fo))
        assert pg.nPtcls() == pg2.nPtcls() == po.nPtcls() and pg.capacity() == pg2.capacity()
        assert np.array_equal(pg.slot_info()[1], pg2.slot_info()[1])
        cap = pg.capacity()
        io, xo = common.by_id(po.member(2)[0, :po.capacity()], po.slot_info()[1], po.member(0)[:, :po.capacity()])
        ig, xg = common.by_id(pg.member(2)[0, :cap], pg.slot_info()[1], pg.member(0)[:, :cap])
        assert np.array_equal(io, ig) and np.array_equal(xo, xg)
        # a plain scatter after the fused call still sees the new population
        assert np.array_equal(capi.gyro_scatter(mg, pg, fg).to_host(), wf.to_host())


@pytest.mark.parametrize("shuffle", [False, True])
def test_rebuild_speculative_and_checked_paths_alternate(ppo, synth, capi, shuffle):
    """The rebuild enqueues its tail speculatively when the buffers have room and falls back to the
    checked path otherwise (DESIGN "The host sync").  Sequence: steady rebuilds (speculation holds) ->
    a burst of new particles that outgrows every buffer (speculation fails after it was enqueued) ->
    steady again -> every particle into one element (the layout sort needs more radix passes than
    the previous rebuild predicted) -> almost everything deleted so that fewer than C elements hold particles (chunk
    height changes: fails) -> an invalid id (error, the structure stays usable) -> steady.  After every
    step the population equals the oracle's by particle id."""
    pop = common.population_2d(synth, num_ptcls=3000)
    ne = len(pop["e2v"])
    po = ppo.PS.scs(ppo.PARTICLE_XGCM, ne, pop["ppe"], C_max=64, particle_elements=pop["elem"],
                    particle_info=pop["info"])
    pg = capi.PS.scs(capi.PARTICLE_XGCM, ne, pop["ppe"], C_=64, particle_elements=pop["elem"],
                     particle_info=pop["info"])
    common.set_shuffling(po, pg, on=shuffle)
    rng = np.random.default_rng(17)
    next_id = 3000

    def step(move_frac, n_new, keep=None, all_to=None):
        nonlocal next_id
        dec = rng.integers(0, ne, size=next_id).astype(np.int32)
        if all_to is not None:
            dec[:] = all_to
        mv = rng.random(next_id) < move_frac
        add_e = rng.integers(0, ne, size=n_new).astype(np.int32)
        add = None
        if n_new:
            add = [rng.random((3, n_new)), rng.random((3, n_new)), np.arange(next_id, next_id + n_new, dtype=np.int32),
                   rng.random(n_new).astype(np.float32), rng.random(n_new).astype(np.float32)]
        outs = []
        for ps_ in (po, pg):
            se, mk = ps_.slot_info()
            ids = ps_.member(2)[0, :ps_.capacity()]
            new = np.full(len(se), -1, dtype=np.int32)
            live = mk.astype(bool)
            i = ids[live]
            e = np.where(mv[i], dec[i], se[live])
            if keep is not None:
                e = np.where(np.isin(i, keep), e, -1)
            new[live] = e
            outs.append(new)
        po.rebuild(outs[0], add_e if n_new else None, add)
        pg.rebuild(outs[1], add_e if n_new else None, add)
        next_id += n_new
        assert po.nPtcls() == pg.nPtcls()
        _check_same_population(po, pg, ppo.PARTICLE_XGCM)
        lo, lg = po.layout(), pg.layout()  # same reshuffle-or-rebuild decision => same layout arrays
        for k in ("C", "num_chunks", "num_slices", "capacity", "num_rows"):
            assert lo[k] == lg[k], (k, lo[k], lg[k])
        for k in ("offsets", "slice_to_chunk", "row_to_element", "element_to_row"):
            assert np.array_equal(lo[k], lg[k]), k
        assert bool(po.s.last_rebuild_was_shuffle) == (pg.rebuild_stats()[0] > stats[0])
        stats[0] = pg.rebuild_stats()[0]

    stats = [0]
    for _ in range(3):
        step(0.3, 0)
    step(0.3, 90000)           # outgrows mask / slot / staging / swap buffers
    for _ in range(2):
        step(0.3, 0)
    step(1.0, 0, all_to=7)     # one element takes everything: the sort key outgrows the predicted passes
    step(1.0, 0)               # and spreads out again
    step(0.3, 0)
    keep = np.arange(0, next_id, next_id // 30)   # ~30 particles left: fewer than 64 non-empty elements
    step(0.0, 0, keep=keep)
    step(0.5, 0)
    bad = np.full(max(pg.capacity(), 1), -1, dtype=np.int32)
    se, mk = pg.slot_info()
    bad[:len(se)] = np.where(mk.astype(bool), se, -1)
    bad[np.flatnonzero(mk)[0]] = ne + 5
    with pytest.raises(capi.PPError):
        pg.rebuild(bad)
    step(0.5, 2000)
    for _ in range(2):
        step(0.2, 0)


@pytest.mark.parametrize("dim,kind", [(2, "scs"), (3, "scs"), (2, "csr")])
def test_gyro_scatter_per_particle_radius(ppo, synth, capi, dim, kind):
    """pp_gyro_scatter_radius: the reference's TODO radius (gyroScatter.hpp:184) per particle + a weight.
    With the reference's constant radius and weight 1 it reproduces pp_gyro_scatter bit for bit; with
    random radii / weights it agrees with the oracle's particle-by-particle loop to 1e-12 (the sums
    are no longer exact integers: one atomic per (element, ring) per row run, then a vertex gather)."""
    pop = common.population_2d(synth, num_ptcls=6000) if dim == 2 else common.population_3d(synth, num_ptcls=6000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, kind)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, kind)
    rmax, gnr, gppr = 0.038, 4, 8
    fo, _ = ppo.create_gyro_ring_mappings(mo, rmax, gnr, gppr, 0.0, trig=1)
    fg, _ = capi.create_gyro_ring_mappings(mg, rmax, gnr, gppr, 0.0)
    capo, capg = po.capacity(), max(pg.capacity(), 1)
    ids_o, ids_g = po.member(2)[0, :capo], pg.member(2)[0, :capg]
    rng = np.random.default_rng(3)
    npt = 6000
    # keyed by particle id so both layouts see the same per-particle values
    rad_by_id = rng.uniform(0.5 * rmax / gnr, 1.2 * rmax, npt)       # includes radii past the last ring
    w_by_id = rng.uniform(0.25, 4.0, npt)
    live_o, live_g = po.slot_info()[1].astype(bool), pg.slot_info()[1].astype(bool)
    rad_o, w_o = np.zeros(capo), np.zeros(capo)
    rad_o[live_o], w_o[live_o] = rad_by_id[ids_o[live_o]], w_by_id[ids_o[live_o]]
    rad_g, w_g = np.zeros(capg), np.zeros(capg)
    lg = np.flatnonzero(live_g)
    rad_g[lg], w_g[lg] = rad_by_id[ids_g[lg]], w_by_id[ids_g[lg]]
    ref, clip_o = ppo.gyro_scatter_radius(mo, po, rad_o, fo, w_o, rmax, gnr, gppr)
    got, clip_g = capi.gyro_scatter_radius(mg, pg, capi.DevArray.from_host(rad_g), fg, capi.DevArray.from_host(w_g),
                                           rmax, gnr, gppr)
    got = got.to_host()[:mo.nverts]
    assert clip_o == clip_g and clip_o > 0
    scale = np.abs(ref).max()
    assert scale > 0 and np.abs(got - ref).max() <= 1e-12 * scale
    # the reference's constant radius, weight 1: exact integers / eighths in any order
    const = np.full(capg, (rmax / gnr) * 1.125)
    a, clip = capi.gyro_scatter_radius(mg, pg, capi.DevArray.from_host(const), fg, None, rmax, gnr, gppr)
    assert clip == 0
    assert np.array_equal(a.to_host()[:mo.nverts], capi.gyro_scatter(mg, pg, fg, rmax, gnr, gppr).to_host()[:mo.nverts])
    assert np.array_equal(a.to_host()[:mo.nverts], ppo.gyro_scatter(mo, po, fo, rmax, gnr, gppr))


@pytest.mark.parametrize("segment", [False, True])
def test_ray_and_segment_triangle_match_oracle(ppo, capi, segment):
    """pp_ray_intersects_triangle: ray (adjacency.tpp:152-178) and segment (tpp:192-201) forms, bit for bit
    against the oracle on random triangles, both orientations, rays that stop short of / pass through
    the face (test/moller_trumbore_line_tri_test.cpp:51-162 distinguishes exactly those)."""
    rng = np.random.default_rng(12)
    n = 4000
    tris = rng.normal(size=(n, 9))
    # aim at a point of the triangle (or near it), stop before or beyond it
    bary = rng.dirichlet([1, 1, 1], size=n) + rng.normal(scale=0.2, size=(n, 3)) * (rng.random((n, 1)) < 0.3)
    target = (tris.reshape(n, 3, 3) * bary[:, :, None]).sum(axis=1)
    orig = target + rng.normal(size=(n, 3))
    dest = orig + (target - orig) * rng.uniform(0.3, 1.8, size=(n, 1))
    flip = rng.integers(0, 2, size=n)
    hit, xp, o3 = capi.ray_intersects_triangle(tris, orig, dest, 1e-8, flip, segment)
    nh = 0
    for i in range(n):
        h, x, dproj, close, par = ppo.ray_triangle(tris[i], orig[i], dest[i], 1e-8, int(flip[i]), segment)
        assert h == hit[i], i
        assert np.array_equal(x, xp[i]) and (dproj, close, par) == tuple(o3[i]), i
        nh += h
    assert 0.05 * n < nh < 0.95 * n
    if segment:  # the segment form rejects hits beyond the destination that the ray form accepts
        hit_ray, _, _ = capi.ray_intersects_triangle(tris, orig, dest, 1e-8, flip, False)
        assert (hit_ray & ~hit).sum() > 0 and not (hit & ~hit_ray).any()


def test_wall_points_off_their_face_are_the_references_fallback(ppo, synth, capi):
    """test/test_adj.cpp:640-652 checks that a recorded wall intersection lies INSIDE its face.  On fine meshes one or two
    rays per 10^6 fail that check in the reference's own algorithm: when no face of an element passes the
    Moeller-Trumbore test (the ray leaves through an edge within the tolerance), adjacency.tpp:343-352 falls back to
    the face with the best `closeness` and keeps THAT face's intersection point -- of a face the ray does not cross.
    Here: 10^6 rays on 105 456 tets; the wall hits whose point is off their face are few, and for exactly those
    particles (plus a sample of ordinary ones) the oracle's restatement of the reference returns the same element,
    face and point bit for bit -- the library reproduces the reference, fallback included."""
    n, npt = 26, 2_000_000
    coords, e2v, cls = synth.kuhn_box(n)
    ne = len(e2v)
    ppe = np.full(ne, npt // ne, dtype=np.int32)
    ppe[:npt % ne] += 1
    elem = np.repeat(np.arange(ne, dtype=np.int32), ppe)
    rng = np.random.default_rng(5)
    # uniform in the element WITHOUT a margin (test_adj.cpp:440-505): some particles start within 1e-9 of a face
    u = np.sort(rng.random((npt, 3)), axis=1)
    wgt = np.stack([u[:, 0], u[:, 1] - u[:, 0], u[:, 2] - u[:, 1], 1 - u[:, 2]], axis=1)
    xyz = np.einsum("nj,njk->nk", wgt, coords[e2v[elem]]).T.copy()
    d = rng.standard_normal((3, npt))
    d /= np.linalg.norm(d, axis=0)
    # half of the particles as test_adj.cpp:283-437 places them: ON a vertex of their element, heading for another
    # vertex of it or for a point of one of its edges -- rays along edges and through vertices
    deg = np.arange(npt) % 2 == 1
    tv = e2v[elem[deg]]
    k0 = rng.integers(0, 4, deg.sum())
    k1 = (k0 + rng.integers(1, 4, deg.sum())) % 4
    k2 = (k1 + rng.integers(1, 4, deg.sum())) % 4
    p0 = coords[tv[np.arange(len(tv)), k0]]
    w = np.where(rng.random(deg.sum()) < 0.5, 0.0, rng.random(deg.sum()))[:, None]  # a vertex, or along an edge from it
    goal = coords[tv[np.arange(len(tv)), k1]] * (1 - w) + coords[tv[np.arange(len(tv)), k2]] * w
    dd = goal - p0
    nz = np.linalg.norm(dd, axis=1) > 0
    dd[nz] /= np.linalg.norm(dd[nz], axis=1)[:, None]
    xyz[:, deg] = p0.T
    d[:, deg] = dd.T
    tgt = xyz + d * (10.0 / (3.0 * ne ** (1.0 / 3.0)))  # ten pushes of test_adj.cpp:551's distance
    ids = np.arange(npt, dtype=np.int32)
    step = 10.0 / (3.0 * ne ** (1.0 / 3.0))
    for stage in (1, 2):
        if stage == 2:
            # test_adj.cpp:801-812, 857-870: the particles that hit the wall are put ON it, turned round and pushed again
            keep = np.flatnonzero(mask & (xf >= 0) & np.isfinite(xp).all(axis=1))
            sel = pid[keep]
            o2 = np.argsort(el[keep], kind="stable")
            sel, keep = sel[o2], keep[o2]
            elem, xyz, d = el[keep].astype(np.int32), xp[keep].T.copy(), -d[:, sel]
            ids = np.arange(len(sel), dtype=np.int32)
            npt = len(sel)
            ppe = np.bincount(elem, minlength=ne).astype(np.int32)
            tgt = xyz + d * step
        _stage(ppo, capi, coords, e2v, cls, ne, npt, ppe, elem, xyz, tgt, ids, rng, stage)
        if stage == 1:
            mask, pid, xf, xp, el = _stage.last


def _stage(ppo, capi, coords, e2v, cls, ne, npt, ppe, elem, xyz, tgt, ids, rng, stage):
    pop = dict(dim=3, coords=coords, e2v=e2v, cls=cls, ppe=ppe, elem=elem, info=[xyz, tgt, ids])
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_PUSH)
    rg = capi.search_mesh(mg, pg, require_intersection=True)
    assert rg["found"]
    cap = pg.capacity()
    mask = pg.slot_info()[1].astype(bool)
    pid = pg.member(2)[0, :cap]
    xf = rg["inter_faces"].to_host()[:cap]
    xp = rg["inter_points"].to_host()[:cap * 3].reshape(cap, 3)
    el = rg["elem_ids"].to_host()[:cap]
    hit = np.flatnonzero(mask & (xf >= 0))
    assert len(hit) > 0.3 * npt  # most of these rays reach the wall
    _stage.last = (mask, pid, xf, xp, el)
    s2v = mg.array(capi.MESH_SIDE2VERTS).reshape(-1, 3)
    a, b, c = (coords[s2v[xf[hit], k]] for k in range(3))
    nrm = np.cross(b - a, c - a)
    area2 = np.einsum("ij,ij->i", nrm, nrm)
    with np.errstate(invalid="ignore", divide="ignore"):
        bc = np.stack([np.einsum("ij,ij->i", nrm, np.cross(b - a, xp[hit] - a)),
                       np.einsum("ij,ij->i", nrm, np.cross(c - b, xp[hit] - b)),
                       np.einsum("ij,ij->i", nrm, np.cross(xp[hit] - a, c - a))]) / area2
    off = hit[~(np.isfinite(bc).all(axis=0) & (bc >= -1e-8).all(axis=0))]
    assert len(off) <= 20, len(off)  # a handful per million, not a population
    print("stage %d: wall hits %d, off their face %d" % (stage, len(hit), len(off)))
    # the oracle on exactly those particles + 3000 ordinary ones
    take = np.unique(np.concatenate([pid[off], rng.choice(npt, min(3000, npt), replace=False)]))
    order = np.argsort(elem[take], kind="stable")
    take = take[order]
    ppe2 = np.bincount(elem[take], minlength=ne).astype(np.int32)
    pop2 = dict(dim=3, coords=coords, e2v=e2v, cls=cls, ppe=ppe2, elem=elem[take],
                info=[xyz[:, take], tgt[:, take], ids[take]])
    mo, po = common.oracle_pair(ppo, pop2, ppo.PARTICLE_PUSH)
    ro = ppo.search_mesh(mo, po, require_intersection=True)
    ocap = po.capacity()
    omask = po.slot_info()[1].astype(bool)
    opid = po.member(2)[0, :ocap][omask]
    oel, oxf = ro["elem_ids"][:ocap][omask], ro["inter_faces"][:ocap][omask]
    oxp = ro["inter_points"][:ocap * 3].reshape(ocap, 3)[omask]
    slot_of = np.full(npt, -1, dtype=np.int64)
    slot_of[pid[mask]] = np.flatnonzero(mask)
    gs = slot_of[opid]
    assert (gs >= 0).all()
    assert np.array_equal(oel, el[gs]) and np.array_equal(oxf, xf[gs])
    assert np.array_equal(oxp.view(np.uint64), xp[gs].view(np.uint64))  # bit for bit, NaN included
