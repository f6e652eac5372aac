"""Pins the CPU oracle against every reference KAT that survives the missing pumipic-data
submodule.  CPU only."""
import numpy as np
import pytest


# ---------------------------------------------------------------- src/unit_tests.hpp:101-177
def test_barycentric1_unit_vectors(ppo):
    M = np.array([[0.0, 1.0, 0.0], [0.5, 0.0, 0.0], [1.0, 1.0, 0.0], [0.5, 1.0, 0.5]])
    opp = [3, 2, 0, 1]  # simplex_opposite_template(3,2,i)
    for i in range(4):
        _, old, coords = ppo.barycentric_tet(M, M[opp[i]], 1.0)
        expect = np.zeros(4)
        expect[i] = 1.0
        assert np.abs(old - expect).max() <= 1e-10
        assert np.abs(coords - expect).max() <= 1e-10
    # p2 must NOT have bcc[2]==1 (intended-to-fail case, unit_tests.hpp:137-139)
    _, old, _ = ppo.barycentric_tet(M, M[1], 1.0)
    assert abs(old[2] - 1.0) > 1e-10


def test_barycentric2_values(ppo):
    M = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.5, 0.25, 1.0]])
    _, old, coords = ppo.barycentric_tet(M, [0.2, 0.1, 0.1], 1.0)
    assert np.abs(old - [0.1, 0.15, 0.675, 0.075]).max() <= 1e-10
    assert np.abs(coords - [0.1, 0.15, 0.675, 0.075]).max() <= 1e-10
    for p, pos, val in [([1.5, 0.1, 0.1], 0, 0.1), ([0.1, 0.2, 0.1], 1, 0.35),
                        ([0.1, -0.2, 0.1], 2, 1.075), ([0.1, -0.2, -0.1], 3, 0.325)]:
        _, old, _ = ppo.barycentric_tet(M, p, 1.0)
        assert abs(old[pos] - val) <= 1e-10 * max(1.0, abs(val))


def test_barycentric_tet_new_sums_to_six(ppo):
    """tpp barycentric_tet fed the true volume returns 6x barycentrics."""
    M = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.5, 0.25, 1.0]])
    b0, b1, b2 = M[1] - M[0], M[2] - M[0], M[3] - M[0]
    vol = np.dot(np.cross(b0, b1), b2) / 6.0
    new, old, _ = ppo.barycentric_tet(M, [0.2, 0.1, 0.1], vol)
    assert abs(new.sum() - 6.0) < 1e-12
    assert np.allclose(new / 6.0, old, rtol=0, atol=1e-14)


def test_helpers(ppo):
    assert ppo.all_positive([1e-11, -1e-11, 0.5])
    assert not ppo.all_positive([1e-11, -2e-10, 0.5])
    assert not ppo.all_positive([np.nan, 1.0, 1.0])
    assert ppo.min3([0.2, 0.2, 0.3]) == 1          # (a0<a1)?0:1 -> 1 on ties
    assert ppo.min3([0.2, 0.3, 0.2]) == 2          # (a[idx]<a2)?idx:2 -> 2 on ties
    assert ppo.min3([0.1, 0.3, 0.2]) == 0
    assert ppo.min_index([0.2, 0.2, 0.1, 0.1]) == 2  # first minimum wins
    assert ppo.max_index([0.2, 0.3, 0.3, 0.1]) == 1


def test_barycentric_tri_edge_major(ppo):
    fc = [[0, 0], [1, 0], [0, 1]]
    b = ppo.barycentric_tri(fc, [0.25, 0.25], 0.5)
    # edge 0 = (v0,v1) -> coordinate of the opposite vertex v2, etc.
    assert np.allclose(b, [0.25, 0.5, 0.25])
    assert abs(b.sum() - 1) < 1e-15


# ---------------------------------------------------------------- shared sincos
def test_sincos_within_one_ulp_of_libm(ppo):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-4, 4, 20000), rng.uniform(-1e3, 1e3, 20000),
                         rng.uniform(-1e6, 1e6, 5000), [0.0, np.pi / 4, -np.pi / 4, np.pi / 2,
                                                         np.pi, 1e-300, 5e-324]])
    worst = 0.0
    for x in xs:
        s, c = ppo.sincos(x)
        for got, ref in ((s, np.sin(x)), (c, np.cos(x))):
            u = np.spacing(abs(ref)) if ref != 0 else 5e-324
            worst = max(worst, abs(got - ref) / u)
    assert worst <= 1.0, worst
    s, c = ppo.sincos(float("nan"))
    assert np.isnan(s) and np.isnan(c)
    s, c = ppo.sincos(float("inf"))
    assert np.isnan(s) and np.isnan(c)


# ---------------------------------------------------------------- test/search2d.cpp:205-308
TRI8_CASES = [
    # (parent, start, end, dest, alt)
    (5, (.60, .80), (.60, .99), 5, None),
    (5, (.60, .80), (.940, .950), 5, None),
    (5, (.60, .80), (.510, .91), 5, None),
    (0, (.40, .20), (.495, .470), 0, None),
    (0, (.40, .20), (.110, .1), 0, None),
    (0, (.40, .20), (.40, .010), 0, None),
    (5, (.60, .80), (.40, .730), 1, None),
    (0, (.50, .50), (.80, .80), 3, 5),
    (0, (.50, .50), (.80, 0.0), 7, None),
    (0, (.250, .250), (.40, .40), 0, 2),
    (6, (.750, .250), (.750, .60), 3, None),
    (5, (.80, .80), (.40, .40), 0, 2),
    (6, (.750, .250), (.40, .60), 1, None),
    (6, (.60, .40), (.20, .80), 4, None),
]


@pytest.fixture(scope="module")
def tri8(ppo, synth):
    c, e, cl = synth.plate_tri8_pardiag()
    return ppo.Mesh(2, c, e, cl)


def _one_particle_ps(ppo, members, ne, parent, start, end, C_max=1):
    ppe = np.zeros(ne, dtype=np.int32)
    ppe[parent] = 1
    x = np.zeros((3, 1))
    x[:len(start), 0] = start
    xt = np.zeros((3, 1))
    xt[:len(end), 0] = end
    info = [x, xt, np.zeros(1, dtype=np.int32)] + [np.zeros(1, np.float32)] * (len(members) - 3)
    return ppo.PS.scs(members, ne, ppe, C_max=C_max, particle_elements=[parent],
                      particle_info=info)


@pytest.mark.parametrize("case", TRI8_CASES)
def test_search2d_tri8_cases(ppo, tri8, case):
    parent, start, end, dest, alt = case
    ps = _one_particle_ps(ppo, ppo.PARTICLE_PUSH, 8, parent, start, end)
    found, elem_ids, loops = ppo.search_mesh_2d(tri8, ps, looplimit=100)
    _, mask = ps.slot_info()
    got = elem_ids[mask.astype(bool)]
    assert found and len(got) == 1
    assert got[0] == dest or (alt is not None and got[0] == alt), (got, dest, alt)
    # the new (tpp) BCC search must agree with search_mesh_2d
    r = ppo.search_mesh(tri8, ps, require_intersection=False, looplimit=100)
    assert r["found"] and r["elem_ids"][mask.astype(bool)][0] == got[0]
    # and the single-point walker
    e, _ = ppo.search_mesh_2d_pt(tri8, start, end, parent, looplimit=100)
    assert e == got[0]


def test_search2d_leaves_domain(ppo, tri8):
    ps = _one_particle_ps(ppo, ppo.PARTICLE_PUSH, 8, 5, (.6, .8), (.6, 1.5))
    found, elem_ids, _ = ppo.search_mesh_2d(tri8, ps, looplimit=100)
    _, mask = ps.slot_info()
    assert found and elem_ids[mask.astype(bool)][0] == -1
    # intersection mode records the wall edge and the hit point (tpp:378-380)
    r = ppo.search_mesh(tri8, ps, require_intersection=True, looplimit=100)
    i = np.flatnonzero(mask)[0]
    assert r["found"] and r["inter_faces"][i] >= 0
    assert tri8.side_exposed[r["inter_faces"][i]] == 1
    assert np.allclose(r["inter_points"][i], [.6, 1.0], atol=1e-12)
    assert r["elem_ids"][i] == 5  # stays in the last element when the wall is hit


# ---------------------------------------------------------------- test/pseudoXGCm_scatter.cpp
def test_gyro_scatter_kat(ppo, tri8):
    rmax, gnr, gppr, theta = .2, 2, 6, 15
    fwd, bkwd = ppo.create_gyro_ring_mappings(tri8, rmax, gnr, gppr, theta)

    def modify(inmap):  # pseudoXGCm_scatter.cpp:58-81
        m = inmap.reshape(tri8.nverts, gnr * gppr * 3).copy()
        keep = m[3].copy()
        m[:] = 2
        m[3] = keep
        return m.reshape(-1)

    ppe = np.zeros(8, dtype=np.int32)
    ppe[0] = 1
    ps = ppo.PS.scs(ppo.PARTICLE_XGCM, 8, ppe, C_max=1, V=32)
    w = ppo.gyro_scatter(tri8, ps, modify(fwd), rmax, gnr, gppr)
    wb = ppo.gyro_scatter(tri8, ps, modify(bkwd), rmax, gnr, gppr)
    for i in range(tri8.nverts):
        expect = {3: 2.0, 2: 12.0, 8: 0.0}.get(i, 2.0 / 3.0)
        assert abs(w[i] - expect) <= 1e-10 * max(1.0, expect), (i, w[i], expect)
    assert np.array_equal(w, wb)


# ---------------------------------------------------------------- moller_trumbore_line_tri_test
def test_ray_vs_segment_semantics(ppo, synth):
    """test/moller_trumbore_line_tri_test.cpp:51-162 on an own 6-tet cube: a path that ends
    INSIDE a tet hits no face as a segment, and exactly one face as a ray."""
    c, e, cl = synth.kuhn_box(1, lo=(-.5, -.5, -.5), hi=(.5, .5, 1.0))
    m = ppo.Mesh(3, c, e, cl)
    tol = m.tolerance()
    o = np.array([0.0, -0.2, -0.45])
    z = np.array([0.0, -0.2, 0.9])
    # which tet holds z?
    holder = None
    for t in range(m.nelems):
        new, _, _ = ppo.barycentric_tet(m.coords[m.elem2verts[t]], z, m.elem_measure[t])
        if ppo.all_positive(new, 1e-10):
            holder = t
    assert holder is not None
    tv = m.elem2verts[holder]
    seg, ray, zs = [], [], []
    for fi in range(4):
        fid = m.elem2sides[holder, fi]
        fv = m.side2verts[fid]
        i1, i2 = [2, 1, 1, 3, 2, 3, 0, 3][2 * fi:2 * fi + 2]
        idx = 1 if fv[0] == tv[i1] else (2 if fv[1] == tv[i1] else 0)
        flip = tv[i2] != fv[idx]
        hs, _, _, _, _ = ppo.ray_triangle(m.coords[fv], o, z, tol, flip, segment=True)
        hr, xp, _, _, par = ppo.ray_triangle(m.coords[fv], o, z, tol, flip, segment=False)
        seg.append(hs)
        ray.append(hr)
        if hr:
            zs.append((xp[2], par))
    assert seg == [False] * 4
    assert sum(ray) == 1
    assert abs(zs[0][0] - 1.0) < tol and zs[0][1] > 1.0
