"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py with the CPU
oracle): the CPU half pins the oracle against silent drift, the GPU half checks the HIP path
against the same files without the oracle in the loop."""
import os

import numpy as np
import pytest

import common

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H, K, D = 1.72479370 - .08, .020558260, 0.6


def _load(name):
    return np.load(os.path.join(GOLD, name))


def _scs_cases():
    g = _load("scs_layouts.npz")
    keys = sorted(k[:-4] for k in g.files if k.endswith("_ppe"))
    return g, keys


def _parse(key):
    p, C, V, s, pad = key.split("_")
    sigma = int(s[1:])
    return int(C[1:]), int(V[1:]), (2**31 - 1 if sigma == 99 else sigma), int(pad[3:])


# ---------------------------------------------------------------- CPU: oracle == golden
def test_oracle_scs_layouts_golden(ppo):
    g, keys = _scs_cases()
    assert len(keys) == 12
    for key in keys:
        C, V, sigma, pad = _parse(key)
        ppe = g[key + "_ppe"]
        ps = ppo.PS.scs([(np.int32, 1)], len(ppe), ppe, C_max=C, sigma=sigma, V=V, pad_strat=pad)
        L = ps.layout()
        for k in ("offsets", "slice_to_chunk", "row_to_element", "mask"):
            assert np.array_equal(L[k], g[key + "_" + k]), (key, k)
        assert [L["C"], L["capacity"]] == list(g[key + "_C"])


def test_oracle_xgcm2d_golden(ppo, synth):
    g = _load("xgcm2d_24x96_1000p_30steps.npz")
    pop = common.population_2d(synth, n_b=24, n_theta=96, num_ptcls=1000, mdl_face=6, band_width=3)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=32)  # golden was made with C=1
    fwd, _ = ppo.create_gyro_ring_mappings(mesh, trig=1)
    assert np.array_equal(fwd, g["fwd_map"])
    for step in range(30):
        ppo.elliptical_push(ps, mesh, H, K, D, 2.0, trig=1)
        _, ids, _ = ppo.search_mesh_2d(mesh, ps, looplimit=200)
        cap = ps.capacity()
        _, e = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], ids[:cap])
        assert np.array_equal(e, g["elem_ids"][step]), step
        ppo.update_positions(ps)
        ps.rebuild(ids)
    cap = ps.capacity()
    _, x = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], ps.member(0)[:, :cap])
    assert np.array_equal(x, g["final_x"])
    w = ppo.gyro_scatter(mesh, ps, fwd)
    np.testing.assert_allclose(w, g["scatter_fwd"], rtol=1e-12, atol=1e-12)


def test_oracle_xgcm3d_golden(ppo, synth):
    g = _load("xgcm3d_6x24x8_1000p_12steps.npz")
    pop = common.population_3d(synth, n_b=6, n_theta=24, n_planes=8, num_ptcls=1000, mdl_face=5)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=64)
    ids = None
    cap = ps.capacity()
    for step in range(12):
        ppo.toroidal_push(ps, mesh, H, K, D, 6.0, trig=1)
        ids = ppo.search_mesh(mesh, ps, elem_ids=ids, looplimit=200)["elem_ids"]
        _, e = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], ids[:cap])
        assert np.array_equal(e, g["elem_ids"][step]), step
        a, b = ps.member(0), ps.member(1)
        tmp = a.copy()
        a[:] = b
        b[:] = tmp
    _, x = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], ps.member(0)[:, :cap])
    assert np.array_equal(x, g["final_x"])


def test_oracle_pushsearch_golden(ppo, synth):
    g = _load("pushsearch_box6_1000p.npz")
    pop = common.population_box(synth, n=6, num_ptcls=1000)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=16)
    for step in range(len(g["elem_ids"])):
        ppo.linear_push(ps, 1.0 / 20, -0.5, 0.8, 0.0)
        r = ppo.search_mesh_legacy3d(mesh, ps, looplimit=100)
        cap = ps.capacity()
        pid, e = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], r["elem_ids"][:cap])
        _, f = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], r["xface"][:cap])
        assert np.array_equal(e, g["elem_ids"][step][pid]) and np.array_equal(f, g["xface"][step][pid])
        assert np.all(np.delete(g["elem_ids"][step], pid) == -2)  # the rest were deleted earlier
        ppo.update_positions(ps)
        ps.rebuild(r["elem_ids"])


def _walls_and_parts(api, synth, on_gpu):
    """the checks of walls_and_parts.npz, shared by the oracle (CPU) and the HIP path (GPU)"""
    g = _load("walls_and_parts.npz")
    host = (lambda a: a.to_host()) if on_gpu else (lambda a: np.asarray(a))
    pair = common.gpu_pair if on_gpu else common.oracle_pair
    pop = common.population_box(synth, n=5, num_ptcls=800)
    mesh, ps = pair(api, pop, api.PARTICLE_PUSH)
    api.linear_push(ps, 0.55, -0.5, 0.8, 0.15)
    r = api.search_mesh_3d(mesh, ps, looplimit=200)
    cap = ps.capacity()
    ids, mk = ps.member(2)[0, :cap], ps.slot_info()[1]
    pid, e = common.by_id(ids, mk, host(r["elem_ids"])[:cap])
    _, f = common.by_id(ids, mk, host(r["xface"])[:cap])
    _, xp = common.by_id(ids, mk, host(r["xpoints"]).reshape(-1, 3)[:cap].T)
    assert np.array_equal(pid, g["s3d_pid"]) and np.array_equal(e, g["s3d_elem"])
    assert np.array_equal(f, g["s3d_xface"]) and np.array_equal(xp[:, f >= 0], g["s3d_xpoints"][:, f >= 0])
    coords, e2v, _ = synth.kuhn_box(5)
    cls = (1 + np.floor(coords[e2v][:, :, 1].mean(axis=1) * 3)).astype(np.int32)
    return g, dict(pop, cls=cls), pair, host


def test_oracle_walls_and_parts_golden(ppo, synth):
    g, pop2, pair, host = _walls_and_parts(ppo, synth, False)
    mesh2, ps2 = pair(ppo, pop2, ppo.PARTICLE_PUSH)
    ppo.linear_push(ps2, 0.55, -0.5, 0.8, 0.15)
    cap = ps2.capacity()
    ids, mk = ps2.member(2)[0, :cap], ps2.slot_info()[1]
    for mt in (0, 1):
        w = ppo.trace_particle_through_mesh(mesh2, ps2, common.class_interface_functor(mesh2, mk, []),
                                            require_intersection=bool(mt), looplimit=200)
        assert np.array_equal(common.by_id(ids, mk, w["elem_ids"][:cap])[1], g["wall%d_elem" % mt])
        assert np.array_equal(common.by_id(ids, mk, w["inter_faces"][:cap])[1], g["wall%d_face" % mt])
    for wn in (0, 1):
        res = [ppo.closest_point_on_triangle(g["cp_tris"][i], g["cp_pts"][i], wnormal=bool(wn), reg0=-7)
               for i in range(len(g["cp_pts"]))]
        assert np.array_equal(np.array([q for q, _ in res]), g["cp%d_q" % wn])
        assert np.array_equal(np.array([r_ for _, r_ in res]), g["cp%d_reg" % wn])
    c3, e3, k3 = synth.torus_tet(n_b=4, n_theta=12, n_planes=8)
    m3 = ppo.Mesh(3, c3, e3, k3)
    owner = (np.arange(m3.nelems, dtype=np.int64) * 6 // m3.nelems).astype(np.int32)
    for bridge in (0, 2):
        safe, part = ppo.bfs_buffer_layers(m3, owner, 2, 6, 2, 3, bridge)
        assert np.array_equal(safe, g["bfs%d_safe" % bridge]) and np.array_equal(part, g["bfs%d_part" % bridge])
        assert np.array_equal(ppo.bfs_safe_inward(m3, owner, 2, 2, part, bridge), g["bfs%d_inward" % bridge])
    elems = g["redist_elems"]
    ppe = np.bincount(elems, minlength=300).astype(np.int32)
    psr = ppo.PS.scs([(np.int32, 1)], 300, ppe, C_max=32, sigma=300, V=1024, particle_elements=elems,
                     particle_info=[np.arange(5000, dtype=np.int32)[None, :]])
    assert np.array_equal(ppo.redistribute_particles(psr, 0.4, seed=12345), g["redist_new"])


# ---------------------------------------------------------------- GPU: HIP path == golden
@pytest.fixture(scope="module")
def capi(pp):
    from pumipic_amd import capi as c
    c.init(0)
    return c


@pytest.mark.gpu
def test_gpu_scs_layouts_golden(capi):
    g, keys = _scs_cases()
    for key in keys:
        C, V, sigma, pad = _parse(key)
        ppe = g[key + "_ppe"]
        ps = capi.PS.scs([(np.int32, 1)], len(ppe), ppe, C_=C, sigma=sigma, V=V, pad_strat=pad)
        L = ps.layout()
        for k in ("offsets", "slice_to_chunk", "row_to_element", "mask"):
            assert np.array_equal(L[k], g[key + "_" + k]), (key, k)
        assert [L["C"], L["capacity"]] == list(g[key + "_C"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["scs", "csr"])
def test_gpu_xgcm2d_golden(synth, capi, kind):
    g = _load("xgcm2d_24x96_1000p_30steps.npz")
    pop = common.population_2d(synth, n_b=24, n_theta=96, num_ptcls=1000, mdl_face=6, band_width=3)
    mesh, ps = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, kind)
    fwd, _ = capi.create_gyro_ring_mappings(mesh)
    assert np.array_equal(fwd.to_host(), g["fwd_map"].ravel())
    for step in range(30):
        cap = ps.capacity()
        ids = capi.DevArray.from_host(np.full(cap, -1, dtype=np.int32))
        capi.push_search(mesh, ps, H, K, D, 2.0, ids, seeded=True, looplimit=200)
        _, e = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], ids.to_host()[:cap])
        assert np.array_equal(e, g["elem_ids"][step]), step
        ps.rebuild_commit(ids, 0, 1)
    cap = ps.capacity()
    _, x = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], ps.member(0)[:, :cap])
    assert np.array_equal(x, g["final_x"])
    w = capi.gyro_scatter(mesh, ps, fwd).to_host()
    np.testing.assert_allclose(w, g["scatter_fwd"], rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_gpu_xgcm3d_golden(synth, capi):
    g = _load("xgcm3d_6x24x8_1000p_12steps.npz")
    pop = common.population_3d(synth, n_b=6, n_theta=24, n_planes=8, num_ptcls=1000, mdl_face=5)
    mesh, ps = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    cap = ps.capacity()
    ids = capi.DevArray.from_host(np.full(cap, -1, dtype=np.int32))
    for step in range(12):
        capi.push_search(mesh, ps, H, K, D, 6.0, ids, seeded=(step > 0), looplimit=200)
        _, e = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], ids.to_host()[:cap])
        assert np.array_equal(e, g["elem_ids"][step]), step
        ps.swap_members(0, 1)
    _, x = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], ps.member(0)[:, :cap])
    assert np.array_equal(x, g["final_x"])


@pytest.mark.gpu
def test_gpu_pushsearch_golden(synth, capi):
    g = _load("pushsearch_box6_1000p.npz")
    pop = common.population_box(synth, n=6, num_ptcls=1000)
    mesh, ps = common.gpu_pair(capi, pop, capi.PARTICLE_PUSH)
    for step in range(len(g["elem_ids"])):
        capi.linear_push(ps, 1.0 / 20, -0.5, 0.8, 0.0)
        r = capi.search_mesh_legacy3d(mesh, ps, looplimit=100)
        cap = ps.capacity()
        ids = r["elem_ids"].to_host()[:cap]
        pid, e = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], ids)
        _, f = common.by_id(ps.member(2)[0, :cap], ps.slot_info()[1], r["xface"].to_host()[:cap])
        assert np.array_equal(e, g["elem_ids"][step][pid]) and np.array_equal(f, g["xface"][step][pid])
        capi.update_positions(ps)
        ps.rebuild(ids)


@pytest.mark.gpu
def test_gpu_walls_and_parts_golden(ppo, synth, capi):
    g, pop2, pair, host = _walls_and_parts(capi, synth, True)
    mesh2, ps2 = pair(capi, pop2, capi.PARTICLE_PUSH)
    capi.linear_push(ps2, 0.55, -0.5, 0.8, 0.15)
    cap = ps2.capacity()
    ids, mk = ps2.member(2)[0, :cap], ps2.slot_info()[1]
    topo = ppo.Mesh(3, pop2["coords"], pop2["e2v"], pop2["cls"])  # side numbering for the host functor
    for mt in (0, 1):
        w = capi.trace_particle_through_mesh(
            mesh2, ps2, common.on_device(common.class_interface_functor(topo, mk, [])),
            require_intersection=bool(mt), looplimit=200)
        assert np.array_equal(common.by_id(ids, mk, w["elem_ids"].to_host()[:cap])[1], g["wall%d_elem" % mt])
        assert np.array_equal(common.by_id(ids, mk, w["inter_faces"].to_host()[:cap])[1], g["wall%d_face" % mt])
    for wn in (0, 1):
        q, reg = capi.closest_point_on_triangle(g["cp_tris"], g["cp_pts"], wnormal=bool(wn), reg0=-7)
        assert np.array_equal(q, g["cp%d_q" % wn]) and np.array_equal(reg, g["cp%d_reg" % wn])
    c3, e3, k3 = synth.torus_tet(n_b=4, n_theta=12, n_planes=8)
    m3 = capi.Mesh(3, c3, e3, k3)
    owner = (np.arange(m3.nelems, dtype=np.int64) * 6 // m3.nelems).astype(np.int32)
    d_owner = capi.DevArray.from_host(owner)
    for bridge in (0, 2):
        safe, part = capi.bfs_buffer_layers(m3, d_owner, 2, 6, 2, 3, bridge)
        assert np.array_equal(safe.to_host()[:m3.nelems], g["bfs%d_safe" % bridge])
        assert np.array_equal(part, g["bfs%d_part" % bridge])
        inward = capi.bfs_safe_inward(m3, d_owner, 2, 6, 2, part, bridge)
        assert np.array_equal(inward.to_host()[:m3.nelems], g["bfs%d_inward" % bridge])
    elems = g["redist_elems"]
    ppe = np.bincount(elems, minlength=300).astype(np.int32)
    psr = capi.PS.scs([(np.int32, 1)], 300, ppe, C_=32, sigma=300, V=1024, particle_elements=elems,
                      particle_info=[np.arange(5000, dtype=np.int32)[None, :]])
    assert np.array_equal(capi.redistribute_particles(psr, 0.4, seed=12345).to_host()[:psr.capacity()],
                          g["redist_new"])


# ---------------------------------------------------------------- PICparts, comm arrays, balancer
def _picpart_cases(synth):
    from test_picpart_oracle import slab_owners
    return (("box", synth.kuhn_box(4), 3, 0), ("ann", synth.annulus_tri(n_b=6, n_theta=24, band_width=3), 2, 1)), slab_owners


def test_oracle_picparts_golden(ppo, synth):
    import pumipic_amd_loader
    opp = pumipic_amd_loader.load_oracle_picpart()
    g = _load("picparts.npz")
    cases, slab_owners = _picpart_cases(synth)
    for tag, (c, e, k), dim, axis in cases:
        owner = slab_owners(c, e, 4, axis=axis)
        assert np.array_equal(owner, g[tag + "_owner"])
        mesh = ppo.Mesh(dim, c, e, k)
        P = opp.PicParts(mesh, owner, 4, opp.BFS, opp.BFS, buffer_layers=1, safe_layers=0)
        for r, p in enumerate(P.parts):
            for d in range(dim + 1):
                assert np.array_equal(p.gids[d], g["%s_r%d_d%d_gids" % (tag, r, d)])
                assert np.array_equal(p.comm_index[d], g["%s_r%d_d%d_comm_index" % (tag, r, d)])
                assert np.array_equal(p.full_ids[d], g["%s_r%d_d%d_full_ids" % (tag, r, d)])
                assert np.array_equal(p.is_complete[d], g["%s_r%d_d%d_complete" % (tag, r, d)])
            assert np.array_equal(p.safe, g["%s_r%d_safe" % (tag, r)])
        for d in range(dim + 1):
            red = P.reduce(d, opp.SUM_OP, [g["%s_r%d_d%d_in" % (tag, r, d)] for r in range(4)])
            for r in range(4):
                assert np.array_equal(red[r], g["%s_r%d_d%d_sum" % (tag, r, d)])
        PB = opp.PicParts(mesh, owner, 4, opp.BFS, opp.FULL, buffer_layers=3, safe_layers=1)
        bal = opp.Balancer(PB)
        plan, W, _ = bal.partition_counts([np.full(p.nents[dim], (p.rank + 1) * 50, dtype=np.int32) for p in PB.parts], 1.05)
        assert bal.masks == g[tag + "_sbars"].tolist() and W == g[tag + "_weights_after"].tolist()
        for r in range(4):
            assert np.array_equal(bal.part_index[r], g["%s_r%d_sbar_ids" % (tag, r)])
            assert np.array_equal(np.asarray(plan[r], dtype=np.int64).reshape(-1, 3), g["%s_r%d_plan" % (tag, r)])


@pytest.mark.gpu
def test_gpu_picparts_golden(synth, capi):
    """the HIP parts, reductions and balancer plans against the committed file (no oracle in the loop)"""
    g = _load("picparts.npz")
    cases, slab_owners = _picpart_cases(synth)
    for tag, (c, e, k), dim, axis in cases:
        owner = g[tag + "_owner"]
        mesh = capi.Mesh(dim, c, e, k)
        comms = capi.Comm.local(4)
        parts = [capi.PicPart(mesh, owner, comms[r], capi.PART_BFS, capi.PART_BFS, 0, 1, 0) for r in range(4)]
        for r, p in enumerate(parts):
            for d in range(dim + 1):
                assert np.array_equal(p.array(capi.PART_GIDS, d), g["%s_r%d_d%d_gids" % (tag, r, d)])
                assert np.array_equal(p.array(capi.PART_COMM_INDEX, d), g["%s_r%d_d%d_comm_index" % (tag, r, d)])
                assert np.array_equal(p.array(capi.PART_FULL_IDS, d), g["%s_r%d_d%d_full_ids" % (tag, r, d)])
                assert np.array_equal(p.complete_parts(d), g["%s_r%d_d%d_complete" % (tag, r, d)])
            assert np.array_equal(p.array(capi.PART_SAFE).astype(np.int32), g["%s_r%d_safe" % (tag, r)])
        for d in range(dim + 1):
            devs = [capi.DevArray.from_host(g["%s_r%d_d%d_in" % (tag, r, d)]) for r in range(4)]
            capi.picpart_reduce_all(parts, d, capi.OP_SUM, devs)
            for r in range(4):
                assert np.array_equal(devs[r].to_host(), g["%s_r%d_d%d_sum" % (tag, r, d)])
        lb_comms = capi.Comm.local(4)
        lb_parts = [capi.PicPart(mesh, owner, lb_comms[r], capi.PART_BFS, capi.PART_FULL, 0, 3, 1) for r in range(4)]
        bals = [capi.Balancer(p) for p in lb_parts]
        for r, (p, b) in enumerate(zip(lb_parts, bals)):
            assert np.array_equal(b.sbars(), g[tag + "_sbars"])
            assert np.array_equal(b.sbar_ids(), g["%s_r%d_sbar_ids" % (tag, r)])
            b.partition_begin(np.full(p.nents[dim], (r + 1) * 50, dtype=np.int32))
        for r, b in enumerate(bals):
            b.partition_end(1.05)
            plan, W = b.last_plan()
            assert np.array_equal(np.asarray(plan, dtype=np.int64).reshape(-1, 3), g["%s_r%d_plan" % (tag, r)])
            assert np.array_equal(W, g[tag + "_weights_after"])
        for cm in comms + lb_comms:
            cm.destroy()
