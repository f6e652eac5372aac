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
