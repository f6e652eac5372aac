"""The deferred second pass of the full re-layout and the record-fed fused push (DESIGN "Rebuild: the
record-fed push"; pp_ps::lazy_rec).  After pp_ps_rebuild_commit / _scatter the particles of a pseudoXGCm-typed
structure stay in the move's 32-B staging records (+ the third member beside them); the next pp_push_search reads them there, anything else
makes the library run the deferred pass first.  Whatever the order of calls, a caller must see exactly what
the reference defines -- checked against the CPU oracle (test/pseudoXGCm.cpp:504-534, scs/SCS_rebuild.h:122-314)."""
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


def _layouts_equal(po, pg):
    lo, lg = po.layout(), pg.layout()
    for k in ("C", "num_chunks", "num_slices", "capacity", "num_rows"):
        assert lo[k] == lg[k], (k, lo[k], lg[k])
    for k in ("offsets", "slice_to_chunk", "row_to_element", "element_to_row"):
        assert np.array_equal(lo[k], lg[k]), k


def _same_population(po, pg, nmembers=5):
    so, mo = po.slot_info()
    sg, mg = pg.slot_info()
    capo, capg = po.capacity(), pg.capacity()
    ido, idg = po.member(2)[0, :capo], pg.member(2)[0, :capg]
    io, eo = common.by_id(ido, mo, so)
    ig, eg = common.by_id(idg, mg, sg)
    assert np.array_equal(io, ig) and np.array_equal(eo, eg)
    for m in range(nmembers):
        _, a = common.by_id(ido, mo, po.member(m)[:, :capo])
        _, b = common.by_id(idg, mg, pg.member(m)[:, :capg])
        assert np.array_equal(a, b), m


def _oracle_step(ppo, mo, po, deg):
    ppo.toroidal_push(po, mo, H, K, D, deg, trig=1)
    return ppo.search_mesh(mo, po, looplimit=200)["elem_ids"]


@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("look", ["never", "after_push", "after_rebuild", "both"])
def test_record_fed_push_and_deferred_pass(ppo, synth, capi, look, shuffle):
    """10 steps of push -> search -> updatePtclPositions + rebuild -> gyroScatter x2 on tets.  `look` says
    when the test reads members through the C-ABI: never (the records go from rebuild to push and only the
    origin is ever missing from the SoA arrays), after the push (origin unpacked on demand), after the
    rebuild (full deferred pass on demand, the push then takes the SoA form), or both."""
    pop = common.population_3d(synth, n_b=5, n_theta=20, n_planes=8, num_ptcls=6000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    common.set_shuffling(po, pg, on=shuffle)
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg)
    pid_g = pg.member(2)[0, :pg.capacity()].copy()
    mask_g = pg.slot_info()[1].copy()
    for step in range(10):
        ids_o = _oracle_step(ppo, mo, po, 6.0)
        ids_g = capi.DevArray(max(pg.capacity(), 1), np.int32)
        capi.push_search(mg, pg, H, K, D, 6.0, ids_g, seeded=False, looplimit=200)
        if step >= 1:
            pg.set_origin_trust(True)
        if look in ("after_push", "both"):
            # x (origin) and x_tgt side by side, phi, b and the id: everything the reference's search leaves
            for m in range(5):
                _, a = common.by_id(po.member(2)[0, :po.capacity()], po.slot_info()[1], po.member(m)[:, :po.capacity()])
                _, b = common.by_id(pg.member(2)[0, :pg.capacity()], pg.slot_info()[1], pg.member(m)[:, :pg.capacity()])
                assert np.array_equal(a, b), (step, m)
            io, eo = common.by_id(po.member(2)[0, :po.capacity()], po.slot_info()[1], ids_o[:po.capacity()])
            ig, eg = common.by_id(pg.member(2)[0, :pg.capacity()], pg.slot_info()[1], ids_g.to_host()[:pg.capacity()])
            assert np.array_equal(io, ig) and np.array_equal(eo, eg), step
        ppo.update_positions(po)
        po.rebuild(ids_o)
        wf, wb = capi.rebuild_scatter(pg, mg, ids_g, [fg, bg], commit=True)
        assert po.nPtcls() == pg.nPtcls() > 0
        _layouts_equal(po, pg)
        assert np.array_equal(ppo.gyro_scatter(mo, po, fo), wf.to_host()), step
        assert np.array_equal(ppo.gyro_scatter(mo, po, bo), wb.to_host()), step
        if look in ("after_rebuild", "both"):
            _same_population(po, pg)
    _same_population(po, pg)
    assert capi.push_search_counters()[0] == 0
    del pid_g, mask_g


def test_deferred_pass_before_other_entry_points(ppo, synth, capi):
    """calls that are not the fused push after a rebuild that deferred its second pass: a plain rebuild
    (no commit: x and x_tgt both travel), a member swap, edited ids with deletions"""
    pop = common.population_3d(synth, n_b=5, n_theta=20, n_planes=8, num_ptcls=5000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    common.set_shuffling(po, pg, on=False)
    rng = np.random.default_rng(5)
    for step in range(8):
        ids_o = _oracle_step(ppo, mo, po, 6.0)
        ids_g = capi.DevArray(max(pg.capacity(), 1), np.int32)
        capi.push_search(mg, pg, H, K, D, 6.0, ids_g, seeded=False, looplimit=200)
        if step % 2 == 0:  # delete a tenth of the particles, by particle id, on both sides
            doomed = rng.choice(5000, 500, replace=False)
            ido = po.member(2)[0, :po.capacity()]
            ids_o = ids_o.copy()
            ids_o[:po.capacity()][np.isin(ido, doomed) & (po.slot_info()[1] > 0)] = -1
            idg = pg.member(2)[0, :pg.capacity()]
            h = ids_g.to_host()
            h[:pg.capacity()][np.isin(idg, doomed) & (pg.slot_info()[1] > 0)] = -1
            ids_g = capi.DevArray.from_host(h)
        if step % 3 == 2:  # no commit: x stays, x_tgt stays
            po.rebuild(ids_o)
            pg.rebuild(ids_g)
        else:
            ppo.update_positions(po)
            po.rebuild(ids_o)
            pg.rebuild_commit(ids_g, 0, 1)
        assert po.nPtcls() == pg.nPtcls() > 0
        _layouts_equal(po, pg)
        if step == 5:  # O(1) member swap right after a deferred rebuild, and back
            pg.swap_members(0, 1)
            pg.swap_members(0, 1)
    _same_population(po, pg)


@pytest.mark.parametrize("seeded", [True, False])
@pytest.mark.parametrize("look", ["never", "after_push", "after_rebuild"])
def test_record_fed_push_2d(ppo, synth, capi, look, seeded):
    """the literal pseudoXGCm loop (triangles, ellipticalPush + search_mesh_2d): the 2-D push writes two
    components of x_tgt, the third one -- logically zero after updatePtclPositions -- is written by the
    record-fed kernel itself"""
    pop = common.population_2d(synth, num_ptcls=6000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    common.set_shuffling(po, pg)
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg)
    for step in range(12):
        ppo.elliptical_push(po, mo, H, K, D, 2.0, trig=1)
        ids_o = ppo.search_mesh_2d(mo, po, looplimit=200)[1]
        if seeded:
            ids_g = capi.DevArray.from_host(np.full(max(pg.capacity(), 1), -1, dtype=np.int32))
        else:  # "no seeds" = every seed -1, the array is not even initialised
            ids_g = capi.DevArray.from_host(np.full(max(pg.capacity(), 1), 123456789, dtype=np.int32))
        capi.push_search(mg, pg, H, K, D, 2.0, ids_g, seeded=seeded, looplimit=200)
        if look == "after_push":
            for m in range(5):
                _, a = common.by_id(po.member(2)[0, :po.capacity()], po.slot_info()[1], po.member(m)[:, :po.capacity()])
                _, b = common.by_id(pg.member(2)[0, :pg.capacity()], pg.slot_info()[1], pg.member(m)[:, :pg.capacity()])
                assert np.array_equal(a, b), (step, m)
        ppo.update_positions(po)
        po.rebuild(ids_o)
        wf, wb = capi.rebuild_scatter(pg, mg, ids_g, [fg, bg], commit=True)
        assert po.nPtcls() == pg.nPtcls() > 0
        _layouts_equal(po, pg)
        assert np.array_equal(ppo.gyro_scatter(mo, po, fo), wf.to_host()), step
        assert np.array_equal(ppo.gyro_scatter(mo, po, bo), wb.to_host()), step
        if look == "after_rebuild":
            _same_population(po, pg)
    _same_population(po, pg)


@pytest.mark.parametrize("dim", [3, 2])
@pytest.mark.parametrize("C", [32, 12, 6, 1])
def test_record_fed_push_other_chunk_heights(ppo, synth, capi, C, dim):
    """The 32-B records and their side words at chunk heights other than 64: a multiple of four takes the
    cooperative four-column fetch (tets) with rows of two tiles in one wave, the others make the library run the
    deferred pass before the push (tets) or read the records lane by lane (triangles).  Eight steps without looking
    at a member, deletions on the way, every member by particle id at the end."""
    if dim == 3:
        pop = common.population_3d(synth, n_b=4, n_theta=16, n_planes=8, num_ptcls=5000)
    else:
        pop = common.population_2d(synth, n_b=10, n_theta=40, num_ptcls=5000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=C)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM, C=C)
    common.set_shuffling(po, pg, on=False)
    rng = np.random.default_rng(C + dim)
    for step in range(8):
        if dim == 3:
            ids_o = _oracle_step(ppo, mo, po, 5.0)
        else:
            ppo.elliptical_push(po, mo, H, K, D, 5.0, trig=1)
            ids_o = ppo.search_mesh_2d(mo, po, looplimit=200)[1]
        ids_g = capi.DevArray(max(pg.capacity(), 1), np.int32)
        if dim == 2:
            ids_g.fill_bytes(0xff)
        capi.push_search(mg, pg, H, K, D, 5.0, ids_g, seeded=False, looplimit=200)
        if step in (2, 5):  # delete by SLOT-independent rule: particles whose new element is in a random tenth
            gone = rng.random(len(pop["e2v"])) < 0.1
            ids_o = ids_o.copy()
            live = ids_o >= 0
            ids_o[live & gone[np.clip(ids_o, 0, None)]] = -1
            h = ids_g.to_host()
            live = h >= 0
            h[live & gone[np.clip(h, 0, None)]] = -1
            ids_g = capi.DevArray.from_host(h)
        ppo.update_positions(po)
        po.rebuild(ids_o)
        pg.rebuild_commit(ids_g, 0, 1)
        assert po.nPtcls() == pg.nPtcls() > 0
        _layouts_equal(po, pg)
    _same_population(po, pg)
