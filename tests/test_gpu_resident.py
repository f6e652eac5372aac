"""Resident records (pp_ps_set_resident_records, include/pumipic_hip.h): the pseudoXGCm step with the
particles kept as one 64-B record per slot between the fused push and the rebuild.  Everything the
reference defines must come out the same as with the SoA arrays: element ids by particle id, every member
of every particle, the layout arrays after every rebuild, the gyroScatter fields -- all against the CPU
oracle (test/pseudoXGCm.cpp:504-534, scs/SCS_rebuild.h:122-314)."""
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


def _oracle_step(ppo, dim, mo, po, deg):
    if dim == 3:
        ppo.toroidal_push(po, mo, H, K, D, deg, trig=1)
        return ppo.search_mesh(mo, po, looplimit=200)["elem_ids"]
    ppo.elliptical_push(po, mo, H, K, D, deg, trig=1)
    return ppo.search_mesh_2d(mo, po, looplimit=200)[1]


@pytest.mark.parametrize("inspect", [True, False])
@pytest.mark.parametrize("dim", [3, 2])
def test_resident_records_pseudo_xgcm_steps(ppo, synth, capi, dim, inspect):
    """inspect: the test reads members between the calls, so every step goes SoA -> records (pack) ->
    fused push -> one-pass rebuild -> SoA (materialise).  Without: the particles stay in records from step
    to step and are looked at only at the end."""
    pop = (common.population_3d(synth, n_b=5, n_theta=20, n_planes=8, num_ptcls=6000) if dim == 3 else
           common.population_2d(synth, num_ptcls=6000))
    deg = 6.0 if dim == 3 else 2.0
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    common.set_shuffling(po, pg)
    pg.set_resident_records(True)
    assert pg.resident_records() == 1
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg)
    stayed_on_records = 0
    for step in range(10):
        ids_o = _oracle_step(ppo, dim, mo, po, deg)
        if inspect:
            pid_g, mask_g = pg.member(2)[0, :pg.capacity()].copy(), pg.slot_info()[1].copy()
            assert pg.resident_records() == 1  # looking at a member wrote the SoA arrays back
        ids_g = capi.DevArray(max(pg.capacity(), 1), np.int32)
        capi.push_search(mg, pg, H, K, D, deg, ids_g, seeded=False, looplimit=200)
        assert pg.resident_records() == 2
        if step >= 1 and dim == 3:
            pg.set_origin_trust(True)
        if inspect:
            io, eo = common.by_id(po.member(2)[0, :po.capacity()], po.slot_info()[1], ids_o[:po.capacity()])
            ig, eg = common.by_id(pid_g, mask_g, ids_g.to_host()[:pg.capacity()])
            assert np.array_equal(io, ig) and np.array_equal(eo, eg), step
        ppo.update_positions(po)
        po.rebuild(ids_o)
        wf, wb = capi.rebuild_scatter(pg, mg, ids_g, [fg, bg], commit=True)
        stayed_on_records += pg.resident_records() == 2
        assert po.nPtcls() == pg.nPtcls() > 0
        _layouts_equal(po, pg)
        assert np.array_equal(ppo.gyro_scatter(mo, po, fo), wf.to_host()), step
        assert np.array_equal(ppo.gyro_scatter(mo, po, bo), wb.to_host()), step
        if inspect:
            _same_population(po, pg)
    assert stayed_on_records >= 5  # (a rebuild that keeps the layout takes the SoA in-place path)
    _same_population(po, pg)
    assert capi.push_search_counters()[0] == 0


def test_resident_records_edited_ids_and_plain_rebuild(ppo, synth, capi):
    """ids edited after the search (deletions): pp_ps_ids_modified makes the rebuild recount; a rebuild
    without the commit keeps x and x_tgt apart; switching the mode off writes the SoA arrays back."""
    pop = common.population_3d(synth, n_b=5, n_theta=20, n_planes=8, num_ptcls=5000)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    common.set_shuffling(po, pg, on=False)
    pg.set_resident_records(True)
    rng = np.random.default_rng(5)
    for step in range(6):
        ids_o = _oracle_step(ppo, 3, mo, po, 6.0)
        pid_g, mask_g = pg.member(2)[0, :pg.capacity()].copy(), pg.slot_info()[1].copy()
        ids_g = capi.DevArray(max(pg.capacity(), 1), np.int32)
        capi.push_search(mg, pg, H, K, D, 6.0, ids_g, seeded=False, looplimit=200)
        if step % 2 == 0:  # delete a tenth of the particles, by particle id, on both sides
            doomed = rng.choice(5000, 500, replace=False)
            ido = po.member(2)[0, :po.capacity()]
            ids_o = ids_o.copy()
            ids_o[:po.capacity()][np.isin(ido, doomed) & (po.slot_info()[1] > 0)] = -1
            h = ids_g.to_host()
            h[:pg.capacity()][np.isin(pid_g, doomed) & (mask_g > 0)] = -1
            ids_g = capi.DevArray.from_host(h)
            pg.ids_modified()
        if step % 3 == 2:  # no commit: x stays, x_tgt stays
            po.rebuild(ids_o)
            pg.rebuild(ids_g)
        else:
            ppo.update_positions(po)
            po.rebuild(ids_o)
            pg.rebuild_commit(ids_g, 0, 1)
        assert po.nPtcls() == pg.nPtcls() > 0
        _layouts_equal(po, pg)
    pg.set_resident_records(False)
    assert pg.resident_records() == 0
    _same_population(po, pg)


def test_resident_records_other_particle_types_stay_soa(ppo, synth, capi):
    """a structure whose members are not the 60-byte pseudoXGCm type ignores the switch"""
    pop = common.population_box(synth, n=4, num_ptcls=500)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_PUSH)
    pg.set_resident_records(True)
    ids = capi.DevArray(max(pg.capacity(), 1), np.int32)
    assert pg.resident_records() == 1
    pg.rebuild(capi.DevArray.from_host(pg.slot_info()[0].astype(np.int32)))
    assert pg.resident_records() == 1 and pg.nPtcls() == 500
    del ids


@pytest.mark.parametrize("dim", [3, 2])
def test_resident_records_in_place_rebuild(ppo, synth, capi, dim):
    """shuffle mode 2 on records: the counting happens where the walks end, only the particles that change
    rows move (whole records), rows that overflow trade places or move into appended chunks.  The layout is
    this library's (valid SCS, rows not sorted by count), so what is compared with the oracle is what the
    reference defines: element ids by particle id, every member of every particle, the per-element
    populations and the gyroScatter fields."""
    pop = (common.population_3d(synth, n_b=6, n_theta=24, n_planes=8, num_ptcls=30000) if dim == 3 else
           common.population_2d(synth, n_b=16, n_theta=64, num_ptcls=30000, mdl_face=4, band_width=4))
    ne = len(pop["e2v"])
    deg = 6.0 if dim == 3 else 2.0
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM)
    mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
    pg.set_try_shuffling(2)
    pg.set_resident_records(True)
    fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
    fg, bg = capi.create_gyro_ring_mappings(mg)
    in_place = 0
    for step in range(12):
        ids_o = _oracle_step(ppo, dim, mo, po, deg)
        inspect = step % 4 == 3
        if inspect:
            pid_g, mask_g = pg.member(2)[0, :pg.capacity()].copy(), pg.slot_info()[1].copy()
        ids_g = capi.DevArray(max(pg.capacity(), 1), np.int32)
        capi.push_search(mg, pg, H, K, D, deg, ids_g, seeded=False, looplimit=200)
        assert pg.resident_records() == 2
        if step >= 1 and dim == 3:
            pg.set_origin_trust(True)
        if inspect:
            io, eo = common.by_id(po.member(2)[0, :po.capacity()], po.slot_info()[1], ids_o[:po.capacity()])
            ig, eg = common.by_id(pid_g, mask_g, ids_g.to_host()[:pg.capacity()])
            assert np.array_equal(io, ig) and np.array_equal(eo, eg), step
        ppo.update_positions(po)
        po.rebuild(ids_o)
        st0 = pg.rebuild_stats()
        wf, wb = capi.rebuild_scatter(pg, mg, ids_g, [fg, bg], commit=True)
        in_place += pg.rebuild_stats()[0] > st0[0]
        assert po.nPtcls() == pg.nPtcls() > 0
        assert np.array_equal(ppo.gyro_scatter(mo, po, fo), wf.to_host()), step
        assert np.array_equal(ppo.gyro_scatter(mo, po, bo), wb.to_host()), step
        if inspect or step == 11:
            _same_population(po, pg)
            common.check_scs_valid(pg, ne)
    assert in_place >= 6, in_place
