#!/usr/bin/env python
"""Generates tests/golden/*.npz with the CPU oracle (oracle/).  The reference's own meshes are an
empty submodule, so golden vectors are produced on the synthetic inputs of SURVEY 8(c)
(3)-(5): per-step element ids keyed by particle id, final positions and scatter sums.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pumipic_amd_loader  # noqa: E402
import common  # noqa: E402

H, K, D = 1.72479370 - .08, .020558260, 0.6
OUT = os.path.dirname(os.path.abspath(__file__))


def by_id(ps, values):
    se, mk = ps.slot_info()
    cap = ps.capacity()
    ids = ps.member(2)[0, :cap]
    return common.by_id(ids, mk, np.asarray(values)[..., :cap])


def xgcm_2d(ppo, synth):
    pop = common.population_2d(synth, n_b=24, n_theta=96, num_ptcls=1000, mdl_face=6, band_width=3)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=1)
    ps.set_try_shuffling(False)
    fwd, _ = ppo.create_gyro_ring_mappings(mesh, trig=1)
    elems = []
    for step in range(30):
        ppo.elliptical_push(ps, mesh, H, K, D, 2.0, trig=1)
        _, ids, _ = ppo.search_mesh_2d(mesh, ps, looplimit=200)
        elems.append(by_id(ps, ids)[1].copy())
        ppo.update_positions(ps)
        ps.rebuild(ids)
    w = ppo.gyro_scatter(mesh, ps, fwd)
    pid, x = by_id(ps, ps.member(0))
    np.savez_compressed(os.path.join(OUT, "xgcm2d_24x96_1000p_30steps.npz"),
                        elem_ids=np.array(elems, dtype=np.int32), final_x=x, scatter_fwd=w,
                        fwd_map=fwd.astype(np.int32), particle_ids=pid)


def xgcm_3d(ppo, synth):
    pop = common.population_3d(synth, n_b=6, n_theta=24, n_planes=8, num_ptcls=1000, mdl_face=5)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_XGCM, C=1)
    elems = []
    ids = None
    for step in range(12):
        ppo.toroidal_push(ps, mesh, H, K, D, 6.0, trig=1)
        ids = ppo.search_mesh(mesh, ps, elem_ids=ids, looplimit=200)["elem_ids"]
        elems.append(by_id(ps, ids)[1].copy())
        a, b = ps.member(0), ps.member(1)
        tmp = a.copy()
        a[:] = b
        b[:] = tmp
    pid, x = by_id(ps, ps.member(0))
    np.savez_compressed(os.path.join(OUT, "xgcm3d_6x24x8_1000p_12steps.npz"),
                        elem_ids=np.array(elems, dtype=np.int32), final_x=x, particle_ids=pid)


def push_and_search(ppo, synth):
    pop = common.population_box(synth, n=6, num_ptcls=1000)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=1)
    ps.set_try_shuffling(False)
    elems, faces = [], []
    for step in range(12):
        ppo.linear_push(ps, 1.0 / 20, -0.5, 0.8, 0.0)
        r = ppo.search_mesh_legacy3d(mesh, ps, looplimit=100)
        pid, e = by_id(ps, r["elem_ids"])
        _, f = by_id(ps, r["xface"])
        full_e = np.full(1000, -2, dtype=np.int32)
        full_f = np.full(1000, -2, dtype=np.int32)
        full_e[pid] = e
        full_f[pid] = f
        elems.append(full_e)
        faces.append(full_f)
        ppo.update_positions(ps)
        ps.rebuild(r["elem_ids"])
        if ps.nPtcls() == 0:
            break
    np.savez_compressed(os.path.join(OUT, "pushsearch_box6_1000p.npz"),
                        elem_ids=np.array(elems), xface=np.array(faces))


def scs_layouts(ppo):
    out = {}
    rng = np.random.default_rng(42)
    ppes = [np.array([3, 0, 5, 1, 1, 0, 7, 2, 2, 4], dtype=np.int32),
            rng.integers(0, 9, size=37).astype(np.int32),
            np.zeros(6, dtype=np.int32)]
    for i, ppe in enumerate(ppes):
        for C, V, sigma, pad in [(1, 1024, 2**31 - 1, 0), (4, 2, 1, 0), (32, 1024, 2**31 - 1, 1),
                                 (64, 3, 5, 2)]:
            ps = ppo.PS.scs([(np.int32, 1)], len(ppe), ppe, C_max=C, sigma=sigma, V=V, pad_strat=pad)
            L = ps.layout()
            key = "p%d_C%d_V%d_s%d_pad%d" % (i, C, V, min(sigma, 99), pad)
            out[key + "_ppe"] = ppe
            for k in ("offsets", "slice_to_chunk", "row_to_element", "mask"):
                out[key + "_" + k] = L[k]
            out[key + "_C"] = np.array([L["C"], L["capacity"]])
    np.savez_compressed(os.path.join(OUT, "scs_layouts.npz"), **out)


def walls_and_parts(ppo, synth):
    """search_mesh_3d, the functor walk, closest_point_on_triangle, the PICpart BFS layers and
    redistribute_particles on fixed inputs."""
    out = {}
    pop = common.population_box(synth, n=5, num_ptcls=800)
    mesh, ps = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=1)
    ppo.linear_push(ps, 0.55, -0.5, 0.8, 0.15)
    r = ppo.search_mesh_3d(mesh, ps, looplimit=200)
    pid, e = by_id(ps, r["elem_ids"])
    _, f = by_id(ps, r["xface"])
    _, xp = by_id(ps, r["xpoints"].T)
    out.update(s3d_pid=pid, s3d_elem=e, s3d_xface=f, s3d_xpoints=xp)
    # functor walk on a mesh with three material slabs
    coords, e2v, _ = synth.kuhn_box(5)
    cls = (1 + np.floor(coords[e2v][:, :, 1].mean(axis=1) * 3)).astype(np.int32)
    pop2 = dict(pop, cls=cls)
    mesh2, ps2 = common.oracle_pair(ppo, pop2, ppo.PARTICLE_PUSH, C=1)
    ppo.linear_push(ps2, 0.55, -0.5, 0.8, 0.15)
    se, mk = ps2.slot_info()
    for mt in (0, 1):
        w = ppo.trace_particle_through_mesh(mesh2, ps2, common.class_interface_functor(mesh2, mk, []),
                                            require_intersection=bool(mt), looplimit=200)
        pid, e = by_id(ps2, w["elem_ids"])
        _, f = by_id(ps2, w["inter_faces"])
        out["wall%d_elem" % mt] = e
        out["wall%d_face" % mt] = f
    rng = np.random.default_rng(99)
    tris, pts = rng.normal(size=(300, 9)), rng.normal(size=(300, 3)) * 2
    for wn in (0, 1):
        res = [ppo.closest_point_on_triangle(tris[i], pts[i], wnormal=bool(wn), reg0=-7) for i in range(300)]
        out["cp%d_q" % wn] = np.array([q for q, _ in res])
        out["cp%d_reg" % wn] = np.array([g for _, g in res], dtype=np.int32)
    out.update(cp_tris=tris, cp_pts=pts)
    c3, e3, k3 = synth.torus_tet(n_b=4, n_theta=12, n_planes=8)
    m3 = ppo.Mesh(3, c3, e3, k3)
    owner = (np.arange(m3.nelems, dtype=np.int64) * 6 // m3.nelems).astype(np.int32)
    for bridge in (0, 2):
        safe, part = ppo.bfs_buffer_layers(m3, owner, 2, 6, 2, 3, bridge)
        out["bfs%d_safe" % bridge] = safe.copy()
        out["bfs%d_part" % bridge] = part.copy()
        out["bfs%d_inward" % bridge] = ppo.bfs_safe_inward(m3, owner, 2, 2, part, bridge).copy()
    ne = 300
    elems = np.sort(rng.integers(0, ne, size=5000).astype(np.int32))
    ppe = np.bincount(elems, minlength=ne).astype(np.int32)
    psr = ppo.PS.scs([(np.int32, 1)], ne, ppe, C_max=32, sigma=ne, V=1024, particle_elements=elems,
                     particle_info=[np.arange(5000, dtype=np.int32)[None, :]])
    out.update(redist_elems=elems, redist_new=ppo.redistribute_particles(psr, 0.4, seed=12345).copy())
    np.savez_compressed(os.path.join(OUT, "walls_and_parts.npz"), **out)


def picparts(ppo, synth):
    """PICparts, comm arrays and the balancer (oracle/ppo_picpart.py): 4 ranks, slabs of a Kuhn box (tets) and
    of an annulus (triangles), 1-layer BFS buffer / core safe zone for the comm arrays, the reference
    test_lb.cpp's Input (BFS buffer 3, FULL safe) for the balancer"""
    opp = pumipic_amd_loader.load_oracle_picpart()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_picpart_oracle import slab_owners
    out = {}
    for tag, (c, e, k), dim, axis in (("box", synth.kuhn_box(4), 3, 0),
                                      ("ann", synth.annulus_tri(n_b=6, n_theta=24, band_width=3), 2, 1)):
        owner = slab_owners(c, e, 4, axis=axis)
        mesh = ppo.Mesh(dim, c, e, k)
        P = opp.PicParts(mesh, owner, 4, opp.BFS, opp.BFS, buffer_layers=1, safe_layers=0)
        rng = np.random.default_rng(3)
        out[tag + "_owner"] = owner
        for r, p in enumerate(P.parts):
            for d in (0, dim - 1, dim):
                out["%s_r%d_d%d_gids" % (tag, r, d)] = p.gids[d]
                out["%s_r%d_d%d_comm_index" % (tag, r, d)] = p.comm_index[d]
                out["%s_r%d_d%d_full_ids" % (tag, r, d)] = p.full_ids[d]
                out["%s_r%d_d%d_complete" % (tag, r, d)] = p.is_complete[d]
            out["%s_r%d_safe" % (tag, r)] = p.safe
        for d in (0, dim - 1, dim):
            arrs = [rng.standard_normal(p.nents[d] * 2) for p in P.parts]
            red = P.reduce(d, opp.SUM_OP, arrs)
            for r in range(4):
                out["%s_r%d_d%d_in" % (tag, r, d)] = arrs[r]
                out["%s_r%d_d%d_sum" % (tag, r, d)] = red[r]
        if dim == 3:  # round 3: entity dimension 1 of the tet mesh (edges), with draws of its own so that the
            rng1 = np.random.default_rng(31)  # vectors of rounds 1-2 stay what they were
            for r, p in enumerate(P.parts):
                out["%s_r%d_d1_gids" % (tag, r)] = p.gids[1]
                out["%s_r%d_d1_comm_index" % (tag, r)] = p.comm_index[1]
                out["%s_r%d_d1_full_ids" % (tag, r)] = p.full_ids[1]
                out["%s_r%d_d1_complete" % (tag, r)] = p.is_complete[1]
            arrs = [rng1.standard_normal(p.nents[1] * 2) for p in P.parts]
            red = P.reduce(1, opp.SUM_OP, arrs)
            for r in range(4):
                out["%s_r%d_d1_in" % (tag, r)] = arrs[r]
                out["%s_r%d_d1_sum" % (tag, r)] = red[r]
            out[tag + "_edge2verts"] = P.mid[1][0]
        PB = opp.PicParts(mesh, owner, 4, opp.BFS, opp.FULL, buffer_layers=3, safe_layers=1)
        bal = opp.Balancer(PB)
        ppe = [np.full(p.nents[dim], (p.rank + 1) * 50, dtype=np.int32) for p in PB.parts]
        plan, W, w = bal.partition_counts(ppe, 1.05)
        out[tag + "_sbars"] = np.asarray(bal.masks, dtype=np.uint64)
        out[tag + "_weights_after"] = np.asarray(W, dtype=np.int64)
        for r in range(4):
            out["%s_r%d_sbar_ids" % (tag, r)] = bal.part_index[r]
            out["%s_r%d_plan" % (tag, r)] = np.asarray(plan[r], dtype=np.int64).reshape(-1, 3)
    np.savez_compressed(os.path.join(OUT, "picparts.npz"), **out)


if __name__ == "__main__":
    pp = pumipic_amd_loader.load()
    ppo = pumipic_amd_loader.load_oracle()
    picparts(ppo, pp.synth)
    xgcm_2d(ppo, pp.synth)
    xgcm_3d(ppo, pp.synth)
    push_and_search(ppo, pp.synth)
    scs_layouts(ppo)
    walls_and_parts(ppo, pp.synth)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
