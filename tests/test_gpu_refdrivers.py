"""The reference's OWN drivers, compiled unchanged against this library (tools/ref_conformance.py --build, run by
__graft_entry__.build() where /root/reference exists; the executables under tests/_refdrivers/ travel to the GPU
box like the library's .so), run on the MI355X.

test/pseudoXGCm.cpp is the reference text byte for byte: its push is the user lambda of test/ellipticalPush.hpp, its
scatter the user lambdas of test/gyroScatter.hpp (double atomics through ps::parallel_for), its search / rebuild /
migrate are this library.  The final state (PP_DUMP_ON_DELETE, a hook of the mirror headers -- the driver source
cannot be touched) must equal, particle by particle, the final state of drivers/pseudoXGCm, the adaptation that
tests/test_gpu_driver.py checks against the oracle(libm)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "pumi-pic_amd", "drivers")
REFDRV = os.path.join(ROOT, "tests", "_refdrivers")


def _need(name):
    exe = os.path.join(REFDRV, name)
    if not os.path.exists(exe):
        pytest.skip("tests/_refdrivers/%s was not built (needs the reference tree at build time)" % name)
    return exe


def _load_ref_dump(prefix, name):
    import glob
    metas = sorted(glob.glob("%s_ps_%s_r0_*_meta.txt" % (prefix, name)))
    assert metas, "no dump of structure %r" % name
    base = metas[-1][:-len("_meta.txt")]
    f = open(base + "_meta.txt").read().split()
    cap, stride, nptcls, ne, nm = (int(v) for v in f[:5])
    bytes_ncomp = [(int(f[5 + 2 * m]), int(f[6 + 2 * m])) for m in range(nm)]
    out = dict(cap=cap, stride=stride, n=nptcls, ne=ne,
               mask=np.fromfile(base + "_mask.u8", dtype=np.uint8)[:cap].astype(bool),
               elem=np.fromfile(base + "_elem.i32", dtype=np.int32)[:cap], members=[])
    for m, (b, c) in enumerate(bytes_ncomp):
        raw = np.fromfile(base + "_m%d.bin" % m, dtype=np.uint8)
        assert raw.size == stride * b * c
        out["members"].append(raw.reshape(c, stride * b))
    return out, bytes_ncomp


def _by_id(ids, mask, values):
    live = np.flatnonzero(mask)
    order = np.argsort(ids[live], kind="stable")
    return ids[live][order], np.asarray(values)[..., live[order]]


@pytest.mark.parametrize("nptcl,steps", [(2_000_000, 8)])
def test_reference_pseudoxgcm_source_runs_unchanged(pp, tmp_path, nptcl, steps):
    exe = _need("pseudoXGCm")
    s = pp.synth
    subprocess.check_call(["make", "-C", DRV, "-s"])
    # an annulus whose triangles are wide enough that every ring point (radius <= 0.038) is reached within the 100
    # walk iterations test/gyroScatter.hpp:64 allows: 25 088 triangles, inner cells ~4 mm wide
    coords, e2v, cls = s.annulus_tri(n_b=49, n_theta=256, b_lo=0.2, band_width=4)
    mesh_file = str(tmp_path / "annulus.bin")
    s.write_mesh_bin(mesh_file, 2, coords, e2v, cls)
    deg, mdl = 0.5, 12
    args = [mesh_file, str(nptcl), str(mdl), str(steps), str(deg), "0"]
    ref_prefix, mir_prefix = str(tmp_path / "ref"), str(tmp_path / "mir")
    r = subprocess.run([exe] + args, env=dict(os.environ, PP_DUMP_ON_DELETE=ref_prefix), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "loop limit" not in r.stderr, r.stderr[-2000:]  # every ring point and every particle was found
    assert "done" in r.stderr and ("iter %d particles " % steps) in r.stderr, r.stderr[-2000:]
    assert ("iter 1 particles %d" % nptcl) in r.stderr
    assert re.search(r"Ptcl LB <max, min, avg, imb>: (\d+) \1 ", r.stdout), r.stdout[-1500:]
    m = subprocess.run([os.path.join(DRV, "pseudoXGCm")] + args, env=dict(os.environ, PP_DRIVER_DUMP=mir_prefix),
                       capture_output=True, text=True, timeout=900)
    assert m.returncode == 0, m.stderr[-2000:]

    ref, bn = _load_ref_dump(ref_prefix, "ps")
    assert bn == [(8, 3), (8, 3), (4, 1), (4, 1), (4, 1)]  # MemberTypes<Vector3d, Vector3d, int, float, float>
    cap, stride = ref["cap"], ref["stride"]
    x = np.stack([ref["members"][0][c].view(np.float64)[:cap] for c in range(3)])
    xt = np.stack([ref["members"][1][c].view(np.float64)[:cap] for c in range(3)])
    pid = ref["members"][2][0].view(np.int32)[:cap]
    b = ref["members"][3][0].view(np.float32)[:cap]
    phi = ref["members"][4][0].view(np.float32)[:cap]

    base = "%s_r0_final" % mir_prefix
    mcap, mstride, mn = (int(v) for v in open(base + "_meta.txt").read().split())
    mmask = np.fromfile(base + "_mask.u8", dtype=np.uint8)[:mcap].astype(bool)
    mx = np.fromfile(base + "_x.f64", dtype=np.float64)
    mx = np.stack([mx[c * mstride:c * mstride + mcap] for c in range(3)])
    melem = np.fromfile(base + "_elem.i32", dtype=np.int32)[:mcap]
    mid = np.fromfile(base + "_id.i32", dtype=np.int32)[:mcap]
    mb = np.fromfile(base + "_b.f32", dtype=np.float32)[:mcap]
    mphi = np.fromfile(base + "_phi.f32", dtype=np.float32)[:mcap]

    # (a particle whose ellipse leaves the polygonal domain between two boundary vertices exits and is deleted, in
    #  both drivers alike)
    assert ref["n"] == mn == int(ref["mask"].sum()) and nptcl - 16 <= mn <= nptcl
    ir, er = _by_id(pid, ref["mask"], ref["elem"])
    im, em = _by_id(mid, mmask, melem)
    assert np.array_equal(ir, im), "the two drivers hold different particles"
    assert np.array_equal(er, em), "%d particles end in another element" % int((er != em).sum())
    for a, bb in ((x, mx), (b, mb), (phi, mphi)):
        _, va = _by_id(pid, ref["mask"], a)
        _, vb = _by_id(mid, mmask, bb)
        assert np.array_equal(va, vb)
    _, vt = _by_id(pid, ref["mask"], xt)
    assert not vt.any()  # updatePtclPositions zeroed the targets
    # the scatter: the reference's user lambdas (double atomics) against the library's gyro scatter kernel
    fwd = np.fromfile(ref_prefix + "_tag_0_ptclToMeshScatterFwd_r0.f64")
    bkwd = np.fromfile(ref_prefix + "_tag_0_ptclToMeshScatterBkwd_r0.f64")
    sync = np.fromfile(ref_prefix + "_tag_0_ptclToMeshSync_r0.f64")
    g_f, g_b = np.fromfile(base + "_fwd.f64"), np.fromfile(base + "_bkwd.f64")
    # (the reference places the ring points with the device libm's cos / sin, the library with the sincos it shares
    #  with the oracle: a point that lies on an edge may map to the neighbouring triangle -- a handful of vertices)
    ndiff = int((fwd != g_f).sum())
    print("scatter: %d of %d vertices differ between the reference's lambdas and the library's kernel" % (ndiff, len(fwd)))
    assert ndiff <= 64 and abs(float(fwd.sum()) - float(g_f.sum())) <= 1e-9 * float(g_f.sum())
    assert np.array_equal(fwd, bkwd) and np.array_equal(g_f, g_b)
    assert np.array_equal(sync[0::2], fwd) and np.array_equal(sync[1::2], bkwd)  # gyroSync on one rank


def test_reference_pseudoxgcm_small_prints_every_particle(pp, tmp_path):
    """<= 30 particles: the reference source prints every particle's element after every rebuild through device
    printf (test/pseudoXGCm.cpp:121-140); particles keep their ids and every particle is reported once per step."""
    exe = _need("pseudoXGCm")
    s = pp.synth
    coords, e2v, cls = s.annulus_tri(n_b=14, n_theta=64)
    mesh_file = str(tmp_path / "small.bin")
    s.write_mesh_bin(mesh_file, 2, coords, e2v, cls)
    r = subprocess.run([exe, mesh_file, "24", "4", "3", "2.0", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2000:])
    rows = re.findall(r"Rank 0 Ptcl: (\d+) has Element (\d+) and id (\d+)", r.stdout)
    assert len(rows) == 3 * 24
    ids = sorted(int(i) for _, _, i in rows)
    assert ids == sorted(list(range(24)) * 3) or len(set(ids)) == 24
    assert all(0 <= int(e) < len(e2v) for _, e, _ in rows)


@pytest.mark.parametrize("structure,strat", [(0, 1), (1, 2), (0, 3)])
def test_reference_ps_combo160_source_runs_unchanged(tmp_path, structure, strat):
    """performance_tests/ps_combo160.cpp + particle_structs/test/Distribute.cpp, unchanged: 100 pseudo-pushes and 100
    redistribute + migrate rounds.  The pseudo-push gives every particle a unique (nums, lint) record; the records
    must survive the 100 rebuilds as a set."""
    exe = _need("ps_combo160")
    ne, npt = 5000, 200000
    prefix = str(tmp_path / "combo")
    r = subprocess.run([exe, str(ne), str(npt), str(strat), str(structure)],
                       env=dict(os.environ, PP_DUMP_ON_DELETE=prefix), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Beginning migrate on structure" in r.stdout
    # the row the reference's structures add to the timing table (scs/SCS_migrate.h:21, csr/CSR_migrate.hpp:30)
    assert re.search(r"(Sell-32-ne|CSR) particle migration +[0-9.]+ +100 ", r.stdout + r.stderr), r.stderr[-1500:]
    name = "Sell-32-ne" if structure == 0 else "ptcls"
    d, bn = _load_ref_dump(prefix, name)
    assert bn == [(8, 17), (4, 4), (8, 1)]  # MemberTypes<double[17], int[4], long>
    cap = d["cap"]
    assert d["n"] == npt == int(d["mask"].sum())
    live = d["mask"]
    lint = d["members"][2][0].view(np.int64)[:cap][live]
    nums = np.stack([d["members"][1][c].view(np.int32)[:cap][live] for c in range(4)])
    assert len(np.unique(lint)) == npt  # every record is still there, once
    for i in range(4):
        assert np.array_equal(nums[i], 4 * lint.astype(np.int32) + i)
    elem = d["elem"][live]
    assert elem.min() >= 0 and elem.max() < ne
    # (dbls = 10.3^3 / sqrt(p) / sqrt(e) + parentElmData(e): infinite for the particles pushed in element 0, as in
    #  the reference)
    dbl0 = d["members"][0][0].view(np.float64)[:cap][live]
    assert np.isfinite(dbl0).mean() > 0.99 and (dbl0[np.isfinite(dbl0)] > 0).all()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_reference_pseudoxgcm_source_two_ranks(pp, tmp_path, world):
    """(4 ranks: testing.cmake pseudoXGCm_24kElms_4 / _120kElms_4.)
    The unchanged test/pseudoXGCm.cpp as TWO rank processes sharing the GPU (PP_COMM=tcp): pumipic::read cuts the
    element-block parts, `p::Distributor<> dist(nBuffers, buffered_ranks)` lists self + the other rank,
    migrate_lb_ptcls -> ParticleStructure::migrate moves the particles that leave a block, gyroSync ->
    reduceCommArray sums the fields, MPI_Allreduce / MPI_Barrier / printPtclImb run over the library's communicator.
    No particle is lost, every particle ends on the rank that owns its element, both ranks hold the same synced field."""
    import glob
    exe = _need("pseudoXGCm")
    s = pp.synth
    c, e, cl = s.annulus_tri(n_b=24, n_theta=96, b_lo=0.2, band_width=3)
    ne = len(e)
    mesh_file = str(tmp_path / "annulus.bin")
    s.write_mesh_bin(mesh_file, 2, c, e, cl)
    npt, steps = 40000, 10
    port = _free_port()
    prefix = str(tmp_path / "two")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), PP_DEVICE="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PP_COMM="tcp", PP_COMM_PORT=str(port),
                   PP_DUMP_ON_DELETE=prefix)
        procs.append(subprocess.Popen([exe, mesh_file, str(npt), "6", str(steps), "2.0", "1"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1500:], se[-3000:])
    so0, se0 = outs[0]
    assert ("world ranks %d" % world) in so0 and "pre-barrier enabled" in se0 and "done" in se0
    m = re.search(r"particles created (\d+)", se0)
    assert m
    created = int(m.group(1))
    assert 0 < created <= npt
    assert ("iter %d particles %d" % (steps, created)) in se0, se0[-1500:]  # nobody lost across the migrations
    assert re.search(r"Ptcl LB <max, min, avg, imb>: \d+ \d+", so0)
    assert "Reduced Timing Summary with" in se0 and "migration" in se0
    total, fields = 0, []
    for r in range(world):
        metas = sorted(glob.glob("%s_ps_ps_r%d_*_meta.txt" % (prefix, r)))
        assert metas, "rank %d wrote no dump" % r
        base = metas[-1][:-len("_meta.txt")]
        f = open(base + "_meta.txt").read().split()
        cap, nptcls = int(f[0]), int(f[2])
        mask = np.fromfile(base + "_mask.u8", dtype=np.uint8)[:cap].astype(bool)
        elem = np.fromfile(base + "_elem.i32", dtype=np.int32)[:cap]
        assert int(mask.sum()) == nptcls
        total += nptcls
        owner = (elem[mask].astype(np.int64) * world) // ne
        assert (owner == r).all(), "rank %d holds particles of another rank's elements" % r
        fields.append(np.fromfile("%s_tag_0_ptclToMeshSync_r%d.f64" % (prefix, r)))
    assert total == created
    assert all(np.array_equal(fields[0], f) for f in fields[1:]) and fields[0].sum() > 0


# ---------------------------------------------------------------- the reference's OWN particle-structure tests
# particle_structs/test/testing.cmake:1-33, unchanged sources (tools/ref_conformance.py): the programs print
# "All tests passed" / return the number of failed checks.
def _run_ranks(cmd, world, cwd, timeout=600, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), PP_DEVICE="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PP_COMM="tcp", PP_COMM_PORT=str(port))
        env.update(extra_env or {})
        procs.append(subprocess.Popen(cmd, env=env, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    return [(p.returncode, so, se) for p, (so, se) in zip(procs, outs)]


@pytest.mark.parametrize("name", ["typeTest", "initParticles", "buildSCSTest", "lambdaTest", "test_scs_padding"])
def test_reference_particle_structs_unit_programs(tmp_path, name):
    """typeTest / initParticles / buildSCSTest / lambdaTest / scs_padding of particle_structs/test, unchanged"""
    exe = _need(name)
    (rc, so, se), = _run_ranks([exe], 1, str(tmp_path), timeout=300)
    assert rc == 0, (so[-2000:], se[-2000:])
    assert "FAIL" not in so.upper().replace("FAILS", "") or "All tests passed" in so, so[-2000:]


@pytest.mark.parametrize("ranks,ne,npt,estrat,pstrat,tag", [
    (1, 5, 25, 0, 0, "small_ptcls_e5_p25_r0"),
    (1, 500, 100000, 0, 2, "medium_ptcls_e500_p10e5_r0"),
    (4, 5, 25, 0, 2, "small_ptcls_e5_p25_r4"),
    (4, 100, 10000, 0, 2, "small_ptcls_e100_p10k_r4"),
    (4, 0, 0, 0, 0, "empty_ptcls"),
    (4, 100, 0, 0, 0, "no_ptcls_e100"),
])
def test_reference_test_structure_passes(tmp_path, ranks, ne, npt, estrat, pstrat, tag):
    """write_particles <ne> <np> <element strat> <particle strat> <prefix>, then test_structure <prefix>, with the rank
    counts of the reference's ctest plan (testing.cmake:17-31): the reference's own checks of counts, parallel_for,
    getPIDs, the five rebuild scenarios, migrateSendRight / migrateSendToOne, printMetrics, copy to the host and back,
    getComponents, migrate-to-empty-and-refill -- on Sell-32-sigma-max, Sell-32-sigma-1-V10 and CSR -- must all pass
    on this library."""
    wr, ts = _need("write_particles"), _need("test_structure")
    res = _run_ranks([wr, str(ne), str(npt), str(estrat), str(pstrat), tag], ranks, str(tmp_path))
    for rc, so, se in res:
        assert rc == 0, (so[-1500:], se[-1500:])
    for r in range(ranks):
        assert os.path.exists(os.path.join(str(tmp_path), "%s_%d.ptl" % (tag, r)))
    res = _run_ranks([ts, tag], ranks, str(tmp_path), timeout=900)
    for r, (rc, so, se) in enumerate(res):
        assert rc == 0, "rank %d: %s\n%s" % (r, so[-3000:], se[-3000:])
        assert "[ERROR]" not in so and "[ERROR]" not in se, (so[-3000:], se[-3000:])
    assert "All tests passed" in res[0][1], res[0][1][-2000:]


# ---------------------------------------------------------------- the reference's OWN 2-D search test
# test/search2d.cpp (test/CMakeLists.txt: `search2d ${TEST_DATA_DIR}`), unchanged source built WITH asserts: each of its
# 14 walks on the 8-triangle plate ends in a device-side `assert(e == destElm || e == altDestElm)` (search2d.cpp:176).
# Its second half reads xgc/24k.osh, a mesh of the separate pumipic-data repository that is not in the reference tree:
# the program must get through all of testTri8 and stop exactly there.
_SEARCH2D_DEST = [(5,), (5,), (5,), (0,), (0,), (0,), (1,), (3, 5), (7,), (0, 2), (3,), (0, 2), (1,), (4,)]


def test_reference_search2d_tri8_asserts_hold(pp, tmp_path):
    exe = _need("search2d")
    s = pp.synth
    coords, e2v, cls = s.plate_tri8_pardiag()
    os.makedirs(os.path.join(str(tmp_path), "plate"))
    s.write_mesh_bin(os.path.join(str(tmp_path), "plate", "tri8_parDiag.osh"), 2, coords, e2v, cls)
    (rc, so, se), = _run_ranks([exe, str(tmp_path)], 1, str(tmp_path), timeout=300)
    assert "Mesh loaded with <v e f r> 9 16 8 8" in so, (so[-2000:], se[-2000:])
    import re
    got = [int(m.group(1)) for m in re.finditer(r"pid 0 elm (\d+) \(x,y\)", so)]
    assert len(got) == len(_SEARCH2D_DEST), (got, so[-3000:], se[-3000:])
    for e, ok in zip(got, _SEARCH2D_DEST):
        assert e in ok, (got, _SEARCH2D_DEST)
    # every assert held (an abort would show as a signal, rc < 0); the only failure is the absent 24k mesh
    assert rc == 1 and "xgc/24k.osh: not a mesh container" in se, (rc, se[-1500:])
    assert "Assertion" not in se


# ---------------------------------------------------------------- the reference's OWN search test: test/test_adj.cpp
# (testing.cmake: `test_adj plate/tri8.osh`, `test_adj cube/7k.osh`), unchanged source, asserts on.  The program takes
# any plate or cube ("Assumes the plate or cube mesh are used", test_adj.cpp:586); the meshes here are this repo's.
def _plate_tris(n):
    """the unit square in 2 n^2 triangles, counter-clockwise"""
    g = np.arange(n + 1)
    X, Y = np.meshgrid(g, g, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1).astype(np.float64) / n
    vid = lambda i, j: i * (n + 1) + j
    tris = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
            tris += [[a, b, c], [a, c, d]] if (i + j) % 2 == 0 else [[a, b, d], [b, c, d]]
    e2v = np.asarray(tris, dtype=np.int32)
    return coords, e2v, np.ones(len(e2v), dtype=np.int32)


@pytest.mark.parametrize("mesh_kind", ["tri8", "plate_2x24x24", "cube_7986", "cube_48"])
def test_reference_test_adj_passes(pp, tmp_path, mesh_kind):
    """100 and 1 000 000 particles (PP_USE_GPU): internal and vertex / edge / face starts, barycentric walk and walk
    with wall intersections, each judged by the reference's own check_initial_parents, intersection-on-face,
    along-the-path and inside-the-bounding-box checks (test_adj.cpp:565-760) -- on this library's search."""
    exe = _need("test_adj")
    s = pp.synth
    if mesh_kind == "tri8":
        dim, (c, e, cl) = 2, s.plate_tri8_pardiag()
    elif mesh_kind.startswith("plate"):  # (fine plates and cubes: the next test)
        dim, (c, e, cl) = 2, _plate_tris(24)
    else:
        dim, (c, e, cl) = 3, s.kuhn_box({"cube_7986": 11, "cube_48": 2}[mesh_kind])
    mesh_file = str(tmp_path / (mesh_kind + ".osh"))
    s.write_mesh_bin(mesh_file, dim, c, e, cl)
    (rc, so, se), = _run_ranks([exe, mesh_file], 1, str(tmp_path), timeout=900)
    assert rc == 0 and "All Tests Passed" in se, (rc, so[-3000:], se[-3000:])
    assert "[ERROR]" not in so and "[ERROR]" not in se, (so[-3000:], se[-3000:])


def test_reference_moller_trumbore_test_passes(pp, tmp_path):
    """test/moller_trumbore_line_tri_test.cpp wants o = (0, -0.2, -0.5) inside element 0, z = (0, -0.2, 0.9) inside
    element 12, and the ray o -> z to leave element 12 through its FIRST face at height 1 (the file of the reference's
    data repository is not here): two Kuhn cubes stacked in z, renumbered so."""
    exe = _need("moller_trumbore_test")
    s = pp.synth
    c, e, cl = s.kuhn_box(2, lo=(-0.6, -0.9, -1.0), hi=(1.4, 1.1, 1.0))
    e = e.copy()

    def holder(p):
        for t in range(len(e)):
            M = c[e[t]]
            T = np.column_stack([M[1] - M[0], M[2] - M[0], M[3] - M[0]])
            w = np.linalg.solve(T, p - M[0])
            if w.min() > 1e-9 and w.sum() < 1 - 1e-9:
                return t
        raise AssertionError("point on an element boundary")

    t0, t12 = holder(np.array([0.0, -0.2, -0.5])), holder(np.array([0.0, -0.2, 0.9]))
    order = [t for t in range(len(e)) if t not in (t0, t12)]
    order = [t0] + order[:11] + [t12] + order[11:]
    e = e[order]
    # element 12: the three vertices at height 1 first (face 0 = local vertices {0, 2, 1}), positive volume
    v = list(e[12])
    top = [x for x in v if abs(c[x][2] - 1.0) < 1e-12]
    assert len(top) == 3
    bottom = [x for x in v if x not in top][0]
    for perm in ([0, 1, 2], [0, 2, 1]):
        q = [top[perm[0]], top[perm[1]], top[perm[2]], bottom]
        M = c[q]
        if np.dot(np.cross(M[1] - M[0], M[2] - M[0]), M[3] - M[0]) > 0:
            e[12] = q
    mesh_file = str(tmp_path / "cubes.msh")
    import importlib
    importlib.import_module(pp.__name__ + ".meshio").write_gmsh(mesh_file, 3, c, e, cl[order])
    (rc, so, se), = _run_ranks([exe, mesh_file], 1, str(tmp_path), timeout=300)
    assert rc == 0 and "[ERROR]" not in so, (rc, so[-3000:], se[-2000:])
    assert so.count("intersected face") >= 1 and "did not intersect" in so


def test_reference_pseudoxgcm_scatter_passes(pp, tmp_path):
    """test/pseudoXGCm_scatter.cpp on the 8-triangle plate: the reference's gyro scatter lambdas (test/gyroScatter.hpp)
    through ps::parallel_for and device atomics; its OMEGA_H_CHECKs hold the known vertex sums 2, 12, 0, 2/3."""
    exe = _need("pseudoXGCm_scatter")
    s = pp.synth
    c, e, cl = s.plate_tri8_pardiag()
    mesh_file = str(tmp_path / "tri8_parDiag.osh")
    s.write_mesh_bin(mesh_file, 2, c, e, cl)
    (rc, so, se), = _run_ranks([exe, mesh_file], 1, str(tmp_path), timeout=300)
    assert rc == 0 and "done" in se and "assertion" not in so, (rc, so[-3000:], se[-2000:])


@pytest.mark.parametrize("which", ["test1", "test3"])
def test_reference_barycentric_passes(tmp_path, which):
    """test/test_barycentric.cpp + src/unit_tests.hpp: find_barycentric_tet at the vertices of a tet
    (testing.cmake: barycentric_3 runs test1).  `test2` is not run by the reference's plan and cannot pass anywhere:
    unit_tests.hpp:55-63 stores the coordinates in an integer array before comparing them with 0.1, 0.15, ..."""
    exe = _need("barycentric")
    (rc, so, se), = _run_ranks([exe, which], 1, str(tmp_path), timeout=300)
    assert rc == 0, (so[-2000:], se[-2000:])


# ---------------------------------------------------------------- BASELINE configs[0]: the reference's pseudoPushAndSearch
# test/pseudoPushAndSearch.cpp unchanged (testing.cmake: `pseudoPushAndSearch <mesh> ignored 200 <model face> dx dy dz`).
# The pumipic-data meshes are not here; the mesh is a Kuhn box written as Gmsh 2.2 whose boundary triangles on y == 0
# carry the model-face id, which the reader turns into the sides' class_id (pseudoPushAndSearch.cpp:231).  The oracle
# runs the same loop; the program's per-iteration particle counts and its has_particles tag must equal the oracle's.
# (16, 100 000): BASELINE configs[0] at its size as SURVEY 8(d) restates it -- 24 576 tets, 100 000 particles, 30 steps
@pytest.mark.parametrize("n,npt,ranks", [(6, 20000, 1), (6, 20000, 2), (16, 100000, 1)])
def test_reference_pseudo_push_and_search_source_matches_oracle(pp, ppo, tmp_path, n, npt, ranks):
    import importlib
    import common
    exe = _need("pseudoPushAndSearch")
    s = pp.synth
    coords, e2v, cls = s.kuhn_box(n)
    faces = [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)]
    tris = [[t[a], t[b], t[c]] for t in e2v for a, b, c in faces if (np.abs(coords[[t[a], t[b], t[c]], 1]) < 1e-12).all()]
    mdl_face = 156
    mesh_file = str(tmp_path / "box.msh")
    importlib.import_module(pp.__name__ + ".meshio").write_gmsh(
        mesh_file, 3, coords, e2v, cls, sides=(np.asarray(tris), np.full(len(tris), mdl_face)))
    prefix = str(tmp_path / "dump")
    # (testing.cmake pseudoPushAndSearch_t2_r2: an owner file, one rank per line of elements; all particles start on rank 0,
    # migrate_lb_ptcls moves and balances them -- the sum over ranks and the latest visit over ranks are the serial run's)
    owner_file = str(tmp_path / "owners.ptn")
    with open(owner_file, "w") as f:
        f.write("\n".join(str(int(e * ranks // len(e2v))) for e in range(len(e2v))) + "\n")
    res = _run_ranks([exe, mesh_file, owner_file, str(npt), str(mdl_face), "-0.5", "0.8", "0"], ranks,
                     str(tmp_path), timeout=600, extra_env={"PP_DUMP_ON_DELETE": prefix})
    for rc, so, se in res:
        assert rc == 0, (rc, so[-3000:], se[-3000:])
    so, se = res[0][1], res[0][2]
    assert "done" in se
    assert "mesh elements classified on model face %d: %d" % (mdl_face, len(tris)) in so, so[:3000]
    per_rank = [[int(m.group(1)) for m in re.finditer(r"PS on rank %d has Elements: \d+\. Ptcls (\d+)\." % r, res[r][1])]
                for r in range(ranks)]
    counts = [sum(c) for c in zip(*per_rank)]
    hp = np.max([np.fromfile(prefix + "_itag_3_has_particles_r%d.i32" % r, dtype=np.int32) for r in range(ranks)], axis=0)
    assert len(hp) == len(e2v)
    # the oracle's version of the loop (tests/test_gpu_parity.py::test_cpp_driver_pseudo_push_and_search)
    pop = common.population_box(s, n=n, num_ptcls=npt)
    mo, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH, C=64)
    common.set_shuffling(po)
    ppo.set_threads(ppo.max_threads())
    tag = np.full(len(e2v), -1, dtype=np.int32)
    se_, mk = po.slot_info()
    tag[np.unique(se_[mk.astype(bool)])] = 0
    expect, it = [], 1
    while it <= 30 and po.nPtcls() > 0:
        ppo.linear_push(po, 1.0 / 20, -0.5, 0.8, 0.0)
        r = ppo.search_mesh_legacy3d(mo, po, looplimit=100)
        ppo.update_positions(po)
        po.rebuild(r["elem_ids"])
        expect.append(po.nPtcls())
        if po.nPtcls() == 0:
            break
        se_, mk = po.slot_info()
        tag[np.unique(se_[mk.astype(bool)])] = it
        it += 1
    ppo.set_threads(1)
    assert counts == expect, (counts, expect)
    assert np.array_equal(hp, tag), int((hp != tag).sum())


# ---------------------------------------------------------------- the rows either side of the path: parts and balancer
def _cube_with_partition(pp, tmp_path, n, ranks):
    import importlib
    s = pp.synth
    coords, e2v, cls = s.kuhn_box(n)
    mesh_file = str(tmp_path / "cube.msh")
    importlib.import_module(pp.__name__ + ".meshio").write_gmsh(mesh_file, 3, coords, e2v, cls)
    ptn = str(tmp_path / "cube.ptn")
    with open(ptn, "w") as f:
        f.write("\n".join(str(int(e * ranks // len(e2v))) for e in range(len(e2v))) + "\n")
    return mesh_file, ptn


def test_reference_input_construct_passes(pp, tmp_path):
    """test/test_input_construct.cpp (testing.cmake: input_construct_cube, 4 ranks): PICparts from an Input with a FULL
    buffer and a 3-layer BFS safe zone over the sides, and with MINIMUM / NONE"""
    exe = _need("input_construct")
    mesh_file, ptn = _cube_with_partition(pp, tmp_path, 4, 4)
    res = _run_ranks([exe, mesh_file, ptn], 4, str(tmp_path), timeout=600)
    for r, (rc, so, se) in enumerate(res):
        assert rc == 0, "rank %d: %s\n%s" % (r, so[-2000:], se[-2000:])
    assert "All tests passed" in res[0][1]


@pytest.mark.parametrize("ranks", [1, 4])
def test_reference_test_lb_passes(pp, tmp_path, ranks):
    """test/test_lb.cpp (testing.cmake: lb_r1, lb_r4): ParticleBalancer::partition on an array of particle counts
    (imbalance after balancing <= 1.3) and ::repartition + migrate on a structure filled on even ranks only (<= 1.5)"""
    exe = _need("test_lb")
    mesh_file, ptn = _cube_with_partition(pp, tmp_path, 4, ranks)
    res = _run_ranks([exe, mesh_file, ptn if ranks > 1 else "ignored"], ranks, str(tmp_path), timeout=600)
    for r, (rc, so, se) in enumerate(res):
        assert rc == 0, "rank %d: %s\n%s" % (r, so[-2000:], se[-2000:])
    assert "All Tests Passed" in res[0][2]


@pytest.mark.parametrize("dim", [3, 2])
def test_reference_comm_array_passes(pp, tmp_path, dim):
    """test/test_comm_array.cpp (testing.cmake: comm_array_pisces, comm_array_2d_box; 4 ranks): reduceCommArray on parts
    with a full buffer (SUM = number of ranks on every entity of every dimension) and on parts with one buffer layer
    (MIN of the owners = entOwners, 1/n contributions sum to 1, a 3-component element array sums to 1).  The program
    reports by printing; no report may appear."""
    import importlib
    exe = _need("comm_array")
    s = pp.synth
    if dim == 3:
        coords, e2v, cls = s.kuhn_box(4)
    else:
        coords, e2v, cls = _plate_tris(12)
    mesh_file = str(tmp_path / "mesh.msh")
    importlib.import_module(pp.__name__ + ".meshio").write_gmsh(mesh_file, dim, coords, e2v, cls)
    ptn = str(tmp_path / "mesh.ptn")
    with open(ptn, "w") as f:
        f.write("\n".join(str(int(e * 4 // len(e2v))) for e in range(len(e2v))) + "\n")
    res = _run_ranks([exe, mesh_file, ptn], 4, str(tmp_path), timeout=600)
    for r, (rc, so, se) in enumerate(res):
        assert rc == 0, "rank %d: %s\n%s" % (r, so[-2000:], se[-2000:])
        assert "failed" not in so and "failed" not in se, "rank %d: %s\n%s" % (r, so[-2000:], se[-2000:])


@pytest.mark.parametrize("ranks,dim", [(2, 3), (4, 3), (4, 2)])
def test_reference_ptn_loading_passes(pp, tmp_path, ranks, dim):
    """test/test_ptn_loading.cpp (testing.cmake: ptn_loading_cube / _cube_4 / _2d_box_4 `<mesh> <ptn> 1 3`): parts with 3
    ghost layers and 1 safe layer; every element of a part has the centroid of the full-mesh element whose global id
    it carries"""
    import importlib
    exe = _need("ptn_loading")
    s = pp.synth
    coords, e2v, cls = s.kuhn_box(5) if dim == 3 else _plate_tris(16)
    mesh_file = str(tmp_path / "mesh.msh")
    importlib.import_module(pp.__name__ + ".meshio").write_gmsh(mesh_file, dim, coords, e2v, cls)
    ptn = str(tmp_path / "mesh.ptn")
    with open(ptn, "w") as f:
        f.write("\n".join(str(int(e * ranks // len(e2v))) for e in range(len(e2v))) + "\n")
    res = _run_ranks([exe, mesh_file, ptn, "1", "3"], ranks, str(tmp_path), timeout=600)
    for r, (rc, so, se) in enumerate(res):
        assert rc == 0 and "do not match" not in so, "rank %d: %s\n%s" % (r, so[-2000:], se[-2000:])


def test_reference_full_mesh_passes(pp, tmp_path):
    """test/test_full_mesh.cpp (testing.cmake: full_mesh_pisces, 4 ranks): an Input that reads its owners from a .ptn
    file, FULL buffer and safe zone: entity counts and global ids of the part are the serial mesh's"""
    exe = _need("full_mesh")
    mesh_file, ptn = _cube_with_partition(pp, tmp_path, 4, 4)
    res = _run_ranks([exe, mesh_file, ptn], 4, str(tmp_path), timeout=600)
    for r, (rc, so, se) in enumerate(res):
        assert rc == 0 and "do not match" not in so + se, "rank %d: %s\n%s" % (r, so[-2000:], se[-2000:])


@pytest.mark.parametrize("ranks,buffer,safe", [(4, "bfs", "full"), (4, "bfs", "bfs"), (1, "bfs", "full")])
def test_reference_file_rw_passes(pp, tmp_path, ranks, buffer, safe):
    """test/test_file.cpp (testing.cmake: file_rw_cube_4 `<mesh> <ptn> bfs full <prefix>`, file_rw_xgc_*): pumipic::write,
    then pumipic::read into a new mesh; isFullMesh, entity counts, buffered ranks, global ids, owners, rank-local and
    comm-array indices, offsets and the safe tag of the part read back must equal those of the part written (asserts,
    built with them).  The file is this library's own container (the full mesh, the owners, the rules), not .osh/.ppm."""
    exe = _need("file_rw")
    mesh_file, ptn = _cube_with_partition(pp, tmp_path, 4, ranks)
    res = _run_ranks([exe, mesh_file, ptn if ranks > 1 else "ignored", buffer, safe, str(tmp_path / "parts")], ranks,
                     str(tmp_path), timeout=600)
    for r, (rc, so, se) in enumerate(res):
        assert rc == 0, "rank %d: %s\n%s" % (r, so[-2000:], se[-2000:])
    assert "All Tests Passed" in res[0][1]
    assert os.path.exists(str(tmp_path / ("parts_%d.pparts" % ranks)))


@pytest.mark.parametrize("n", [16, 26, -224, 0])
def test_reference_test_adj_on_fine_meshes_complains_only_about_the_references_own_fallback(pp, ppo, tmp_path, n):
    """(n < 0: the plate of 2 n^2 = 100 352 triangles, where a segment through a vertex can fail both edges' tests within
    the tolerance and the particle stops where it is, adjacency.tpp:290-310.)
    On 24 576 and 105 456 tets, 10^6 rays, test_adj's wall-intersection check flags one or two particles per run: wall
    hits whose recorded point lies outside their face (and, rarely, a ray that ends without a face).  That is the
    reference's own algorithm -- when no face of an element passes the Moeller-Trumbore test, adjacency.tpp:343-352 keeps
    the point of the face with the best `closeness`, a face the ray does not cross.  Proof: every search_mesh call of the
    program is dumped (PP_SEARCH_DUMP, a hook of the mirror header) and the flagged particles, with 2 000 ordinary ones
    per call, are replayed through the oracle's restatement of that algorithm from the same origins, targets and seed
    elements: same element, same face, same point, bit for bit.  The program's complaints are exactly those particles."""
    import importlib.util
    exe = _need("test_adj")
    s = pp.synth
    # (n == 0: the 100 800-tet torus of configs[1-2] -- curved, non-convex, tets of many shapes; the program's bounding-box
    #  check assumes a box, so only the oracle comparison and the wall-point count are asserted there)
    dim, (c, e, cl) = (3, s.kuhn_box(n)) if n > 0 else (2, _plate_tris(-n)) if n < 0 else (3, s.torus_tet())
    mesh_file = str(tmp_path / "mesh.osh")
    s.write_mesh_bin(mesh_file, dim, c, e, cl)
    prefix = str(tmp_path / "sd")
    (rc, so, se), = _run_ranks([exe, mesh_file], 1, str(tmp_path), timeout=900, extra_env={"PP_SEARCH_DUMP": prefix})
    spec = importlib.util.spec_from_file_location("replay_search_dump", os.path.join(ROOT, "tools", "replay_search_dump.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.replay(mesh_file, prefix, ppo, verbose=False)
    # (two particle counts) x (internal, edge starts) x (two searches each), by barycentric walk and in intersection mode
    assert len(res) == 16 and sum(r["mode"] == "bcc" for r in res) == 8
    assert all(r["identical"] for r in res), res
    res = [r for r in res if r["mode"] == "intersection"]
    flagged = sum(r["off_face"] for r in res)
    # a handful per 10^6 rays, not a population (the plate, whose segments stop where an edge test fails: a few more)
    if n != 0:
        assert flagged <= (8 if dim == 3 else 50) and sum(r["lost"] for r in res) <= (4 if dim == 3 else 50), res
    print("flagged", [(r["call"], r["hits"], r["off_face"], r["lost"]) for r in res])
    assert so.count("outside the intersection face") + so.count("outside the intersection edge") == flagged, \
        (flagged, so[-2000:])
    if flagged == 0 and sum(r["lost"] for r in res) == 0:
        assert rc == 0 and "All Tests Passed" in se
