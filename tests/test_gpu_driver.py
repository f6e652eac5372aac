"""The drop-in driver path against the oracle in its REFERENCE mode (libm trig).

drivers/pseudoXGCm is the reference's test/pseudoXGCm.cpp:504-534 step loop on the particle_structs mirror: the
push is the reference's USER lambda through ps::parallel_for (device libm sin/cos, test/ellipticalPush.hpp:36-70),
then search_mesh_2d, the updatePtclPositions lambda, migrate_lb_ptcls (one rank: rebuild) and two gyroScatter
calls -- the unfused path a pseudoXGCm user gets without touching the source.  The driver dumps its structure
after the set-up and after the last step (PP_DRIVER_DUMP); the oracle replays the run from the initial dump with
`trig=0` (glibc sin/cos: what the reference's Kokkos::Serial build calls) and the two final states are compared
particle by particle, matched by particle id."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "pumi-pic_amd", "drivers")


@pytest.fixture(scope="module")
def capi(pp):
    from pumipic_amd import capi as c
    c.init(0)
    return c


def _load(prefix, tag):
    base = "%s_r0_%s" % (prefix, tag)
    cap, stride, nptcls = (int(v) for v in open(base + "_meta.txt").read().split())
    mask = np.fromfile(base + "_mask.u8", dtype=np.uint8)[:cap].astype(bool)
    x = np.fromfile(base + "_x.f64", dtype=np.float64)
    return dict(cap=cap, n=nptcls, mask=mask,
                elem=np.fromfile(base + "_elem.i32", dtype=np.int32)[:cap],
                x=np.stack([x[c * stride:c * stride + cap] for c in range(3)]),
                id=np.fromfile(base + "_id.i32", dtype=np.int32)[:cap],
                b=np.fromfile(base + "_b.f32", dtype=np.float32)[:cap],
                phi=np.fromfile(base + "_phi.f32", dtype=np.float32)[:cap])


def _edge_distance(coords, e2v, elem, xy):
    """smallest barycentric coordinate of the points in their triangles (numpy, independent of both sides)"""
    v = coords[e2v[elem]]  # [n, 3, 2]
    a, b, c = v[:, 0], v[:, 1], v[:, 2]
    det = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (c[:, 0] - a[:, 0]) * (b[:, 1] - a[:, 1])
    l1 = ((b[:, 0] - xy[:, 0]) * (c[:, 1] - xy[:, 1]) - (c[:, 0] - xy[:, 0]) * (b[:, 1] - xy[:, 1])) / det
    l2 = ((c[:, 0] - xy[:, 0]) * (a[:, 1] - xy[:, 1]) - (a[:, 0] - xy[:, 0]) * (c[:, 1] - xy[:, 1])) / det
    return np.minimum(np.minimum(l1, l2), 1 - l1 - l2)


@pytest.mark.parametrize("nptcl,steps", [(2_000_000, 12), (10_000_000, 6)])
def test_driver_pseudoxgcm_equals_oracle_libm(pp, ppo, capi, tmp_path, nptcl, steps):
    s = pp.synth
    subprocess.check_call(["make", "-C", DRV, "-s"])
    coords, e2v, cls = s.annulus_tri()  # the 2-D literal of BASELINE configs[2]: 100 352 triangles
    mesh_file = str(tmp_path / "annulus.bin")
    s.write_mesh_bin(mesh_file, 2, coords, e2v, cls)
    prefix = str(tmp_path / "dump")
    deg, mdl = 0.5, 12
    out = subprocess.run([os.path.join(DRV, "pseudoXGCm"), mesh_file, str(nptcl), str(mdl), str(steps), str(deg), "0"],
                         env=dict(os.environ, PP_DRIVER_DUMP=prefix), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"RESULT particles (\d+) scatter_mass (\S+)", out.stdout)
    assert m, out.stdout[-2000:]
    ini, fin = _load(prefix, "initial"), _load(prefix, "final")
    assert ini["n"] == nptcl and int(ini["mask"].sum()) == nptcl

    # ---- the oracle from the driver's initial population: its own set-up and `steps` steps, libm trig
    live = np.flatnonzero(ini["mask"])
    elem0 = ini["elem"][live]
    order = np.argsort(elem0, kind="stable")  # element-major, slot order inside an element
    live, elem0 = live[order], elem0[order]
    ne = len(e2v)
    ppe = np.bincount(elem0, minlength=ne).astype(np.int32)
    xyz = np.ascontiguousarray(ini["x"][:, live])
    info = [xyz, np.zeros_like(xyz), np.ascontiguousarray(ini["id"][live]), np.zeros(nptcl, np.float32),
            np.zeros(nptcl, np.float32)]
    mo = ppo.Mesh(2, coords, e2v, cls)
    po = ppo.PS.scs(ppo.PARTICLE_XGCM, ne, ppe, C_max=64, sigma=2**31 - 1, V=1024, pad_strat=0, shuffle_padding=0.1,
                    extra_padding=0.0, particle_elements=elem0, particle_info=info)
    h, k, d = 1.72479370 - .08, .020558260, 0.6  # the driver's literals (test/pseudoXGCm.cpp:470-472)
    ppo.set_threads(ppo.max_threads())
    try:
        ppo.elliptical_setup(po, h, k, d)
        # the set-up (atan2 / sin of the DEVICE libm in the driver, glibc here): b and phi are float32 state
        so, mko = po.slot_info()
        capo = po.capacity()
        io, b_o = pp_by_id(po.member(2)[0, :capo], mko, po.member(3)[0, :capo])
        _, phi_o = pp_by_id(po.member(2)[0, :capo], mko, po.member(4)[0, :capo])
        ig, b_g = pp_by_id(ini["id"], ini["mask"], ini["b"])
        _, phi_g = pp_by_id(ini["id"], ini["mask"], ini["phi"])
        assert np.array_equal(io, ig)
        nb, nphi = int((b_o != b_g).sum()), int((phi_o != phi_g).sum())
        print("set-up: %d of %d b values and %d phi values differ in their float32 rounding (device libm vs glibc)"
              % (nb, nptcl, nphi))
        assert nb <= nptcl // 10000 and nphi <= nptcl // 10000
        if nb or nphi:  # never more than one float32 ulp
            assert np.all(np.abs(b_o - b_g) <= np.spacing(np.abs(b_o)))
            assert np.all(np.abs(phi_o - phi_g) <= np.spacing(np.abs(phi_o)))
        # ring maps: the library places the ring points with the sincos it shares with the oracle (trig=1, bit-equal
        # maps: tests/test_gpu_parity.py); the reference's libm placement may put a point that lies on an edge into
        # the neighbouring triangle -- counted here, the fields are compared on the library's map
        fo, bo = ppo.create_gyro_ring_mappings(mo, trig=1)
        f0, _ = ppo.create_gyro_ring_mappings(mo, trig=0)
        nmap = int((f0.reshape(-1, 3) != fo.reshape(-1, 3)).any(axis=1).sum())
        print("ring maps: %d of %d ring points map to another triangle under libm" % (nmap, len(fo) // 3))
        assert nmap <= 8
        for _ in range(steps):
            ppo.elliptical_push(po, mo, h, k, d, deg, trig=0)
            found, ids_o, _ = ppo.search_mesh_2d(mo, po, looplimit=200)
            assert found
            ppo.update_positions(po)
            po.rebuild(ids_o)
        w_f, w_b = ppo.gyro_scatter(mo, po, fo), ppo.gyro_scatter(mo, po, bo)
    finally:
        ppo.set_threads(1)

    # ---- final states, particle by particle
    assert po.nPtcls() == fin["n"] == int(m.group(1))
    so, mko = po.slot_info()
    capo = po.capacity()
    ido = po.member(2)[0, :capo]
    io, eo = pp_by_id(ido, mko, so[:capo])
    ig, eg = pp_by_id(fin["id"], fin["mask"], fin["elem"])
    assert np.array_equal(io, ig), "the two runs hold different particles"
    differ = np.flatnonzero(eo != eg)
    _, xo = pp_by_id(ido, mko, po.member(0)[:, :capo])
    _, xg = pp_by_id(fin["id"], fin["mask"], fin["x"])
    print("final: %d of %d particles in a different element than the oracle(libm) run" % (len(differ), len(io)))
    # north_star: positions within 1e-12 relative
    rel = np.abs(xo[:2] - xg[:2]) / np.maximum(np.abs(xo[:2]).max(axis=0, keepdims=True), 1e-300)
    assert rel.max() <= 1e-12, rel.max()
    assert np.array_equal(xo[2], xg[2])
    if len(differ):
        # a particle may only sit in another element than the oracle's when both positions lie within rounding
        # distance of the common edge: its smallest barycentric coordinate in EITHER element is ~0
        dg = _edge_distance(coords, e2v, eg[differ], xg[:2, differ].T)
        do = _edge_distance(coords, e2v, eo[differ], xo[:2, differ].T)
        assert np.all(np.abs(dg) < 1e-9) and np.all(np.abs(do) < 1e-9), (dg, do)
        assert len(differ) <= 4, "more element differences than edge coincidences can explain: %d" % len(differ)
    _, phio = pp_by_id(ido, mko, po.member(4)[0, :capo])
    _, phig = pp_by_id(fin["id"], fin["mask"], fin["phi"])
    assert int((phio != phig).sum()) <= nphi  # the push adds the same double increment on both sides
    # the scatter fields are sums of exact multiples of 1/8 over the per-element counts: equal when the
    # elements are; with k particles in other elements at most 2 * 9 * k vertices see another value
    base = "%s_r0_final" % prefix
    g_f, g_b = np.fromfile(base + "_fwd.f64"), np.fromfile(base + "_bkwd.f64")
    assert np.array_equal(g_f, g_b)
    if len(differ) == 0:
        assert np.array_equal(w_f, g_f) and np.array_equal(w_b, g_b)
    else:
        assert int((w_f != g_f).sum()) <= 64 * len(differ)
    assert abs(float(g_f.sum()) - float(m.group(2))) <= 1e-6 * float(g_f.sum())


def pp_by_id(ids, mask, values):
    live = np.flatnonzero(mask)
    order = np.argsort(ids[live], kind="stable")
    return ids[live][order], np.asarray(values)[..., live[order]]


def test_cpp_boundary_api(capi, tmp_path):
    """CSR_Input + CSR(Input_T&), ParticleStructure::getPIDs, printFormat and ps::copy<HostSpace> of the mirror headers
    (tests/cpp/boundary_api.cpp), compiled against the library and run on the GPU; the text of printFormat is
    checked here against the populations the program builds."""
    exe = str(tmp_path / "boundary_api")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip",
                           "-ffp-contract=off", "-Wno-unused-result", os.path.join(ROOT, "tests", "cpp", "boundary_api.cpp"),
                           "-o", exe, "-L" + os.path.join(ROOT, "pumi-pic_amd"), "-lpumipic_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "pumi-pic_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])
    assert "BOUNDARY OK" in out.stdout and "FAILED" not in out.stdout
    text = out.stdout
    scs = text[text.index("FORMAT scs"):text.index("FORMAT csr")]
    csr = text[text.index("FORMAT csr"):]
    m = re.search(r"Particle Structures Sell-C-Sigma C: (\d+) sigma: (\d+) V: (\d+)\.\nNumber of Elements: (\d+)\.\n"
                  r"Number of Particles: (\d+)\.\nNumber of Chunks: (\d+)\.\nNumber of Slices: (\d+)\.", scs)
    assert m, scs[:400]
    C_, ne, npt, nchunks, nslices = int(m.group(1)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7))
    assert ne == 150 and nchunks == -(-ne // C_)
    chunk_lines = re.findall(r"^  Chunk (\d+)\. Elements\(GID\):(.*)$", scs, flags=re.M)
    assert [int(c) for c, _ in chunk_lines] == list(range(nchunks))
    elems = []
    for _, body in chunk_lines:
        row = re.findall(r"(\d+)\((-?\d+)\)", body)
        assert len(row) == C_
        for e, g in row:
            e, g = int(e), int(g)
            assert g == (3 * e + 5 if e < ne else -1)  # the gids the program built; padding rows carry none
            elems.append(e)
    assert sorted(elems) == list(range(nchunks * C_))  # every row appears once
    slice_lines = re.findall(r"^    Slice (\d+)((?: \|(?: [01])+)+)$", scs, flags=re.M)
    assert [int(s) for s, _ in slice_lines] == list(range(nslices))
    assert sum(body.count("1") for _, body in slice_lines) == npt  # one `1` per live particle
    for _, body in slice_lines:
        for col in body.split("|")[1:]:
            assert len(col.split()) == C_  # a column of the slice is C slots
    m = re.search(r"Particle Structures CSR\nNumber of Elements: (\d+)\.\nNumber of Particles: (\d+)\.", csr)
    assert m and int(m.group(1)) == ne and int(m.group(2)) == npt
    rows = re.findall(r"^  Element +(\d+)\( *(\d+)\) \|((?: 1)+)$", csr, flags=re.M)
    assert sum(body.count("1") for _, _, body in rows) == npt
    assert all(int(g) == 3 * int(e) + 5 for e, g, _ in rows)
    assert "Element  3(" not in csr  # element 3 holds no particle: no line (CSR.hpp:249)


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("looplimit,deg", [(200, 2.0), (1, 25.0)])
def test_found_rides_with_the_rebuild_totals(pp, ppo, capi, dim, looplimit, deg):
    """pp_ps_last_search_found: the `found` of a pp_push_search that was called WITHOUT a found pointer (no host
    wait in the step) comes to the host with the totals of the rebuild that follows, and equals what the same
    search returns when it is asked directly -- both when every particle is found and when the loop limit cuts
    particles off (found == false, the reference then deletes them)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    s = pp.synth
    pop = (common.population_2d(s, n_b=16, n_theta=64, num_ptcls=40000, mdl_face=4) if dim == 2 else
           common.population_3d(s, n_b=8, n_theta=24, n_planes=8, num_ptcls=40000, mdl_face=6))
    answers = []
    for direct in (True, False):
        mg, pg = common.gpu_pair(capi, pop, capi.PARTICLE_XGCM)
        cap = pg.capacity()
        ids = capi.DevArray.from_host(np.full(cap + cap // 10, -1, dtype=np.int32))
        fwd, bkwd = capi.create_gyro_ring_mappings(mg)
        for step in range(3):  # (the second and third step run the record-fed kernels)
            f = capi.push_search(mg, pg, s.XGC_H, s.XGC_K, s.XGC_D, deg, ids, seeded=False, looplimit=looplimit,
                                 want_found=direct)
            capi.rebuild_scatter(pg, mg, ids, [fwd, bkwd])
            if not direct:
                f = capi.last_search_found(pg)
            answers.append((direct, step, f, pg.nPtcls()))
            cap = pg.capacity()
            if cap > ids.n:
                ids = capi.DevArray.from_host(np.full(cap + cap // 10, -1, dtype=np.int32))
    a = [x[2:] for x in answers if x[0]]
    b = [x[2:] for x in answers if not x[0]]
    assert a == b, answers
    if looplimit == 1:
        assert not a[0][0] and a[0][1] < 40000  # particles were cut off and deleted
    else:
        assert all(f for f, _ in a)
