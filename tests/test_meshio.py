"""Gmsh ASCII .msh input (the mesh format the reference drivers read through Omega_h::gmsh::read,
test/pseudoXGCm.cpp:306-315): python reader/writer round trips and the C++ header reader."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CXX = r"""
#include "pumipic_gmsh.hpp"
int main(int argc, char** argv) {
  pumipic::gmsh::MeshData m;
  std::string err;
  if (!pumipic::gmsh::read(argv[1], m, &err)) { printf("ERROR %s\n", err.c_str()); return 1; }
  double cs = 0; for (size_t i = 0; i < m.coords.size(); ++i) cs += m.coords[i] * (double)(i % 7 + 1);
  long long es = 0; for (size_t i = 0; i < m.elem2verts.size(); ++i) es += (long long)m.elem2verts[i] * (long long)(i % 5 + 1);
  long long ks = 0; for (size_t i = 0; i < m.class_id.size(); ++i) ks += (long long)m.class_id[i] * (long long)(i % 3 + 1);
  long long ss = 0; for (size_t i = 0; i < m.side_class.size(); ++i) {
    long long vs = 0; for (int k = 0; k < m.dim; ++k) vs += m.side_verts[i * m.dim + k];
    ss += (long long)m.side_class[i] * 1000003ll + vs;
  }
  printf("dim %d nverts %zu nelems %zu coords %.17g e2v %lld cls %lld nsides %zu sides %lld\n", m.dim,
         m.coords.size() / m.dim, m.class_id.size(), cs, es, ks, m.side_class.size(), ss);
  return 0;
}
"""


@pytest.fixture(scope="module")
def reader_exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("gmsh")
    src = d / "rd.cpp"
    src.write_text(_CXX)
    exe = str(d / "rd")
    # (AddressSanitizer + UBSan on this CPU build: the parser reads files it did not write)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "pumi-pic_amd", "include"), str(src), "-o", exe])
    return exe


def _summary(dim, coords, e2v, cls):
    c = np.asarray(coords, dtype=np.float64).reshape(-1)
    cs = 0.0
    for i, v in enumerate(c):  # same left-to-right sum as the C++ checker
        cs += v * float(i % 7 + 1)
    e = np.asarray(e2v, dtype=np.int64).reshape(-1)
    k = np.asarray(cls, dtype=np.int64)
    return (dim, len(c) // dim, len(k), cs, int((e * (np.arange(len(e)) % 5 + 1)).sum()),
            int((k * (np.arange(len(k)) % 3 + 1)).sum()))


@pytest.mark.parametrize("version", ["2.2", "4.1"])
@pytest.mark.parametrize("dim", [2, 3])
def test_gmsh_round_trip(pp, reader_exe, tmp_path, dim, version):
    from pumipic_amd import meshio
    synth = pp.synth
    coords, e2v, cls = synth.annulus_tri(n_b=6, n_theta=16, band_width=2) if dim == 2 else synth.kuhn_box(3)
    path = str(tmp_path / "m.msh")
    meshio.write_gmsh(path, dim, coords, e2v, cls, version)
    d2, c2, e2, k2 = meshio.read_gmsh(path)
    assert d2 == dim and np.array_equal(c2, np.asarray(coords).reshape(-1, dim))
    order = np.arange(len(cls)) if version == "2.2" else np.argsort(cls, kind="stable")
    assert np.array_equal(e2, np.asarray(e2v)[order]) and np.array_equal(k2, np.asarray(cls)[order])
    out = subprocess.run([reader_exe, path], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    t = out.stdout.split()
    got = (int(t[1]), int(t[3]), int(t[5]), float(t[7]), int(t[9]), int(t[11]))
    assert got == _summary(d2, c2, e2, k2)


def test_gmsh_rejects_what_it_cannot_read(reader_exe, tmp_path):
    from pumipic_amd import meshio
    p = tmp_path / "bin.msh"
    p.write_text("$MeshFormat\n4.1 1 8\n$EndMeshFormat\n")
    with pytest.raises(ValueError):
        meshio.read_gmsh(str(p))
    out = subprocess.run([reader_exe, str(p)], capture_output=True, text=True)
    assert out.returncode == 1 and "binary" in out.stdout
    q = tmp_path / "lines.msh"  # only 1-D elements
    q.write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n2\n1 0 0 0\n2 1 0 0\n$EndNodes\n"
                 "$Elements\n1\n1 1 2 1 1 1 2\n$EndElements\n")
    with pytest.raises(ValueError):
        meshio.read_gmsh(str(q))
    assert subprocess.run([reader_exe, str(q)], capture_output=True, text=True).returncode == 1


@pytest.mark.parametrize("dim", [2, 3])
def test_gmsh_boundary_elements_become_side_classification(pp, reader_exe, tmp_path, dim):
    """boundary elements of a 2.2 file (lines of a triangle mesh, triangles of a tet mesh) come back from the C++ reader as
    (side vertices, elementary tag): what Omega_h::gmsh::read turns into the sides' class_id
    (test/pseudoPushAndSearch.cpp:231 picks its start elements by it)"""
    from pumipic_amd import meshio
    synth = pp.synth
    if dim == 3:
        coords, e2v, cls = synth.kuhn_box(2)
        faces = [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)]
        sides = [[t[a], t[b], t[c]] for t in e2v for a, b, c in faces
                 if (np.abs(coords[[t[a], t[b], t[c]], 1]) < 1e-12).all()]
    else:
        coords, e2v, cls = synth.plate_tri8_pardiag()
        sides = [[t[a], t[(a + 1) % 3]] for t in e2v for a in range(3)
                 if (np.abs(coords[[t[a], t[(a + 1) % 3]], 0]) < 1e-12).all()]
    tags = np.arange(len(sides)) % 3 + 150
    path = str(tmp_path / "m.msh")
    meshio.write_gmsh(path, dim, coords, e2v, cls, sides=(np.asarray(sides), tags))
    d2, c2, e2, k2 = meshio.read_gmsh(path)  # (the python reader keeps the top-dimensional elements only)
    assert d2 == dim and np.array_equal(e2, e2v) and np.array_equal(k2, cls)
    out = subprocess.run([reader_exe, path], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    t = out.stdout.split()
    assert int(t[13]) == len(sides) and len(sides) > 0
    assert int(t[15]) == int(sum(int(tag) * 1000003 + int(np.sum(sv)) for sv, tag in zip(sides, tags)))


def test_gmsh_reader_survives_truncated_and_garbled_files(pp, reader_exe, tmp_path):
    """the C++ reader under AddressSanitizer / UBSan on files cut at every tenth of their length and with a garbled
    element section: it reports an error or reads what is there -- no crash, no out-of-bounds access"""
    from pumipic_amd import meshio
    coords, e2v, cls = pp.synth.kuhn_box(2)
    path = str(tmp_path / "m.msh")
    meshio.write_gmsh(path, 3, coords, e2v, cls)
    text = open(path).read()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    for k in range(1, 10):
        cut = str(tmp_path / ("cut%d.msh" % k))
        open(cut, "w").write(text[:len(text) * k // 10])
        out = subprocess.run([reader_exe, cut], capture_output=True, text=True, env=env)
        assert out.returncode in (0, 1), (k, out.stderr[-1500:])
        assert "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, (k, out.stderr[-1500:])
    bad = str(tmp_path / "bad.msh")
    open(bad, "w").write(text.replace("$Elements\n", "$Elements\n999999999\n", 1))
    out = subprocess.run([reader_exe, bad], capture_output=True, text=True, env=env)
    assert out.returncode in (0, 1) and "AddressSanitizer" not in out.stderr, out.stderr[-1500:]
    ref = str(tmp_path / "ref.msh")  # an element that names a node the file does not hold
    open(ref, "w").write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n3\n1 0 0 0\n2 1 0 0\n3 0 1 0\n$EndNodes\n"
                         "$Elements\n1\n1 2 2 1 1 1 2 7\n$EndElements\n")
    out = subprocess.run([reader_exe, ref], capture_output=True, text=True, env=env)
    assert out.returncode == 1 and "unknown node" in out.stdout, (out.stdout, out.stderr[-800:])
