"""Particle fixture files (.ptl) of the reference's structure tests
(particle_structs/test/read_particles.hpp:8-118; written by write_particle_file.cpp):

    <num_elems> <num_ptcls>
    <elem gid> <nppe>                                   for each element
    <particle_elem> <id> <v0> <v1> <v2> <short> <int>   for each particle

for the test particle type MemberTypes<int, double[3], short, int> (test_types.hpp:12).  Mirrors
pumi-pic_amd/include/pumipic_ptl.hpp; the result is what capi.PS.scs / PS.csr (and the oracle) take."""
import numpy as np

TEST_TYPES = [(np.int32, 1), (np.float64, 3), (np.int16, 1), (np.int32, 1)]


def read_ptl(path):
    with open(path) as f:
        tok = f.read().split()
    ne, npt = int(tok[0]), int(tok[1])
    pos = 2
    e = np.array(tok[pos:pos + 2 * ne], dtype=np.int64).reshape(ne, 2)
    pos += 2 * ne
    if len(tok) < pos + 7 * npt:
        raise ValueError("truncated particle file " + path)
    rows = np.array(tok[pos:pos + 7 * npt], dtype=object).reshape(npt, 7)
    elem = rows[:, 0].astype(np.int32)
    ids = rows[:, 1].astype(np.int32)
    vals1 = np.ascontiguousarray(rows[:, 2:5].astype(np.float64).T)
    vals2 = rows[:, 5].astype(np.int16)
    vals3 = rows[:, 6].astype(np.int32)
    return dict(num_elems=ne, num_ptcls=npt, gids=e[:, 0].copy(), ppe=e[:, 1].astype(np.int32), elem=elem,
                info=[ids, vals1, vals2, vals3])


def write_ptl(path, gids, ppe, elem, info):
    ids, vals1, vals2, vals3 = info
    vals1 = np.asarray(vals1).reshape(3, -1)
    with open(path, "w") as f:
        f.write("%d %d\n" % (len(ppe), len(elem)))
        for g, n in zip(gids, ppe):
            f.write("%d %d\n" % (g, n))
        f.write("\n")
        for i in range(len(elem)):
            f.write("%d %d %s %d %d\n" % (elem[i], ids[i], " ".join(repr(float(v)) for v in vals1[:, i]), vals2[i],
                                          vals3[i]))
