"""Gmsh ASCII .msh reader / writer (formats 2.2 and 4.1) for the host side of the boundary.

The reference drivers read their meshes with Omega_h::gmsh::read (test/pseudoXGCm.cpp:306-315);
Omega_h is not part of the reference tree, so this follows the published MSH format.  The reader
returns what `pp_mesh_create` takes -- (dim, coords[nverts, dim], elem2verts[nelems, dim+1],
class_id[nelems]) with class_id = the element's elementary entity tag -- and mirrors
pumi-pic_amd/include/pumipic_gmsh.hpp line for line.  The writer exists for tests and for
exporting the synthetic meshes.
"""
import numpy as np

_NVERTS = {1: 2, 2: 3, 3: 4, 4: 4, 5: 8, 6: 6, 7: 5, 8: 3, 9: 6, 10: 9, 11: 10, 15: 1}


def read_gmsh(path):
    with open(path) as f:
        tok = f.read().split("\n")
    version = 0.0
    node_index, xyz, elems = {}, [], []
    i = 0
    while i < len(tok):
        line = tok[i].strip()
        i += 1
        if line.startswith("$MeshFormat"):
            v, ftype, _ = tok[i].split()[:3]
            i += 1
            version = float(v)
            if int(ftype) != 0:
                raise ValueError("binary .msh files are not supported")
            if not (2.0 <= version < 3.0 or 4.0 <= version < 5.0):
                raise ValueError("unsupported .msh version %s" % v)
        elif line.startswith("$Nodes"):
            if version < 3.0:
                n = int(tok[i]); i += 1
                for _ in range(n):
                    p = tok[i].split(); i += 1
                    node_index[int(p[0])] = len(xyz)
                    xyz.append([float(p[1]), float(p[2]), float(p[3])])
            else:
                nblocks = int(tok[i].split()[0]); i += 1
                for _ in range(nblocks):
                    edim, _, parametric, nb = (int(t) for t in tok[i].split()); i += 1
                    tags = [int(tok[i + k]) for k in range(nb)]
                    i += nb
                    for k in range(nb):
                        p = tok[i].split(); i += 1
                        node_index[tags[k]] = len(xyz)
                        xyz.append([float(p[0]), float(p[1]), float(p[2])])
        elif line.startswith("$Elements"):
            if version < 3.0:
                n = int(tok[i]); i += 1
                for _ in range(n):
                    p = [int(t) for t in tok[i].split()]; i += 1
                    etype, ntags = p[1], p[2]
                    elementary = p[4] if ntags >= 2 else 0
                    if etype in (2, 4):
                        elems.append((etype, elementary, p[3 + ntags:3 + ntags + _NVERTS[etype]]))
            else:
                nblocks = int(tok[i].split()[0]); i += 1
                for _ in range(nblocks):
                    _, etag, etype, nb = (int(t) for t in tok[i].split()); i += 1
                    if etype not in _NVERTS:
                        raise ValueError("unsupported element type %d" % etype)
                    for _ in range(nb):
                        p = [int(t) for t in tok[i].split()]; i += 1
                        if etype in (2, 4):
                            elems.append((etype, etag, p[1:1 + _NVERTS[etype]]))
    if version == 0.0:
        raise ValueError("no $MeshFormat section")
    dim = 3 if any(e[0] == 4 for e in elems) else 2
    want = 4 if dim == 3 else 2
    coords = np.array(xyz, dtype=np.float64).reshape(-1, 3)[:, :dim].copy()
    keep = [e for e in elems if e[0] == want]
    if not keep:
        raise ValueError("no triangles or tetrahedra in %s" % path)
    e2v = np.array([[node_index[v] for v in e[2]] for e in keep], dtype=np.int32)
    cls = np.array([e[1] for e in keep], dtype=np.int32)
    return dim, coords, e2v, cls


def write_gmsh(path, dim, coords, e2v, cls, version="2.2", sides=None):
    """ASCII writer: node tags are 1-based vertex ids, the class id goes to the physical and the
    elementary tag (2.2) / to the entity tag of a block per class (4.1).  `sides` = (side vertices [n, dim], tags [n]):
    boundary entities (triangles of a tet mesh, lines of a triangle mesh) written in front of the elements (2.2 only),
    which a reader turns into the classification of the mesh sides."""
    coords = np.asarray(coords, dtype=np.float64).reshape(-1, dim)
    e2v = np.asarray(e2v, dtype=np.int64).reshape(-1, dim + 1)
    cls = np.asarray(cls, dtype=np.int64)
    etype = 2 if dim == 2 else 4
    xyz = np.zeros((len(coords), 3))
    xyz[:, :dim] = coords
    with open(path, "w") as f:
        if version.startswith("2"):
            f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % len(xyz))
            for i, p in enumerate(xyz):
                f.write("%d %s %s %s\n" % (i + 1, repr(float(p[0])), repr(float(p[1])), repr(float(p[2]))))
            ns = 0 if sides is None else len(sides[1])
            f.write("$EndNodes\n$Elements\n%d\n" % (len(e2v) + ns))
            for i in range(ns):
                c = int(sides[1][i])
                f.write("%d %d 2 %d %d %s\n" % (i + 1, 2 if dim == 3 else 1, c, c,
                                               " ".join(str(int(v) + 1) for v in sides[0][i])))
            for i, (vs, c) in enumerate(zip(e2v, cls)):
                f.write("%d %d 2 %d %d %s\n" % (ns + i + 1, etype, c, c, " ".join(str(v + 1) for v in vs)))
            f.write("$EndElements\n")
        else:
            n = len(xyz)
            f.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Nodes\n1 %d 1 %d\n%d 1 0 %d\n" % (n, n, dim, n))
            for i in range(n):
                f.write("%d\n" % (i + 1))
            for p in xyz:
                f.write("%s %s %s\n" % (repr(float(p[0])), repr(float(p[1])), repr(float(p[2]))))
            classes = np.unique(cls)
            f.write("$EndNodes\n$Elements\n%d %d 1 %d\n" % (len(classes), len(e2v), len(e2v)))
            eid = 0
            for c in classes:  # a block per class keeps file order inside a class only
                idx = np.flatnonzero(cls == c)
                f.write("%d %d %d %d\n" % (dim, c, etype, len(idx)))
                for j in idx:
                    eid += 1
                    f.write("%d %s\n" % (eid, " ".join(str(v + 1) for v in e2v[j])))
            f.write("$EndElements\n")
