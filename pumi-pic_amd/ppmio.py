"""The per-rank `.ppm` files of the reference's PICpart writer (pumipic::write / pumipic::read,
src/pumipic_file.cpp:44-204): `<path>_<nranks>.ppm/<prefix>_<rank>.ppm`, the comm-array bookkeeping of one
rank's part.  Field order (file.cpp:80-115, read back :152-181):

    I8  version (= 2)            I8  is_full_mesh
    for dimension 0..3:
        I64 num_entites          I32 num_cores
        LO[] buffered_parts      LO[] offset_ents_per_rank_per_dim     LO[] ent_to_comm_arr_index_per_dim
        LO[] is_complete_part    I32 num_bounds     I32 num_boundaries
        LO[] boundary_parts      LO[] offset_bounded_per_dim           LO[] bounded_ent_ids

Scalars and arrays are framed by Omega_h's binary::write_value / write_array (Omega_h_file.cpp of
SCOREC/omega_h scorec-v10.8.x -- NOT in the reference tree, restated from its published source): a value is
its raw little-endian bytes; an array is an I32 element count followed, in a zlib build (OMEGA_H_USE_ZLIB, the
compile-time switch file.cpp:76-80 reads), by an I64 compressed byte count and one zlib stream
(compress2, Z_BEST_SPEED), else by the raw elements.  The file itself does not say which; with `compress=None` the
reader decides on the file's first array (a positive check: the compressed size fits, the stream inflates to
4 x count bytes) and then requires the whole file to parse in that framing.

PARITY UNPINNED: the reference's data submodule (pumipic-data) is empty here, so no `.ppm` written by the
reference was available; what is checked is the round trip and that the fields written for a part equal the
oracle's (tests/test_ppmio.py).  The mesh of a part, which the reference stores next to the `.ppm` as an
Omega_h `.osh` directory (binary::write(mesh_file, picparts.mesh())), is NOT read or written: that
container's layout (tags, adjacency codes, class sets, version history) lives only in Omega_h's sources;
write_picpart() stores the part's mesh as Gmsh 2.2 (`<prefix>_<rank>.msh`, pumi-pic_amd/meshio.py) instead."""
import os
import struct
import zlib

import numpy as np

VERSION = 2
_ARRAYS = ("buffered_parts", "offset_ents_per_rank", "ent_to_comm_arr_index", "is_complete_part")
_ARRAYS2 = ("boundary_parts", "offset_bounded", "bounded_ent_ids")


def _w_array(out, a, compress):
    a = np.ascontiguousarray(a, dtype="<i4")
    out.append(struct.pack("<i", len(a)))
    raw = a.tobytes()
    if compress:
        z = zlib.compress(raw, 1)  # Z_BEST_SPEED
        out.append(struct.pack("<q", len(z)))
        out.append(z)
    else:
        out.append(raw)


def dumps(part, compress=True):
    """part: {"version", "is_full_mesh", "dims": [4 dicts]} -> bytes"""
    out = [struct.pack("<bb", int(part.get("version", VERSION)), int(bool(part["is_full_mesh"])))]
    for d in part["dims"]:
        out.append(struct.pack("<q", int(d["num_entites"])))
        out.append(struct.pack("<i", int(d["num_cores"])))
        for k in _ARRAYS:
            _w_array(out, d[k], compress)
        out.append(struct.pack("<i", int(d["num_bounds"])))
        out.append(struct.pack("<i", int(d["num_boundaries"])))
        for k in _ARRAYS2:
            _w_array(out, d[k], compress)
    return b"".join(out)


class _Reader:
    def __init__(self, buf, compress):
        self.b, self.p, self.compress = buf, 0, compress

    def value(self, fmt):
        n = struct.calcsize(fmt)
        if self.p + n > len(self.b):
            raise ValueError("truncated .ppm file")
        v = struct.unpack_from(fmt, self.b, self.p)[0]
        self.p += n
        return v

    def array(self):
        n = self.value("<i")
        if n < 0:
            raise ValueError("negative array length in .ppm file")
        if self.compress:
            nz = self.value("<q")
            if nz < 0 or self.p + nz > len(self.b):
                raise ValueError("bad compressed length in .ppm file")
            raw = zlib.decompress(self.b[self.p:self.p + nz])
            self.p += nz
            if len(raw) != 4 * n:
                raise ValueError("array length mismatch in .ppm file")
        else:
            if self.p + 4 * n > len(self.b):
                raise ValueError("truncated .ppm file")
            raw = self.b[self.p:self.p + 4 * n]
            self.p += 4 * n
        return np.frombuffer(raw, dtype="<i4").astype(np.int32)


class _NotZlibFramed(ValueError):
    pass


def _first_array_is_zlib(buf):
    """Positive check of the framing on the FIRST array of the file (offset 2 + 8 + 4: version, is_full_mesh,
    num_entites, num_cores; version 1 has no num_entites): zlib framing = I32 count, I64 compressed size that fits
    the file, a stream that inflates to exactly 4 * count bytes.  The whole file then has to parse in the framing
    detected here -- a file that starts in one framing and continues in the other is an error, not a guess."""
    ver = struct.unpack_from("<b", buf, 0)[0]
    p = 2 + (8 if ver >= 2 else 0) + 4
    if p + 4 > len(buf):
        raise ValueError("truncated .ppm file")
    n = struct.unpack_from("<i", buf, p)[0]
    if n < 0 or p + 12 > len(buf):
        return False
    nz = struct.unpack_from("<q", buf, p + 4)[0]
    if nz < 0 or p + 12 + nz > len(buf):
        return False
    try:
        return len(zlib.decompress(buf[p + 12:p + 12 + nz])) == 4 * n
    except zlib.error:
        return False


def loads(buf, compress=None):
    if compress is None:  # the file does not say which framing it uses: decide on its first array, then insist
        compress = _first_array_is_zlib(buf)
    r = _Reader(buf, compress)
    part = {"version": r.value("<b"), "is_full_mesh": bool(r.value("<b")), "dims": []}
    if part["version"] not in (1, 2):
        raise ValueError("unknown .ppm version %d" % part["version"])
    for _ in range(4):
        d = {"num_entites": r.value("<q") if part["version"] >= 2 else -1, "num_cores": r.value("<i")}
        for k in _ARRAYS:
            d[k] = r.array()
        d["num_bounds"] = r.value("<i")
        d["num_boundaries"] = r.value("<i")
        for k in _ARRAYS2:
            d[k] = r.array()
        part["dims"].append(d)
    if r.p != len(buf):
        raise ValueError("trailing bytes in .ppm file")
    return part


def file_names(path, nranks, rank):
    """(directory, .ppm file, mesh file stem) as pumipic::write names them (file.cpp:47-66)"""
    prefix = os.path.basename(path)
    d = "%s_%d.ppm" % (path, nranks)
    return d, os.path.join(d, "%s_%d.ppm" % (prefix, rank)), os.path.join(d, "%s_%d" % (prefix, rank))


def _empty_dim():
    z = np.zeros(0, dtype=np.int32)
    return dict(num_entites=0, num_cores=0, buffered_parts=z, offset_ents_per_rank=z, ent_to_comm_arr_index=z,
                is_complete_part=z, num_bounds=0, num_boundaries=0, boundary_parts=z, offset_bounded=z,
                bounded_ent_ids=z)


def fields_of_picpart(pic):
    """the .ppm contents of a capi.PicPart (pumi-pic_amd/capi.py), every dimension 0..dim of its mesh"""
    dim = pic.dim
    ncore_elems = len(pic.buffered_ranks(dim))
    dims = []
    for d in range(4):
        if d > dim:
            dims.append(_empty_dim())
            continue
        buf = pic.buffered_ranks(d)
        bparts, boff, bids = pic.bounded(d)
        dims.append(dict(num_entites=pic.num_global(d), num_cores=len(buf), buffered_parts=buf,
                         offset_ents_per_rank=pic.nents_offsets(d), ent_to_comm_arr_index=pic.array(3, d),
                         is_complete_part=pic.complete_parts(d), num_bounds=len(buf) - ncore_elems,
                         num_boundaries=len(bparts), boundary_parts=bparts, offset_bounded=boff, bounded_ent_ids=bids))
    return {"version": VERSION, "is_full_mesh": pic.is_full_mesh, "dims": dims}


def write_picpart(pic, path, rank, nranks, compress=True, mesh_writer=None):
    """pumipic::write(picparts, path) for one rank.  mesh_writer(stem): stores the part's mesh (default: none)."""
    d, ppm, stem = file_names(path, nranks, rank)
    os.makedirs(d, exist_ok=True)
    with open(ppm, "wb") as f:
        f.write(dumps(fields_of_picpart(pic), compress))
    if mesh_writer is not None:
        mesh_writer(stem)
    return ppm


def read_ppm(path, nranks, rank, compress=None):
    """pumipic::read's `.ppm` half for one rank -> dict (see loads)"""
    _, ppm, _ = file_names(path, nranks, rank)
    with open(ppm, "rb") as f:
        return loads(f.read(), compress)
