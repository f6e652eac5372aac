"""The `.ppm` files of pumipic::write / pumipic::read (src/pumipic_file.cpp:44-204): framing round trips,
error handling, and -- field by field -- what a part writes against the oracle's restatement of
constructPICPart / setupComm (oracle/ppo_picpart.py).  PARITY UNPINNED against files written by the
reference itself (pumipic-data is empty in this tree): pumi-pic_amd/ppmio.py says so."""
import os

import numpy as np
import pytest

import pumipic_amd_loader
from test_picpart_oracle import slab_owners


@pytest.fixture(scope="module")
def ppmio(pp):
    return pp.ppmio


@pytest.fixture(scope="module")
def opp():
    return pumipic_amd_loader.load_oracle_picpart()


def _random_part(rng, nranks=5):
    dims = []
    for d in range(4):
        nb = int(rng.integers(0, nranks))
        n = int(rng.integers(0, 400))
        nbd = int(rng.integers(0, nranks))
        dims.append(dict(num_entites=int(rng.integers(0, 2**40)), num_cores=nb,
                         buffered_parts=rng.integers(0, nranks, nb).astype(np.int32),
                         offset_ents_per_rank=np.sort(rng.integers(0, n + 1, nranks + 1)).astype(np.int32),
                         ent_to_comm_arr_index=rng.permutation(n).astype(np.int32),
                         is_complete_part=rng.integers(0, 3, nranks).astype(np.int32), num_bounds=int(rng.integers(0, 4)),
                         num_boundaries=nbd, boundary_parts=rng.integers(0, nranks, nbd).astype(np.int32),
                         offset_bounded=np.sort(rng.integers(0, 50, nbd + 1)).astype(np.int32),
                         bounded_ent_ids=rng.integers(0, 1000, 50).astype(np.int32)))
    return {"version": 2, "is_full_mesh": bool(rng.integers(0, 2)), "dims": dims}


def _same(a, b):
    assert a["version"] == b["version"] and a["is_full_mesh"] == b["is_full_mesh"]
    for x, y in zip(a["dims"], b["dims"]):
        assert set(x) == set(y)
        for k in x:
            assert np.array_equal(x[k], y[k]), k


@pytest.mark.parametrize("compress", [True, False])
def test_round_trip_and_framing(ppmio, compress):
    rng = np.random.default_rng(7)
    for _ in range(10):
        part = _random_part(rng)
        buf = ppmio.dumps(part, compress)
        _same(ppmio.loads(buf, compress), part)
        _same(ppmio.loads(buf), part)                      # the framing is recognised without being told
        # header: two I8, then the first dimension's I64 entity count and I32 core count
        assert buf[0] == 2 and buf[1] == int(part["is_full_mesh"])
        assert int.from_bytes(buf[2:10], "little") == part["dims"][0]["num_entites"]
        assert int.from_bytes(buf[10:14], "little") == part["dims"][0]["num_cores"]
        assert int.from_bytes(buf[14:18], "little") == len(part["dims"][0]["buffered_parts"])   # LO count of array 1
        with pytest.raises(ValueError):
            ppmio.loads(buf[:-3], compress)
        with pytest.raises(ValueError):
            ppmio.loads(buf + b"\0", compress)


def _oracle_fields(O, p, dim):
    dims = []
    ncore_elems = len(p.buffered_parts[dim])
    for d in range(4):
        if d > dim:
            z = np.zeros(0, np.int32)
            dims.append(dict(num_entites=0, num_cores=0, buffered_parts=z, offset_ents_per_rank=z,
                             ent_to_comm_arr_index=z, is_complete_part=z, num_bounds=0, num_boundaries=0,
                             boundary_parts=z, offset_bounded=z, bounded_ent_ids=z))
            continue
        # offset_bounded_per_dim: prefix over the boundary parts only (the oracle keeps one entry per rank)
        off = [0]
        for q in p.boundary_parts[d]:
            off.append(off[-1] + int(p.bounded_offset[d][q + 1] - p.bounded_offset[d][q]))
        dims.append(dict(num_entites=int(O.offsets[d][-1]), num_cores=len(p.buffered_parts[d]),
                         buffered_parts=np.asarray(p.buffered_parts[d], np.int32), offset_ents_per_rank=p.nents_offsets[d],
                         ent_to_comm_arr_index=p.comm_index[d], is_complete_part=p.is_complete[d],
                         num_bounds=len(p.buffered_parts[d]) - ncore_elems, num_boundaries=len(p.boundary_parts[d]),
                         boundary_parts=np.asarray(p.boundary_parts[d], np.int32), offset_bounded=np.asarray(off, np.int32),
                         bounded_ent_ids=p.bounded_ent_ids[d]))
    return {"version": 2, "is_full_mesh": bool(p.is_full_mesh), "dims": dims}


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["box", "annulus"])
def test_part_files_match_oracle(pp, ppo, opp, ppmio, tmp_path, which):
    """what pp_picpart_* hands to the writer == the oracle's fields, for every rank and every dimension 0..dim;
    the files are written under the reference's names and read back"""
    from pumipic_amd import capi
    capi.init(0)
    synth = pp.synth
    if which == "box":
        dim, (c, e, k), axis = 3, synth.kuhn_box(4), 0
    else:
        dim, (c, e, k), axis = 2, synth.annulus_tri(n_b=6, n_theta=24, band_width=3), 1
    owner = slab_owners(c, e, 4, axis=axis)
    O = opp.PicParts(ppo.Mesh(dim, c, e, k), owner, 4, opp.BFS, opp.BFS, buffer_layers=1, safe_layers=0)
    mesh = capi.Mesh(dim, c, e, k)
    comms = capi.Comm.local(4)
    parts = [capi.PicPart(mesh, owner, comms[r], capi.PART_BFS, capi.PART_BFS, 0, 1, 0) for r in range(4)]
    path = os.path.join(str(tmp_path), "mesh_" + which)
    for r, (po, pg) in enumerate(zip(O.parts, parts)):
        want = _oracle_fields(O, po, dim)
        _same(ppmio.fields_of_picpart(pg), want)
        f = ppmio.write_picpart(pg, path, r, 4)
        assert f == os.path.join(path + "_4.ppm", "mesh_%s_%d.ppm" % (which, r))  # file.cpp:47-66
        _same(ppmio.read_ppm(path, 4, r), want)
    for cm in comms:
        cm.destroy()
