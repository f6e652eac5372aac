"""Hand what a reference test program complained about to the oracle.

  PP_SEARCH_DUMP=<prefix> tests/_refdrivers/test_adj <mesh container>     # every search_mesh call dumps its arrays
  python tools/replay_search_dump.py <mesh container> <prefix>

For every dumped call in intersection mode: the wall hits whose recorded point lies outside their face (the check of
test/test_adj.cpp:640-652) and the particles that ended outside the bounding box without a wall face (:586-612) are
replayed -- with a sample of ordinary particles -- through the oracle's restatement of the reference's search
(oracle/: adjacency.tpp:284-361 with its bestFace fallback), from the same origins, targets and seed elements.  Prints,
per call, how many such particles there are and whether the oracle returns the same element, face and point bit for bit."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pumipic_amd_loader  # noqa: E402
import common  # noqa: E402


def read_container(path):
    with open(path, "rb") as f:
        hdr = np.fromfile(f, dtype=np.int32, count=4)
        dim, nv, ne = int(hdr[1]), int(hdr[2]), int(hdr[3])
        coords = np.fromfile(f, dtype=np.float64, count=nv * dim).reshape(nv, dim)
        e2v = np.fromfile(f, dtype=np.int32, count=ne * (dim + 1)).reshape(ne, dim + 1)
        cls = np.fromfile(f, dtype=np.int32, count=ne)
    return dim, coords, e2v, cls


def replay(mesh_file, prefix, ppo=None, verbose=True):
    """-> one dict per dumped intersection-mode call: slots, hits, off_face, lost, sampled, identical"""
    ppo = ppo or pumipic_amd_loader.load_oracle()
    dim, coords, e2v, cls = read_container(mesh_file)
    ne = len(e2v)
    mo = ppo.Mesh(dim, coords, e2v, cls)
    s2v = np.asarray(mo.side2verts).reshape(-1, dim)
    lo, hi = coords.min(axis=0), coords.max(axis=0)
    tol = mo.tolerance()
    rng = np.random.default_rng(1)
    ncalls = len(glob.glob(prefix + "_call*_hdr.bin"))
    out = []
    for k in range(ncalls):
        g = lambda what, dt: np.fromfile("%s_call%d_%s.bin" % (prefix, k, what), dtype=dt)
        cap, stride, seeded, req, looplimit, _ = g("hdr", np.int32)
        xo = g("xo", np.float64).reshape(3, stride)[:, :cap]
        xt = g("xt", np.float64).reshape(3, stride)[:, :cap]
        mask = g("mask", np.uint8)[:cap].astype(bool)
        selem, ein, eout = g("elem", np.int32)[:cap], g("ein", np.int32)[:cap], g("eout", np.int32)[:cap]
        if not req:
            # barycentric walk (no wall points): 5 000 particles of the call, the degenerate starts of the program's
            # "edge" populations among them, must end in the oracle's element
            pool = np.flatnonzero(mask & (ein >= 0 if seeded else True))
            take = np.sort(rng.choice(pool, min(5000, len(pool)), replace=False))
            take = take[np.argsort(selem[take], kind="stable")]
            n = len(take)
            pop = dict(dim=dim, coords=coords, e2v=e2v, cls=cls, ppe=np.bincount(selem[take], minlength=ne).astype(np.int32),
                       elem=selem[take], info=[xo[:, take], xt[:, take], np.arange(n, dtype=np.int32)])
            _, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH)
            ocap = po.capacity()
            om = po.slot_info()[1].astype(bool)
            oid = po.member(2)[0, :ocap]
            seed = None
            if seeded:
                seed = np.full(ocap, -1, dtype=np.int32)
                seed[om] = ein[take][oid[om]]
            ro = ppo.search_mesh(mo, po, elem_ids=seed, require_intersection=False, looplimit=int(looplimit))
            same = np.array_equal(ro["elem_ids"][:ocap][om], eout[take[oid[om]]])
            out.append(dict(call=k, mode="bcc", slots=int(mask.sum()), hits=0, off_face=0, lost=0, sampled=n,
                            identical=bool(same)))
            if verbose:
                print("call %d (barycentric walk): %d slots; oracle on %d of them: %s"
                      % (k, int(mask.sum()), n, "identical" if same else "DIFFERENT"))
            continue
        face, pts = g("face", np.int32)[:cap], g("pts", np.float64)[:cap * dim].reshape(cap, dim)
        hit = np.flatnonzero(mask & (face >= 0))
        if dim == 3:  # the point inside the triangle (test_adj.cpp:640-652)
            a, b, c = (coords[s2v[face[hit], j]] for j in range(3))
            nrm = np.cross(b - a, c - a)
            with np.errstate(invalid="ignore", divide="ignore"):
                bc = np.stack([np.einsum("ij,ij->i", nrm, np.cross(b - a, pts[hit] - a)),
                               np.einsum("ij,ij->i", nrm, np.cross(c - b, pts[hit] - b)),
                               np.einsum("ij,ij->i", nrm, np.cross(pts[hit] - a, c - a))]) / np.einsum("ij,ij->i", nrm, nrm)
            ok = np.isfinite(bc).all(axis=0) & (bc >= -tol).all(axis=0)
        else:  # the point on the edge (test_adj.cpp:614-628, 666-680)
            a, b = coords[s2v[face[hit], 0]], coords[s2v[face[hit], 1]]
            path, edge = pts[hit] - a, b - a
            with np.errstate(invalid="ignore"):
                ok = (np.abs(path[:, 0] * edge[:, 1] - path[:, 1] * edge[:, 0]) <= tol) & \
                     (pts[hit] >= np.minimum(a, b) - tol).all(axis=1) & (pts[hit] <= np.maximum(a, b) + tol).all(axis=1)
        off = hit[~ok]
        live_in = mask & (ein >= 0 if seeded else True)
        lost = np.flatnonzero(live_in & (face < 0) & (((xt.T[:, :dim] < lo - tol) | (xt.T[:, :dim] > hi + tol)).any(axis=1)))
        pool = np.flatnonzero(live_in)
        take = np.unique(np.concatenate([off, lost, rng.choice(pool, min(2000, len(pool)), replace=False)]))
        take = take[np.argsort(selem[take], kind="stable")]
        n = len(take)
        ppe = np.bincount(selem[take], minlength=ne).astype(np.int32)
        pop = dict(dim=dim, coords=coords, e2v=e2v, cls=cls, ppe=ppe, elem=selem[take],
                   info=[xo[:, take], xt[:, take], np.arange(n, dtype=np.int32)])
        _, po = common.oracle_pair(ppo, pop, ppo.PARTICLE_PUSH)
        ocap = po.capacity()
        om = po.slot_info()[1].astype(bool)
        oid = po.member(2)[0, :ocap]
        seed = None
        if seeded:
            seed = np.full(ocap, -1, dtype=np.int32)
            seed[om] = ein[take][oid[om]]
        ro = ppo.search_mesh(mo, po, elem_ids=seed, require_intersection=True, looplimit=int(looplimit))
        src = take[oid[om]]
        same = (np.array_equal(ro["elem_ids"][:ocap][om], eout[src]) and
                np.array_equal(ro["inter_faces"][:ocap][om], face[src]) and
                np.array_equal(ro["inter_points"][:ocap * dim].reshape(ocap, dim)[om].view(np.uint64),
                               pts[src].view(np.uint64)))
        out.append(dict(call=k, mode="intersection", slots=int(mask.sum()), hits=len(hit), off_face=len(off), lost=len(lost),
                        sampled=n - len(off) - len(lost), identical=bool(same)))
        if verbose:
            print("call %d: %d slots, %d wall hits, %d with the point off the face, %d outside without a face; oracle on "
                  "those + %d others: %s" % (k, int(mask.sum()), len(hit), len(off), len(lost), n - len(off) - len(lost),
                                             "identical bit for bit" if same else "DIFFERENT"))
    return out


def main():
    res = replay(sys.argv[1], sys.argv[2])
    print("most off-face / lost particles in one call: %d" % max([r["off_face"] + r["lost"] for r in res] or [0]))
    return 0 if all(r["identical"] for r in res) else 1


if __name__ == "__main__":
    sys.exit(main())
