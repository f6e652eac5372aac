"""Synthetic meshes and particle populations (input generators, numpy only).

The reference's meshes live in the un-vendored `pumipic-data` submodule, so every
BASELINE.json configuration is restated on deterministic synthetic inputs:

* ``plate_tri8_pardiag``  -- the 9-vertex / 8-triangle unit plate used by the reference's
  search2d and scatter KATs; element/vertex numbering reconstructed from the expected answers in
  test/search2d.cpp:205-308 and test/pseudoXGCm_scatter.cpp:58-178.
* ``kuhn_box``            -- Kuhn 6-tet split of an n^3 grid (config 1: 16^3 -> 24 576 tets).
* ``annulus_tri``         -- elliptic-polar triangle annulus around (h,k) (2-D literal pseudoXGCm).
* ``torus_tet``           -- that annulus revolved through n_planes toroidal planes, each wedge
  split into 3 tets with a vertex-id diagonal rule (conforming) (configs 2,3,5).

Seeds follow the reference drivers (ELEMENT_SEED=1024*1024, PARTICLE_SEED=512*512,
test/pseudoXGCm.cpp:14-15); the random STREAM is numpy's PCG64, not libstdc++'s minstd_rand0
(documented deviation: same distributions, different draws).
"""
import numpy as np

ELEMENT_SEED = 1024 * 1024
PARTICLE_SEED = 512 * 512
DISTRIBUTE_SEED = 1024 * 1024

# pseudoXGCm ellipse parameters (test/pseudoXGCm.cpp:470-472)
XGC_H = 1.72479370 - .08
XGC_K = .020558260
XGC_D = 0.6


def plate_tri8_pardiag():
    coords = np.array([[0, 0], [0, .5], [0, 1], [.5, .5], [.5, 0], [.5, 1], [1, .5], [1, 1],
                       [1, 0]], dtype=np.float64)
    e2v = np.array([[0, 4, 3], [1, 3, 5], [0, 3, 1], [3, 6, 7], [1, 5, 2], [3, 7, 5], [4, 6, 3],
                    [4, 8, 6]], dtype=np.int32)
    cls = np.ones(8, dtype=np.int32)
    return coords, e2v, cls


def _fix_tet_orientation(coords, e2v):
    p = coords[e2v]
    b0, b1, b2 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]
    vol = np.einsum("ij,ij->i", np.cross(b0, b1), b2)
    neg = vol < 0
    e2v[neg, 2], e2v[neg, 3] = e2v[neg, 3].copy(), e2v[neg, 2].copy()
    return e2v


def kuhn_box(n, lo=(0., 0., 0.), hi=(1., 1., 1.)):
    """n^3 cubes, 6 tets each (Kuhn/Freudenthal: one tet per axis permutation) -> 6 n^3 tets."""
    g = np.arange(n + 1)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float64) / n
    coords = lo + coords * (hi - lo)

    def vid(i, j, k):
        return (i * (n + 1) + j) * (n + 1) + k

    ci, cj, ck = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    ci, cj, ck = ci.ravel(), cj.ravel(), ck.ravel()
    perms = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
    tets = []
    for perm in perms:
        off = np.zeros((4, 3), dtype=np.int64)
        for s, ax in enumerate(perm):
            off[s + 1] = off[s]
            off[s + 1, ax] += 1
        v = [vid(ci + off[q, 0], cj + off[q, 1], ck + off[q, 2]) for q in range(4)]
        tets.append(np.stack(v, axis=1))
    # interleave so the 6 tets of one cube are consecutive
    e2v = np.stack(tets, axis=1).reshape(-1, 4).astype(np.int32)
    e2v = _fix_tet_orientation(coords, e2v)
    cls = np.ones(len(e2v), dtype=np.int32)
    return coords, e2v, cls


def annulus_tri(n_b=98, n_theta=512, h=XGC_H, k=XGC_K, d=XGC_D, b_lo=0.05, b_hi=0.6,
                band_width=7):
    """Elliptic-polar annulus x=h+d*b*cos(t), y=k+b*sin(t); 2*n_b*n_theta triangles.
    class_id = 1 + ring//band_width (class 1 gets the x0.01 slow-down, ellipticalPush.hpp:53)."""
    b = np.linspace(b_lo, b_hi, n_b + 1)
    t = 2 * np.pi * np.arange(n_theta) / n_theta
    B, T = np.meshgrid(b, t, indexing="ij")
    coords = np.stack([h + d * B * np.cos(T), k + B * np.sin(T)], axis=2).reshape(-1, 2)

    def vid(i, j):
        return i * n_theta + (j % n_theta)

    I, J = np.meshgrid(np.arange(n_b), np.arange(n_theta), indexing="ij")
    I, J = I.ravel(), J.ravel()
    v00, v10, v11, v01 = vid(I, J), vid(I + 1, J), vid(I + 1, J + 1), vid(I, J + 1)
    t0 = np.stack([v00, v10, v11], axis=1)
    t1 = np.stack([v00, v11, v01], axis=1)
    e2v = np.stack([t0, t1], axis=1).reshape(-1, 3).astype(np.int32)
    cls = (1 + np.repeat(I, 2) // band_width).astype(np.int32)
    # counter-clockwise check
    p = coords[e2v]
    a, b = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
    area = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    assert (area > 0).all()
    return np.ascontiguousarray(coords), e2v, cls


def torus_tet(n_b=14, n_theta=75, n_planes=16, h=XGC_H, k=XGC_K, d=XGC_D, b_lo=0.05, b_hi=0.6,
              band_width=1):
    """Revolve the (R,Z) annulus about the Z axis: 2*n_b*n_theta triangles x n_planes wedges x 3
    tets.  Default 14x75x16 -> 100 800 tets (BASELINE '100k-tet tokamak mesh');
    104x100x16 -> 998 400 tets (config 5).  A wedge over triangle (s0<s1<s2 by 2-D vertex id) is
    split {s0,s1,s2,s2'},{s0,s1,s2',s1'},{s0,s1',s2',s0'}: the diagonal of the quad over edge
    (u<v) always runs u(bottom)-v'(top), so neighbouring wedges agree."""
    c2, tri, cls2 = annulus_tri(n_b, n_theta, h, k, d, b_lo, b_hi, band_width)
    nv2 = len(c2)
    zeta = 2 * np.pi * np.arange(n_planes) / n_planes
    R, Z = c2[:, 0], c2[:, 1]
    coords = np.stack([np.outer(np.cos(zeta), R), np.outer(np.sin(zeta), R),
                       np.outer(np.ones(n_planes), Z)], axis=2).reshape(-1, 3)
    s = np.sort(tri, axis=1).astype(np.int64)
    tets = []
    for p in range(n_planes):
        bot = p * nv2
        top = ((p + 1) % n_planes) * nv2
        s0, s1, s2 = s[:, 0] + bot, s[:, 1] + bot, s[:, 2] + bot
        t0, t1, t2 = s[:, 0] + top, s[:, 1] + top, s[:, 2] + top
        w = np.stack([np.stack([s0, s1, s2, t2], 1), np.stack([s0, s1, t2, t1], 1),
                      np.stack([s0, t1, t2, t0], 1)], axis=1)  # (ntri, 3, 4)
        tets.append(w)
    e2v = np.stack(tets, axis=0).reshape(-1, 4).astype(np.int32)
    e2v = _fix_tet_orientation(coords, e2v)
    cls = np.tile(np.repeat(cls2, 3), n_planes).astype(np.int32)
    return np.ascontiguousarray(coords), e2v, cls


# ------------------------------------------------------------------ particle populations
def xgcm_source_counts(class_id, num_ptcls, mdl_face, seed=ELEMENT_SEED, remainder="last"):
    """Particles per element ~ round(Normal(mu, mu/4)) over elements with class_id <= mdl_face in
    element order until the total is reached; remainder into the last touched element
    (test/pseudoXGCm.cpp:167-222).  remainder="spread" is a non-literal option that avoids the
    single outlier element the literal rule creates when the draws fall short of the total."""
    ne = len(class_id)
    marked = np.flatnonzero(class_id <= mdl_face)
    ppe = np.zeros(ne, dtype=np.int32)
    if len(marked) == 0 or num_ptcls == 0:
        return ppe
    nppe = num_ptcls // len(marked)
    rng = np.random.Generator(np.random.PCG64(seed))
    draws = np.rint(rng.normal(nppe, float(nppe // 4), size=len(marked)))  # int division as in :189
    draws = np.maximum(draws, 0).astype(np.int64)
    cum = np.cumsum(draws)
    over = np.searchsorted(cum, num_ptcls, side="left")  # first index reaching the total
    if over < len(marked):
        draws[over] -= cum[over] - num_ptcls
        draws[over + 1:] = 0
        last = over
    else:
        last = len(marked) - 1
        rem = num_ptcls - cum[-1]
        if remainder == "last":   # literal: everything into the last touched element (:210-213)
            draws[last] += rem
        else:                     # "spread": one extra particle per element, round-robin
            draws += rem // len(marked)
            draws[:rem % len(marked)] += 1
    ppe[marked] = draws
    assert ppe.sum() == num_ptcls
    return ppe


def particles_in_elements(coords, e2v, ppe, seed=PARTICLE_SEED, chunk=1 << 21):
    """Uniform positions inside each particle's element (pseudoXGCm.cpp:224-264 for triangles:
    r1,r2 with the x+y>1 fold; tets: sorted-uniform barycentrics).  Returns (elem, xyz[3,np]).
    Generated in chunks (the random stream is consumed in the same order, so the result does not
    depend on the chunk size): 32 M tets' worth of vertex gathers would otherwise peak at ~7 GB per
    rank, and 8 ranks build their populations at the same time."""
    dim = coords.shape[1]
    np_ = int(ppe.sum())
    elem = np.repeat(np.arange(len(ppe), dtype=np.int32), ppe)
    rng = np.random.Generator(np.random.PCG64(seed))
    xyz = np.zeros((3, np_))
    for lo in range(0, np_, chunk):
        hi = min(np_, lo + chunk)
        p = coords[e2v[elem[lo:hi]]]  # (n, dim+1, dim)
        if dim == 2:
            r = rng.random((hi - lo, 2))
            fold = r.sum(axis=1) > 1
            r[fold] = 1 - r[fold]
            xy = p[:, 0] + r[:, :1] * (p[:, 1] - p[:, 0]) + r[:, 1:] * (p[:, 2] - p[:, 0])
            xyz[0, lo:hi], xyz[1, lo:hi] = xy[:, 0], xy[:, 1]
        else:
            u = np.sort(rng.random((hi - lo, 3)), axis=1)
            w = np.stack([u[:, 0], u[:, 1] - u[:, 0], u[:, 2] - u[:, 1], 1 - u[:, 2]], axis=1)
            # pull slightly towards the centroid so no particle starts on a face
            w = 0.98 * w + 0.02 * 0.25
            xyz[:, lo:hi] = np.einsum("nj,njk->nk", w, p).T
    return elem, xyz


def elliptical_state(R, Z, h=XGC_H, k=XGC_K, d=XGC_D):
    """(b, phi) of ellipticalPush::setup (test/ellipticalPush.hpp:22-33), float32 like the
    reference's Particle type.  Host-side initialisation helper (numpy libm)."""
    phi = np.arctan2(d * (Z - k), R - h)
    b = (Z - k) / np.sin(phi)
    return b.astype(np.float32), phi.astype(np.float32)


def push_and_search_population(coords, e2v, num_ptcls, face_axis=1, face_value=0.0, tol=1e-12):
    """pseudoPushAndSearch source rule (test/pseudoPushAndSearch.cpp:228-298): elements having a
    face on the model face (here: the box side `axis == value`), equal count each, remainder to
    the last marked element; particles start at element centroids."""
    p = coords[e2v]  # (ne,4,3)
    on = np.abs(p[:, :, face_axis] - face_value) < tol
    marked = np.flatnonzero(on.sum(axis=1) >= 3)
    ne = len(e2v)
    ppe = np.zeros(ne, dtype=np.int32)
    if len(marked):
        ppe[marked] = num_ptcls // len(marked)
        ppe[marked[-1]] += num_ptcls % len(marked)
    elem = np.repeat(np.arange(ne, dtype=np.int32), ppe)
    # Omega_h average(): ((p0+p1)+p2)+p3 then /4
    c = ((p[:, 0] + p[:, 1]) + p[:, 2]) + p[:, 3]
    c = c / 4
    xyz = np.ascontiguousarray(c[elem].T)
    return ppe, elem, xyz


def distribute_particles(ne, np_, strat, seed=0):
    """particle_structs/test/Distribute.cpp:323-331 strategies 0 (even), 1 (uniform), 2 (gaussian
    ne/2, ne/8 clamped), 3 (exponential-ish), 4 (GITRm-like 85% in first 40%).  Fixed seed
    instead of wall-clock (SURVEY config 4).  Returns (ppe, elem_per_ptcl)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if strat == 0:
        p, r = (np_ // ne, np_ % ne) if ne else (0, 0)
        ppe = np.full(ne, p, dtype=np.int32)
        ppe[:r] += 1
        epp = np.repeat(np.arange(ne, dtype=np.int32), ppe)
        return ppe, epp
    if strat == 1:
        epp = rng.integers(0, ne, size=np_, dtype=np.int32)
    elif strat == 2:
        epp = rng.normal(ne / 2.0, ne / 8.0, size=np_).astype(np.int64)
        epp = np.clip(epp, 0, ne - 1).astype(np.int32)
    elif strat == 3:
        lam = 1.0
        freq_max = -np.log(1.0 / ne)
        uni = rng.integers(0, ne, size=np_)
        pe = uni / ne
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = -1 / lam * np.log(1 - pe) / freq_max
            t1 = -1 / lam * np.log(1 - pe - 1.0 / ne) / freq_max
        start = np.where(uni == ne - 1, 0, (t0 * ne)).astype(np.int64)
        end = np.where(uni == ne - 1, 0, np.nan_to_num(t1 * ne, posinf=ne, nan=ne)).astype(np.int64)
        length = np.maximum(end - start, 1)
        inside = (rng.random(np_) * length).astype(np.int64)
        e = start + np.where(length > 1, inside, 0)
        redo = e >= ne
        e[redo] = rng.integers(0, ne, size=int(redo.sum()))
        e[uni == ne - 1] = 0
        epp = e.astype(np.int32)
    elif strat == 4:
        cutoff = 2 * ne // 5
        first = int(np.ceil(np_ * 0.85))
        epp = np.concatenate([rng.integers(0, max(cutoff, 1), size=first),
                              rng.integers(cutoff, ne, size=np_ - first)]).astype(np.int32)
    else:
        raise ValueError("unknown distribution strategy %d" % strat)
    ppe = np.bincount(epp, minlength=ne).astype(np.int32)
    return ppe, epp


def write_mesh_bin(path, dim, coords, e2v, cls):
    """Tiny binary mesh container for the C++ drivers: int32 header (magic, dim, nverts, nelems)
    then coords f64, elem2verts i32, class_id i32."""
    with open(path, "wb") as f:
        np.array([0x50504D31, dim, len(coords), len(e2v)], dtype=np.int32).tofile(f)
        np.ascontiguousarray(coords, dtype=np.float64).tofile(f)
        np.ascontiguousarray(e2v, dtype=np.int32).tofile(f)
        np.ascontiguousarray(cls, dtype=np.int32).tofile(f)
