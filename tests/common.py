"""Shared builders for parity tests: the same synthetic mesh + particle population is loaded
into the CPU oracle (oracle/ppo.py) and into the HIP library (pumi-pic_amd/capi.py)."""
import os

import numpy as np

# PP_TEST_SHUFFLING=0 runs the suite with the reference's reshuffle fast path switched off on both
# sides (SellCSigma::setShuffling): every rebuild is the full counting-sort re-layout.  Default: on,
# like the reference -- the in-place rebuild is what normally runs.
SHUFFLING = os.environ.get("PP_TEST_SHUFFLING", "1") != "0"


def set_shuffling(*structures, on=None):
    for ps in structures:
        if ps is not None and hasattr(ps, "set_try_shuffling"):
            ps.set_try_shuffling(SHUFFLING if on is None else on)



def population_2d(synth, n_b=12, n_theta=48, num_ptcls=3000, mdl_face=3, band_width=3):
    coords, e2v, cls = synth.annulus_tri(n_b=n_b, n_theta=n_theta, band_width=band_width)
    ppe = synth.xgcm_source_counts(cls, num_ptcls, mdl_face)
    elem, xyz = synth.particles_in_elements(coords, e2v, ppe)
    b, phi = synth.elliptical_state(xyz[0], xyz[1])
    ids = np.arange(num_ptcls, dtype=np.int32)
    info = [xyz, np.zeros_like(xyz), ids, b, phi]
    return dict(dim=2, coords=coords, e2v=e2v, cls=cls, ppe=ppe, elem=elem, info=info)


def population_3d(synth, n_b=6, n_theta=16, n_planes=8, num_ptcls=3000, mdl_face=5):
    coords, e2v, cls = synth.torus_tet(n_b=n_b, n_theta=n_theta, n_planes=n_planes)
    ppe = synth.xgcm_source_counts(cls, num_ptcls, mdl_face)
    elem, xyz = synth.particles_in_elements(coords, e2v, ppe)
    R = np.hypot(xyz[0], xyz[1])
    b, phi = synth.elliptical_state(R, xyz[2])
    ids = np.arange(num_ptcls, dtype=np.int32)
    info = [xyz, np.zeros_like(xyz), ids, b, phi]
    return dict(dim=3, coords=coords, e2v=e2v, cls=cls, ppe=ppe, elem=elem, info=info)


def population_box(synth, n=4, num_ptcls=500):
    coords, e2v, cls = synth.kuhn_box(n)
    ppe, elem, xyz = synth.push_and_search_population(coords, e2v, num_ptcls)
    ids = np.arange(num_ptcls, dtype=np.int32)
    info = [xyz, np.zeros_like(xyz), ids]
    return dict(dim=3, coords=coords, e2v=e2v, cls=cls, ppe=ppe, elem=elem, info=info)


def by_id(ids, mask, values):
    """values of live slots ordered by the particle-id member"""
    live = np.flatnonzero(mask)
    order = np.argsort(ids[live], kind="stable")
    return ids[live][order], np.asarray(values)[..., live[order]]


def oracle_pair(ppo, pop, members, kind="scs", C=64, V=1024, sigma=2**31 - 1):
    mesh = ppo.Mesh(pop["dim"], pop["coords"], pop["e2v"], pop["cls"])
    ne = len(pop["e2v"])
    if kind == "scs":
        ps = ppo.PS.scs(members, ne, pop["ppe"], C_max=C, sigma=sigma, V=V,
                        particle_elements=pop["elem"], particle_info=pop["info"])
    else:
        ps = ppo.PS.csr(members, ne, pop["ppe"], particle_elements=pop["elem"],
                        particle_info=pop["info"])
    return mesh, ps


def gpu_pair(capi, pop, members, kind="scs", C=64, V=1024, sigma=2**31 - 1):
    mesh = capi.Mesh(pop["dim"], pop["coords"], pop["e2v"], pop["cls"])
    ne = len(pop["e2v"])
    if kind == "scs":
        ps = capi.PS.scs(members, ne, pop["ppe"], C_=C, sigma=sigma, V=V,
                         particle_elements=pop["elem"], particle_info=pop["info"])
    else:
        ps = capi.PS.csr(members, ne, pop["ppe"], particle_elements=pop["elem"],
                         particle_info=pop["info"])
    return mesh, ps


def class_interface_functor(topo, mask, hits):
    """A user functor for trace_particle_through_mesh (the `Func` argument, adjacency.tpp:470-476):
    exposed sides behave as in RemoveParticleOnGeometricModelExit; an INTERIOR side between
    elements of different class_id stops the particle in its current element and records the side
    in inter_faces -- what a wall model on an internal material interface does.  Operates on host
    arrays (numpy); `topo` is the oracle mesh (side numbering is shared with the GPU library)."""
    def func(st):
        done, le, elem, faces = st["ptcl_done"], st["last_exit"], st["elem_ids"], st["inter_faces"]
        idx = np.flatnonzero(mask.astype(bool) & (done[:len(mask)] == 0))
        if idx.size == 0:
            return
        bridge = le[idx]
        exposed = topo.side_exposed[bridge].astype(bool)
        first = topo.side2elems_off[bridge]
        a = topo.side2elems[first]
        b = topo.side2elems[np.where(exposed, first, first + 1)]
        iface = ~exposed & (topo.class_id[a] != topo.class_id[b])
        done[idx] = (exposed | iface).astype(done.dtype)
        if st["require_intersection"]:
            faces[idx[exposed | iface]] = bridge[exposed | iface]
            hits.append(int(iface.sum()))
        else:
            elem[idx[exposed]] = -1
            faces[idx[iface]] = bridge[iface]
            hits.append(int(iface.sum()))
    return func


def on_device(func):
    """run a host functor on the device arrays of the stepwise GPU walk (download, apply, upload)"""
    def wrapped(st):
        keys = ("elem_ids", "inter_faces", "last_exit", "inter_points", "ptcl_done")
        host = {k: st[k].to_host() for k in keys}
        host.update({k: v for k, v in st.items() if k not in keys})
        func(host)
        for k in keys:
            st[k].upload(host[k])
    return wrapped


def radial_kick(xt, dim, h, k, seed=3, lo=0.75, hi=1.3):
    """scale the targets about the magnetic axis so that walks cross the class bands of the
    synthetic annulus / torus (the elliptical push alone keeps a particle on its flux surface)"""
    rng = np.random.default_rng(seed)
    s = rng.uniform(lo, hi, xt.shape[1])
    out = xt.copy()
    if dim == 2:
        out[0] = h + (xt[0] - h) * s
        out[1] = k + (xt[1] - k) * s
    else:
        r = np.hypot(xt[0], xt[1])
        rn = h + (r - h) * s
        out[0], out[1] = xt[0] * rn / r, xt[1] * rn / r
        out[2] = k + (xt[2] - k) * s
    return out


def check_scs_valid(ps, ne):
    """structural validity of an SCS layout whatever policy produced it: rows <-> elements is a
    bijection, every slot's parent follows its row, live slots are a prefix of every row, offsets
    tile [0, capacity)"""
    L = ps.layout()
    C_, nrows, cap = L["C"], L["num_rows"], L["capacity"]
    r2e, e2r = L["row_to_element"], L["element_to_row"]
    assert len(r2e) == nrows and np.array_equal(np.sort(r2e), np.arange(nrows))
    assert np.array_equal(e2r[r2e], np.arange(nrows))
    off, s2c = L["offsets"], L["slice_to_chunk"]
    assert off[0] == 0 and off[-1] == cap and np.all(np.diff(off) >= 0) and np.all(np.diff(off) % C_ == 0)
    assert np.all(np.diff(s2c) >= 0)
    se, mk = L["slot_elem"], L["mask"].astype(bool)
    for s in range(len(s2c)):
        c, lo, hi = s2c[s], off[s], off[s + 1]
        if hi == lo:
            continue
        blk_e = se[lo:hi].reshape(-1, C_)
        assert np.all(blk_e == r2e[c * C_:(c + 1) * C_][None, :]), "slot parent != row element"
    # prefix compactness: per chunk, concatenate the slices' columns and check monotone masks
    first = {}
    for s in range(len(s2c)):
        first.setdefault(int(s2c[s]), []).append(s)
    for c, sl in first.items():
        cols = np.concatenate([mk[off[s]:off[s + 1]].reshape(-1, C_) for s in sl], axis=0)
        assert np.all(cols[1:] <= cols[:-1]), "row is not prefix-compact"
    assert np.all(se[mk] < ne)
    return L
