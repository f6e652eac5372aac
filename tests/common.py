"""Shared builders for parity tests: the same synthetic mesh + particle population is loaded
into the CPU oracle (oracle/ppo.py) and into the HIP library (pumi-pic_amd/capi.py)."""
import numpy as np


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
