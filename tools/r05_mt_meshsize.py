#!/usr/bin/env python
"""Intersection-mode walk (k_search_mt3) against the size of the mesh: picoseconds per visited element with the tet
records (128 B each) inside / outside one XCD's 4 MB L2 -- is the walk bound by the L2-miss path?   (GPU box)
   python tools/r05_mt_meshsize.py [particles]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pumipic_amd_loader  # noqa: E402


def main(nptcl=4_000_000):
    pp = pumipic_amd_loader.load()
    from pumipic_amd import capi
    synth = pp.synth
    capi.init(0)
    for n_b, n_theta in ((5, 27), (7, 38), (10, 53), (14, 75), (20, 106), (28, 150), (40, 212)):
        coords, e2v, cls = synth.torus_tet(n_b=n_b, n_theta=n_theta)
        ne = len(e2v)
        ppe = synth.xgcm_source_counts(cls, nptcl, 12, seed=synth.ELEMENT_SEED, remainder="last")
        elem, xyz = synth.particles_in_elements(coords, e2v, ppe, seed=synth.PARTICLE_SEED)
        b, phi = synth.elliptical_state(np.hypot(xyz[0], xyz[1]), xyz[2])
        info = [xyz, np.zeros_like(xyz), np.arange(nptcl, dtype=np.int32), b, phi]
        mesh = capi.Mesh(3, coords, e2v, cls)
        ps = capi.PS.scs(capi.PARTICLE_XGCM, ne, ppe, C_=64, sigma=2**31 - 1, V=1024, pad_strat=0, shuffle_padding=0.1,
                         extra_padding=0.0, particle_elements=elem, particle_info=info)
        cap = ps.capacity()
        ids = capi.DevArray(cap, np.int32)
        xface = capi.DevArray(cap, np.int32)
        xpts = capi.DevArray(3 * cap, np.float64)
        times = []
        for it in range(4):
            capi.toroidal_push(ps, mesh, synth.XGC_H, synth.XGC_K, synth.XGC_D, 0.5)
            capi.sync()
            t0 = time.perf_counter()
            capi.check(capi.lib().pp_search_mesh(mesh.p, ps.p, 0, 1, 2, ids.ptr, 0, 1, xface.ptr, xpts.ptr, 2000, None, None))
            capi.sync()
            times.append(time.perf_counter() - t0)
        dt = min(times[1:])
        visits = capi.search_walk_steps()
        print("%8d tets (%6.1f MB of records): %6.2f ms, %5.1f visits / particle, %6.1f ps / visit, records fetched at %5.2f TB/s"
              % (ne, ne * 128 / 1e6, dt * 1e3, visits / nptcl, dt / visits * 1e12, visits * 128 / dt / 1e12), flush=True)
        del ps, mesh


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000)
