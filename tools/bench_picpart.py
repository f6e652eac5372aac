#!/usr/bin/env python
"""Set-up and reduction times of PICparts at the configs[4] mesh size (run on a GPU box):
   python tools/bench_picpart.py [nranks]
998 400-tet torus, `nranks` toroidal slabs as virtual ranks of one process (the exchange is device-to-device
copies on one GPU: what is timed is the library's pack / combine / unpack kernels and its bookkeeping, not a
fabric), BFS buffer of 2 layers, safe zone of 1."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pumipic_amd_loader  # noqa: E402


def main(P=8):
    pp = pumipic_amd_loader.load()
    from pumipic_amd import capi
    capi.init(0)
    c, e, k = pp.synth.torus_tet(n_b=26, n_theta=400, n_planes=16)
    cen = c[e].mean(axis=1)
    ang = np.arctan2(cen[:, 1], cen[:, 0])
    owner = np.minimum(((ang + np.pi) / (2 * np.pi) * P).astype(np.int32), P - 1)
    t0 = time.perf_counter()
    mesh = capi.Mesh(3, c, e, k)
    print("full mesh: %d tets, %d vertices, pp_mesh_create %.2f s" % (mesh.nelems, mesh.nverts, time.perf_counter() - t0))
    comms = capi.Comm.local(P)
    parts = []
    for r in range(P):
        t0 = time.perf_counter()
        parts.append(capi.PicPart(mesh, owner, comms[r], capi.PART_BFS, capi.PART_BFS, 0, 2, 1))
        if r < 2:
            print("rank %d: pp_picpart_create %.2f s, part of %d tets / %d vertices, %d buffered parts" % (
                r, time.perf_counter() - t0, parts[-1].nents[3], parts[-1].nents[0], parts[-1].num_buffers))
    for d, nv, name in ((0, 2, "vertices x 2 doubles (gyroSync)"), (3, 1, "elements x 1 double")):
        arrs = [capi.DevArray.from_host(np.ones(p.nents[d] * nv)) for p in parts]
        for _ in range(3):
            capi.picpart_reduce_all(parts, d, capi.OP_SUM, arrs)
        capi.sync()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            capi.picpart_reduce_all(parts, d, capi.OP_SUM, arrs)
        capi.sync()
        dt = (time.perf_counter() - t0) / n
        print("reduceCommArray SUM over %s: %.3f ms for all %d ranks (%.3f ms per rank)" % (name, dt * 1e3, P, dt * 1e3 / P))
    t0 = time.perf_counter()
    bal = capi.Balancer(parts[0])
    print("pp_balancer_create %.2f s, %d sbars" % (time.perf_counter() - t0, len(bal.sbars())))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
