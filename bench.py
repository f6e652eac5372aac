#!/usr/bin/env python
"""bench.py -- particles pushed+searched(+scattered+rebuilt) per second on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).
A "step" is one pass of the hot path over the resident particle population:

  c2 (default, BASELINE.json configs[1]): fused toroidal push + BCC adjacency walk on the
      100 800-tet tokamak mesh, 10 M particles per GPU, SCS layout (C=64), no rebuild; positions
      ping-pong x <-> x_tgt and the walk is re-seeded from the previous step's element ids.
  c3 (configs[2]): c2 + updatePtclPositions + SCS rebuild + gyroScatter x2 every step (tet variant
      of the ring map: 4 vertices per ring point); 2dc3 is the 2-D literal of it.  The
      three calls go through pp_ps_rebuild_scatter (same work, one entry point;
      PP_BENCH_SEPARATE_SCATTER=1 issues them separately).
  2d : the literal 2-D pseudoXGCm step (elliptical push + search_mesh_2d) on 100 352 triangles.
  c5 (configs[4], opt-in): c3 with ownership: every rank owns a block of elements; after the search
      the particles whose new element another rank owns are packed into records, exchanged with ONE
      all-to-all-v (RCCL) and enter the receiver's rebuild as new particles; the two scatter fields
      are summed over ranks (gyroSync).  Use with --mesh 1m --particles 32000000 for the config.

  c4 (configs[3], ps_combo160): the 160-byte PerfTypes160 particle (double[17], int[4], long); a step
      is one pseudo-push pass (ps_combo160.cpp:158-178: 160 B written per particle) + one
      redistribute(percentMoved 0.5, uniform) + rebuild round (:186-232).  --c4-elems/--particles
      choose the point: the stress point 1 M / 1 M (default) or the script point 50 000 / 50 M;
      --structure scs|csr.  The roofline kernel is the pseudo-push (161 B per particle).

Inputs are synthetic (pumi-pic_amd/synth.py) and resident in HBM before the timed region.
N > 1: one process per GPU (torch.distributed / RCCL); every rank owns a contiguous block of
elements and the particles inside it; the full mesh is replicated (reference `Input::FULL`
buffering), so in c2 no particle leaves its safe zone and there is no data-path collective:
value = sum over ranks of particles / max-over-ranks time ("weak" scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import pumipic_amd_loader  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# algorithmic bytes per particle and step (DESIGN.md "roofline")
BYTES = {
    # read x(24) b,phi(8) mask(1) seed elem(4) ; write x_tgt(24) phi(4) elem(4)
    "c2": 69.0,
    # c2 + rebuild: read all members 60 + new_element 4, write all members 60 + mask 1
    "c3": 69.0 + 125.0,
    # read b,phi(8) mask(1) seed(4) ; write x_tgt x,y(16) phi(4) elem(4)
    "2d": 37.0,
    # pseudo-push: write double[17]+int[4]+long = 160, read mask 1 (parentElmData 8 B per element)
    "c4": 161.0,
}


def build_workload(pp, capi, name, nptcl, rank, world, deg, remainder="last", mesh_size="100k",
                   sigma=2**31 - 1):
    synth = pp.synth
    if name in ("2d", "2dc3"):
        coords, e2v, cls = synth.annulus_tri()
        dim, mdl = 2, 12
        label = "pseudoXGCm 2-D literal: 100352-tri annulus"
    elif mesh_size == "1m":  # BASELINE configs[4]: 998 400 tets (this GPU's share of the particles)
        coords, e2v, cls = synth.torus_tet(n_b=104, n_theta=100, band_width=8)
        dim, mdl = 3, 12
        label = "pseudoXGCm 998400-tet tokamak mesh"
    else:
        coords, e2v, cls = synth.torus_tet()
        dim, mdl = 3, 12
        label = "pseudoXGCm 100800-tet tokamak mesh"
    ne = len(e2v)
    # element-block ownership: rank r owns elements [r*ne/world, (r+1)*ne/world)
    lo, hi = rank * ne // world, (rank + 1) * ne // world
    cls_own = np.where((np.arange(ne) >= lo) & (np.arange(ne) < hi), cls, 1 << 20)
    # the synthetic population is a pure function of (mesh, n, rank, world): memoise it on disk so
    # repeated profiler passes on one box do not regenerate 10 M particles every time
    cache = os.path.join(os.environ.get("PP_BENCH_CACHE", "/tmp"),
                         "pp_pop_%dd_%d_%d_%d_%d_%s.npz" % (dim, len(e2v), nptcl, rank, world, remainder))
    if os.path.exists(cache):
        z = np.load(cache)
        ppe, elem, xyz, b, phi = z["ppe"], z["elem"], z["xyz"], z["b"], z["phi"]
    else:
        ppe = synth.xgcm_source_counts(cls_own, nptcl, mdl, seed=synth.ELEMENT_SEED + rank,
                                       remainder=remainder)
        elem, xyz = synth.particles_in_elements(coords, e2v, ppe, seed=synth.PARTICLE_SEED + rank)
        R = np.hypot(xyz[0], xyz[1]) if dim == 3 else xyz[0]
        Z = xyz[2] if dim == 3 else xyz[1]
        b, phi = synth.elliptical_state(R, Z)
        try:
            np.savez(cache, ppe=ppe, elem=elem, xyz=xyz, b=b, phi=phi)
        except OSError:
            pass
    info = [xyz, np.zeros_like(xyz), np.arange(nptcl, dtype=np.int32), b, phi]
    mesh = capi.Mesh(dim, coords, e2v, cls)
    ps = capi.PS.scs(capi.PARTICLE_XGCM, ne, ppe, C_=64, sigma=sigma, V=1024, pad_strat=0,
                     shuffle_padding=0.1, extra_padding=0.0, particle_elements=elem,
                     particle_info=info)
    return dict(mesh=mesh, ps=ps, dim=dim, label=label, ne=ne, coords=coords, e2v=e2v, cls=cls,
                ppe=ppe, elem=elem, info=info, rank=rank, world=world)


def build_c4(pp, capi, ne, nptcl, rank, structure):
    """ps_combo160 set-up (performance_tests/ps_combo160.cpp:60-130): uniform distribution (strategy
    1, fixed seed instead of the wall clock), Sell-64-ne (sigma = ne, V = 1024) or CSR."""
    rng = np.random.default_rng(rank)
    elems = np.sort(rng.integers(0, ne, size=nptcl).astype(np.int32))
    ppe = np.bincount(elems, minlength=ne).astype(np.int32)
    info = [np.zeros((17, nptcl)), np.zeros((4, nptcl), dtype=np.int32),
            np.arange(nptcl, dtype=np.int64)[None, :]]
    if structure == "scs":
        ps = capi.PS.scs(capi.PERF160, ne, ppe, C_=64, sigma=ne, V=1024, particle_elements=elems,
                         particle_info=info)
    else:
        ps = capi.PS.csr(capi.PERF160, ne, ppe, particle_elements=elems, particle_info=info)
    parent = capi.DevArray.from_host(np.sqrt(np.arange(ne, dtype=np.float64)) * np.arange(ne))
    return dict(ps=ps, parent=parent, ne=ne, dim=0, rank=rank, world=1,
                label="ps_combo160 %s, %d elements" % ("Sell-64-ne" if structure == "scs" else "CSR", ne))


class StepperC4:
    def __init__(self, capi, w):
        self.capi, self.ps, self.parent = capi, w["ps"], w["parent"]
        self.new_elems = None
        self.kernel_ms = []
        self.round = 0

    def step(self, timed=False):
        capi = self.capi
        # HIP events bracket every `sample_every`-th timed launch (about ten samples per run): an event
        # pair per step costs ~15 us of serialisation on a 0.3 ms step
        if timed:
            self.ntimed = getattr(self, "ntimed", 0) + 1
            timed = (self.ntimed - 1) % getattr(self, "sample_every", 1) == 0 and len(self.kernel_ms) < 64
        if timed:
            e0, e1 = capi.Event(), capi.Event()
            e0.record()
        capi.pseudo_push160(self.ps, self.parent)
        if timed:
            e1.record()
            self.kernel_ms.append((e0, e1))
        self.new_elems = capi.redistribute_particles(self.ps, 0.5, seed=self.round, out=self.new_elems)
        self.round += 1
        self.ps.rebuild(self.new_elems)
        cap = max(self.ps.capacity(), 1)
        if cap > self.new_elems.n:
            self.new_elems = capi.DevArray(cap + cap // 10, np.int32)

    def kernel_avg_ms(self):
        self.capi.sync()
        ms = [a.elapsed_ms(b) for a, b in self.kernel_ms]
        return sum(ms) / len(ms) if ms else None


def cpu_baseline_c4(pp, ne, nptcl, sample, rounds=3):
    """the oracle's pseudo-push + redistribute + rebuild on a bounded sample, one core"""
    from oracle import ppo
    rng = np.random.default_rng(0)
    ne_s = max(1, int(ne * (sample / max(nptcl, 1))))  # same particles per element as the GPU run
    elems = np.sort(rng.integers(0, ne_s, size=sample).astype(np.int32))
    ppe = np.bincount(elems, minlength=ne_s).astype(np.int32)
    info = [np.zeros((17, sample)), np.zeros((4, sample), dtype=np.int32),
            np.arange(sample, dtype=np.int64)[None, :]]
    ps = ppo.PS.scs(ppo.PERF160, ne_s, ppe, C_max=1, sigma=ne_s, V=1024, particle_elements=elems,
                    particle_info=info)
    parent = np.sqrt(np.arange(ne_s, dtype=np.float64)) * np.arange(ne_s)
    t0 = time.perf_counter()
    for r in range(rounds):
        ppo.pseudo_push160(ps, parent)
        ps.rebuild(ppo.redistribute_particles(ps, 0.5, seed=r))
    dt = time.perf_counter() - t0
    return dict(value=sample * rounds / dt, unit="particles/s", cores=1, kind="port",
                sample="%d particles / %d elements x %d rounds of pseudo-push + redistribute + rebuild, "
                       "oracle (C=1 Serial semantics), 1 core" % (sample, ne_s, rounds))


class Stepper:
    def __init__(self, pp, capi, w, name, deg):
        self.capi, self.w, self.name, self.deg = capi, w, name, deg
        s = pp.synth
        self.h, self.k, self.d = s.XGC_H, s.XGC_K, s.XGC_D
        self.ps, self.mesh = w["ps"], w["mesh"]
        cap = max(self.ps.capacity(), 1)
        self.ids = capi.DevArray.from_host(np.full(cap + cap // 10, -1, dtype=np.int32))
        self.first = True
        self.kernel_ms = []
        if name == "c5":
            from pumipic_amd import dist as ppdist
            self.ppdist = ppdist
            self.rank, self.world = w["rank"], w["world"]
            owners = ppdist.element_block_owners(w["ne"], self.world)
            self.owners = capi.DevArray.from_host(owners)
            self.safe = capi.DevArray.from_host((owners == self.rank).astype(np.uint8))
            if w.get("safe_layers", 0) > 0:
                # PICpart safe zone: the core plus `safe_layers` breadth-first layers of the replicated
                # mesh (bfsBufferLayers, pumipic_part_construct.cpp:407-437); particles migrate only
                # when they leave it
                self.safe, _ = capi.bfs_buffer_layers(self.mesh, self.owners, self.rank, self.world,
                                                      w["safe_layers"], w["safe_layers"])
            self.moved = 0
        if name in ("c3", "2dc3", "c5"):
            self.fwd, self.bkwd = capi.create_gyro_ring_mappings(self.mesh)
            self.w_f = capi.DevArray(self.mesh.nverts, np.float64)
            self.w_b = capi.DevArray(self.mesh.nverts, np.float64)

    def step(self, timed=False):
        capi = self.capi
        # HIP events bracket every `sample_every`-th timed launch (about ten samples per run): an event
        # pair per step costs ~15 us of serialisation on a 0.3 ms step
        if timed:
            self.ntimed = getattr(self, "ntimed", 0) + 1
            timed = (self.ntimed - 1) % getattr(self, "sample_every", 1) == 0 and len(self.kernel_ms) < 64
        if timed:
            e0, e1 = capi.Event(), capi.Event()
            e0.record()
        if self.name == "c2":
            capi.push_search(self.mesh, self.ps, self.h, self.k, self.d, self.deg, self.ids,
                             seeded=not self.first, looplimit=200, want_found=False)
        elif self.name in ("c3", "c5"):
            # rebuilt every step: every particle sits in its row's element, no seed ids needed
            capi.push_search(self.mesh, self.ps, self.h, self.k, self.d, self.deg, self.ids,
                             seeded=False, looplimit=200, want_found=False)
        else:
            capi.push_search(self.mesh, self.ps, self.h, self.k, self.d, self.deg, self.ids,
                             seeded=True, looplimit=200, want_found=False)
        if timed:
            e1.record()
            self.kernel_ms.append((e0, e1))
        self.first = False
        if self.name == "c2":
            self.ps.swap_members(0, 1)  # x <-> x_tgt (O(1)); no rebuild in config 2
        elif self.name in ("c3", "2dc3"):
            # the drivers' rebuild(): updatePtclPositions + migrate/rebuild (pseudoXGCm.cpp:116-140)
            if os.environ.get("PP_BENCH_SEPARATE_SCATTER"):
                self.ps.rebuild_commit(self.ids)
                capi.gyro_scatter(self.mesh, self.ps, self.fwd, out=self.w_f)
                capi.gyro_scatter(self.mesh, self.ps, self.bkwd, out=self.w_b)
            else:  # the same three calls as one entry point (scatter enqueued before the rebuild's sync)
                capi.rebuild_scatter(self.ps, self.mesh, self.ids, [self.fwd, self.bkwd],
                                     [self.w_f, self.w_b])
            cap = max(self.ps.capacity(), 1)
            if cap > self.ids.n:  # 10% slack: the capacity wanders by a few chunk widths per rebuild
                self.ids = capi.DevArray(cap + cap // 10, np.int32)
            if self.w["dim"] == 2:
                self.ids.fill_bytes(0xff)  # search_mesh_2d reads its seeds: -1 = the row's element
            # dim 3, unseeded (an "empty elem_ids", adjacency.tpp:504-515): the search writes every
            # slot itself, -1 into the masked ones -- no fill
        elif self.name == "c5":
            self.route = capi.set_unsafe_procs(self.ps, self.ids, self.safe, self.owners, self.rank,
                                               out=getattr(self, "route", None))
            ne_, npr = self.route
            # updatePtclPositions rides in the records / the rebuild, the two scatters behind it
            sent, _ = self.ppdist.migrate(capi, self.ps, ne_, npr, self.rank, self.world, commit=True,
                                          scatter=(self.mesh, [self.fwd, self.bkwd], [self.w_f, self.w_b]))
            self.moved += sent
            if self.world > 1:  # gyroSync: SUM over ranks of the interleaved fields
                self._allreduce_fields()
            cap = max(self.ps.capacity(), 1)
            if cap > self.ids.n:  # 10% slack: the capacity wanders by a few chunk widths per rebuild
                self.ids = capi.DevArray(cap + cap // 10, np.int32)
            self.ids.fill_bytes(0xff)
        # "2d": search_mesh_2d re-seeds from the previous ids as given

    def _allreduce_fields(self):
        import ctypes
        import torch
        nv = self.mesh.nverts
        if not hasattr(self, "sync_t"):
            self.sync_t = torch.empty(2 * nv, dtype=torch.float64, device="cuda")
        # pack both fields into one torch tensor on the device, reduce, leave the sum there
        self.capi.check(self.capi.lib().pp_gyro_sync_pack(nv, self.w_f.ptr, self.w_b.ptr,
                                                          ctypes.c_void_p(self.sync_t.data_ptr())))
        self.capi.sync()
        self.ppdist.allreduce_sum(self.sync_t)

    def kernel_avg_ms(self):
        return float(np.mean([a.elapsed_ms(b) for a, b in self.kernel_ms])) if self.kernel_ms else None


def cpu_baseline(pp, w, name, deg, sample, steps=20):
    """The oracle (restated reference, Kokkos::Serial semantics: C=1, unfused kernels, one pass
    per kernel per walk iteration) timed on one host core on a bounded sample of the workload."""
    ppo = pumipic_amd_loader.load_oracle()
    s = pp.synth
    idx = np.sort(np.random.default_rng(0).choice(len(w["elem"]), size=sample, replace=False))
    elem = w["elem"][idx]
    ppe = np.bincount(elem, minlength=w["ne"]).astype(np.int32)
    info = [np.ascontiguousarray(a[..., idx]) for a in w["info"]]
    mesh = ppo.Mesh(w["dim"], w["coords"], w["e2v"], w["cls"])
    ps = ppo.PS.scs(ppo.PARTICLE_XGCM, w["ne"], ppe, C_max=1, particle_elements=elem,
                    particle_info=info)
    ids = None
    t0 = time.perf_counter()
    for _ in range(steps):
        if w["dim"] == 3:
            ppo.toroidal_push(ps, mesh, s.XGC_H, s.XGC_K, s.XGC_D, deg, trig=0)
            ids = ppo.search_mesh(mesh, ps, elem_ids=ids, looplimit=200)["elem_ids"]
            a, b = ps.member(0), ps.member(1)
            tmp = a.copy()
            a[:] = b
            b[:] = tmp
        else:
            ppo.elliptical_push(ps, mesh, s.XGC_H, s.XGC_K, s.XGC_D, deg, trig=0)
            _, ids, _ = ppo.search_mesh_2d(mesh, ps, elem_ids=ids, looplimit=200)
    dt = time.perf_counter() - t0
    out = dict(value=sample * steps / dt, unit="particles/s", cores=1, kind="port",
               sample="%d particles x %d steps of the same mesh/push, oracle (C=1 Serial semantics, "
                      "libm trig), 1 core" % (sample, steps))
    # the same loop with its per-particle loops spread over all host cores (OpenMP)
    nthr = ppo.max_threads()
    if nthr > 1:
        ppo.set_threads(nthr)
        try:
            t0 = time.perf_counter()
            for _ in range(steps):
                if w["dim"] == 3:
                    ppo.toroidal_push(ps, mesh, s.XGC_H, s.XGC_K, s.XGC_D, deg, trig=0)
                    ids = ppo.search_mesh(mesh, ps, elem_ids=ids, looplimit=200)["elem_ids"]
                    a, b = ps.member(0), ps.member(1)
                    tmp = a.copy()
                    a[:] = b
                    b[:] = tmp
                else:
                    ppo.elliptical_push(ps, mesh, s.XGC_H, s.XGC_K, s.XGC_D, deg, trig=0)
                    _, ids, _ = ppo.search_mesh_2d(mesh, ps, elem_ids=ids, looplimit=200)
            dt = time.perf_counter() - t0
        finally:
            ppo.set_threads(1)
        out["all_cores"] = dict(value=sample * steps / dt, unit="particles/s", cores=nthr,
                                note="same sample, per-particle loops under OpenMP; the position "
                                     "swap and the slot tables stay serial")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5", "2d", "2dc3"])
    ap.add_argument("--sigma", type=int, default=2**31 - 1,
                    help="SCS sorting window (elements); the pseudoXGCm value is INT_MAX = full sort")
    ap.add_argument("--safe-layers", type=int, default=0,
                    help="c5: breadth-first element layers around the owned block that are still safe "
                         "(0 = BASELINE's rule: a particle migrates as soon as it leaves its owner's block)")
    ap.add_argument("--c4-elems", type=int, default=1_000_000, help="c4: number of elements")
    ap.add_argument("--structure", default="scs", choices=["scs", "csr"], help="c4: particle structure")
    ap.add_argument("--particles", type=int, default=10_000_000, help="particles per GPU")
    ap.add_argument("--deg", type=float, default=0.5, help="degrees per push (testing.cmake:117)")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000,
                    help="particles of the bounded CPU-baseline sample (x 20 steps, ~10 s on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mesh", default="100k", choices=["100k", "1m"],
                    help="3-D mesh: 100 800 tets (configs[1-2]) or 998 400 tets (configs[4], per-GPU share)")
    ap.add_argument("--remainder", default="last", choices=["last", "spread"],
                    help="where particles left over by the Gaussian draws go: 'last' = literal "
                         "pseudoXGCm rule (one outlier element), 'spread' = evenly")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    pp = pumipic_amd_loader.load()
    from pumipic_amd import capi
    capi.init(local_rank)  # raises when the HIP library / GPU is missing: no CPU fallback

    if a.workload == "c4":
        if "--particles" not in sys.argv:
            a.particles = 1_000_000  # configs[3] stress point: 1 M elements / 1 M particles
        w = build_c4(pp, capi, a.c4_elems, a.particles, rank, a.structure)
        st = StepperC4(capi, w)
    else:
        w = build_workload(pp, capi, a.workload, a.particles, rank, world, a.deg, a.remainder, a.mesh,
                           a.sigma)
        w["safe_layers"] = a.safe_layers
        st = Stepper(pp, capi, w, a.workload, a.deg)

    def barrier():
        capi.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    st.sample_every = max(1, a.steps // 10)
    # Clock pre-warm.  The GPU's power management needs ~0.1 s of sustained vector-ALU load before the
    # shader clock settles; W = 3 warm-up steps are 1 ms.  Measured on MI355X: c2 0.307 ms per step
    # from idle clocks, 0.270 ms after 0.3 s of ANY FP64 kernel (a memset loop, which only loads the
    # memory side, gets half of that), the same as after 100+ warm-up steps.  The pre-warm runs an
    # unrelated kernel (closest point on a triangle, scratch data): the workload's state is untouched
    # and still gets exactly W warm-up and K timed steps.  PP_BENCH_PREWARM=<seconds> (0 = off).
    prewarm_s = float(os.environ.get("PP_BENCH_PREWARM", "0.3"))
    if prewarm_s > 0:
        n = 1 << 22
        rng = np.random.default_rng(0)
        tri = capi.DevArray.from_host(rng.normal(size=9))
        pts = capi.DevArray.from_host(rng.normal(size=3 * n))
        scratch = capi.DevArray(3 * n, np.float64)
        t_end = time.perf_counter() + prewarm_s
        while time.perf_counter() < t_end:
            for _ in range(20):
                capi.check(capi.lib().pp_closest_point_on_triangle(n, tri.ptr, 0, pts.ptr, 0, scratch.ptr, None))
            capi.sync()
        del tri, pts, scratch
    for _ in range(a.warmup):
        st.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.step(timed=True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    nlive = w["ps"].nPtcls()
    total_particles = nlive
    if dist is not None:
        t = torch.tensor([nlive], device="cuda", dtype=torch.int64)
        dist.all_reduce(t)
        total_particles = int(t.item())

    if rank == 0:
        kms = st.kernel_avg_ms()
        # HBM bytes per launch from the PMC counters: collected with rocprofv3 in separate --pmc
        # passes of THIS command (tools/r01_measure.sh) and calibrated as DESIGN.md section 4 says;
        # a profiler cannot wrap itself, so the committed summary is reported with its provenance
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % a.workload)
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if (tj.get("particles") == a.particles and tj.get("remainder", "last") == a.remainder
                        and a.mesh == "100k" and a.sigma >= 2**31 - 1 and world == 1
                        and (a.workload != "c4" or a.c4_elems == 1_000_000)):
                    traffic = tj["traffic_bytes_per_step"]
            except (ValueError, KeyError):
                traffic = None
        bpp = BYTES[{"2dc3": "2d", "c3": "c2", "c5": "c2"}.get(a.workload, a.workload)]
        if a.workload == "c4" and a.structure != "scs":
            traffic = None  # the committed PMC run is the SCS structure
        if a.workload == "c4":
            out = {
                "metric": "particles pseudo-pushed+redistributed+rebuilt / sec / GPU; achieved HBM GB/s vs peak",
                "value": total_particles * a.steps / dt, "unit": "particles/s", "n_gpus": world,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic", "clock_prewarm_s": prewarm_s,
                "config": {"workload": "%s, %d particles/GPU, uniform distribution, pseudo-push + "
                                       "redistribute(0.5) + rebuild per step" % (w["label"], a.particles),
                           "parallelism": "%d independent rank(s)" % world},
                "roofline": {"bound": "hbm", "achieved": bpp * nlive / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": bpp * nlive / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": traffic,
                             "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, calibrated; "
                                             "profiles/traffic_c4.json)",
                             "kernel": "k_pseudo_push160", "kernel_ms": kms,
                             "bytes_per_particle": bpp},
            }
            if not a.no_cpu_baseline and world == 1:
                out["cpu_baseline"] = cpu_baseline_c4(pp, a.c4_elems, a.particles, min(a.particles, 1_000_000))
            print(json.dumps(out))
            if dist is not None:
                dist.destroy_process_group()
            return
        achieved = bpp * nlive / (kms * 1e-3) / 1e9 if kms else None
        out = {
            "metric": "particles pushed+searched+scattered / sec / GPU; achieved HBM GB/s vs peak",
            "value": total_particles * a.steps / dt,
            "unit": "particles/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "clock_prewarm_s": prewarm_s,
            "config": {"workload": "%s, %d particles/GPU, SCS C=64 sigma=%s V=1024, %s" % (
                w["label"], a.particles, "inf" if a.sigma >= 2**31 - 1 else str(a.sigma),
                {"c2": "push+search only (fused toroidal push + BCC walk), deg/push=%g" % a.deg,
                 "2d": "elliptical push + search_mesh_2d (fused), deg/push=%g" % a.deg,
                 "c3": "push+search+rebuild+gyroScatter x2 (tet ring map), deg/push=%g" % a.deg,
                 "c5": "push+search+migrate(all-to-all-v)+rebuild+gyroScatter x2+gyroSync, deg/push=%g" % a.deg,
                 "2dc3": "push+search+rebuild+gyroScatter x2, deg/push=%g" % a.deg}[a.workload]),
                "parallelism": "element-block partition, %d rank(s), full-mesh replica" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, calibrated; "
                                         "profiles/traffic_%s.json)" % a.workload,
                         "kernel": ("k_push_walk_rowsq<3> + k_walk_pending<3> (one pp_push_search call)"
                                    if w["dim"] == 3 else "k_push_walk_rows<2>"), "kernel_ms": kms,
                         "bytes_per_particle": bpp},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pp, w, a.workload, a.deg, min(a.cpu_sample, a.particles))
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
