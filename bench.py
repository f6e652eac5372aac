#!/usr/bin/env python
"""bench.py -- particles pushed+searched(+scattered+rebuilt) per second on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).
`--gpus N` with WORLD_SIZE unset starts N rank processes itself (before anything touches the GPU)
and fails loudly when the box has fewer than N GPUs; under torchrun (WORLD_SIZE set) every rank
runs this file directly.  Default workload: N = 1 -> c3 (the configuration the metric names),
N > 1 -> c5 (the migrating one, BASELINE configs[4] shares: 998 400 tets, 32 M particles per GPU).
A "step" is one pass of the hot path over the resident particle population:

  c2 (BASELINE.json configs[1]): fused toroidal push + BCC adjacency walk on the
      100 800-tet tokamak mesh, 10 M particles per GPU, SCS layout (C=64), no rebuild; positions
      ping-pong x <-> x_tgt and the walk is re-seeded from the previous step's element ids.
  c3 (configs[2], default at N = 1): c2 + updatePtclPositions + SCS rebuild + gyroScatter x2 every step (tet variant
      of the ring map: 4 vertices per ring point); 2dc3 is the 2-D literal of it.  The
      three calls go through pp_ps_rebuild_scatter (same work, one entry point;
      PP_BENCH_SEPARATE_SCATTER=1 issues them separately).
  c2mt: configs[1] in the reference's INTERSECTION mode (search_mesh with requireIntersection, adjacency.tpp:284-361:
      Moeller-Trumbore per face): toroidal push, then every particle's ray is followed element by element to
      the domain boundary (~tens of tets per particle; exit face + intersection point written).  The same rays
      every step (positions are not committed).  Walk-bound, not stream-bound: the line also reports elements
      visited per particle and the time per visited element.
  2d : the literal 2-D pseudoXGCm step (elliptical push + search_mesh_2d) on 100 352 triangles.
  c5 (configs[4], default at N > 1): c3 with ownership: every rank owns a block of elements; after the search
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
COPY_CEILING_GBS = 6290.0  # same guide: measured float4 copy

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
    # read x(24) b,phi(8) mask(1) ; write x_tgt(24) phi(4) elem(4) inter_face(4) inter_point(24)
    "c2mt": 93.0,
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
        try:  # (not for the 32 M-particle populations: 1.2 GB per rank, possibly on a memory-backed /tmp)
            if nptcl <= 16_000_000 or os.environ.get("PP_BENCH_CACHE"):
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


def build_c4(pp, capi, ne, nptcl, rank, structure, dist=1):
    """ps_combo160 set-up (performance_tests/ps_combo160.cpp:60-130): distribution strategy `dist` (1 uniform,
    2 gaussian, 3 exponential, 4 GITRm approximation; fixed seed instead of the wall clock), Sell-64-ne
    (sigma = ne, V = 1024) or CSR."""
    if dist == 1:
        rng = np.random.default_rng(rank)
        elems = np.sort(rng.integers(0, ne, size=nptcl).astype(np.int32))
        ppe = np.bincount(elems, minlength=ne).astype(np.int32)
    else:
        ppe, elems = pp.synth.distribute_particles(ne, nptcl, dist, seed=rank)
        elems = np.sort(elems)
    info = [np.zeros((17, nptcl)), np.zeros((4, nptcl), dtype=np.int32),
            np.arange(nptcl, dtype=np.int64)[None, :]]
    if structure == "scs":
        ps = capi.PS.scs(capi.PERF160, ne, ppe, C_=64, sigma=ne, V=1024, particle_elements=elems,
                         particle_info=info)
    else:
        ps = capi.PS.csr(capi.PERF160, ne, ppe, particle_elements=elems, particle_info=info)
    parent = capi.DevArray.from_host(np.sqrt(np.arange(ne, dtype=np.float64)) * np.arange(ne))
    names = {1: "uniform", 2: "gaussian", 3: "exponential", 4: "GITRm-like"}
    return dict(ps=ps, parent=parent, ne=ne, dim=0, rank=rank, world=1, dist=dist,
                label="ps_combo160 %s, %d elements, %s distribution" % (
                    "Sell-64-ne" if structure == "scs" else "CSR", ne, names[dist]))


class StepperC4:
    def __init__(self, capi, w):
        self.capi, self.ps, self.parent = capi, w["ps"], w["parent"]
        self.dist = w.get("dist", 1)
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
        self.new_elems = capi.redistribute_particles(self.ps, 0.5, seed=self.round, out=self.new_elems,
                                                     strat=self.dist)
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
        self.steps_done = 0
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
            self.comm, self.comm_kind = None, w.get("comm", "rccl")
            if self.comm_kind == "tcp":  # the library's host-staged transport (rehearsal on a smaller box)
                port = int(os.environ.get("PP_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
                self.comm = (capi.Comm.env() if self.world == 1 else
                             capi.Comm.tcp(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, self.rank, self.world))
            if self.comm_kind == "rccl":
                try:
                    beat("RCCL communicator")
                    self.comm = self._make_comm()
                except Exception as e:  # noqa: BLE001 -- keep the run alive on the host-staged transport, loudly
                    sys.stderr.write("bench.py: RCCL communicator behind the C-ABI failed (%r); falling back to the "
                                     "library's host-staged TCP transport (--comm tcp): NOT an xGMI measurement\n" % (e,))
                    self.comm_kind = "tcp"
                    port = int(os.environ.get("PP_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
                    self.comm = capi.Comm.tcp(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, self.rank, self.world)
            # pre-flight: the exchange and the all-reduce of a migration step with known contents, checked
            # word by word, BEFORE anything is timed (pp_comm_selftest).  A failure here is a failure of
            # the run: the exception leaves main() with a non-zero exit code.
            self.preflight = None
            if self.comm is not None and self.world > 1:
                beat("pre-flight exchange")
                t0 = time.perf_counter()
                everywhere = self._selftest_everywhere()
                if not everywhere and self.comm_kind != "rccl":
                    raise RuntimeError("the pre-flight exchange failed on some rank (transport %s)" % self.comm_kind)
                if not everywhere:
                    # the RCCL data path returned an error or wrong contents on some rank (and did not hang -- a hang
                    # is the watchdog's): every rank moves to the host-staged transport together, loudly
                    sys.stderr.write("bench.py: the pre-flight exchange over RCCL failed on some rank; falling back to "
                                     "the library's host-staged TCP transport: NOT an xGMI measurement\n")
                    self.comm.destroy()
                    self.comm_kind = "tcp"
                    port = int(os.environ.get("PP_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
                    self.comm = capi.Comm.tcp(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, self.rank, self.world)
                    self.comm.selftest(5)
                self.preflight = {"ok": True, "seconds": time.perf_counter() - t0, "transport": self.comm.kind(),
                                  "what": "5-7 records of 80 B to every peer through the migration's count "
                                          "exchange + grouped send/recv, then the gyroSync all-reduce, contents "
                                          "checked on every rank (pp_comm_selftest)"}
        if name == "c2mt":
            self.xface = capi.DevArray(cap + cap // 10, np.int32)
            self.xpts = capi.DevArray(3 * (cap + cap // 10), np.float64)
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
        whole = True
        if timed:
            # a sampled step is bracketed EITHER as a whole (e0 .. e2) OR around its pp_push_search call (e0 .. e1),
            # alternately: an event record is a barrier packet (~6 us), and one in the middle of a span would be
            # measured by it
            whole = len(self.kernel_ms) % 2 == 0 or self.name in ("c2", "2d", "c2mt")
            e0, e1, e2 = capi.Event(), capi.Event(), capi.Event()
            e0.record()
        beat("step %d: push + search" % self.steps_done)
        if self.steps_done == 1 and os.environ.get("PP_BENCH_STALL_RANK") == str(self.w.get("rank", 0)):
            time.sleep(3600)  # test hook (tests/test_gpu_comm.py): this rank hangs; the watchdogs end the job
        if self.name == "c2mt":
            capi.toroidal_push(self.ps, self.mesh, self.h, self.k, self.d, self.deg)
            capi.check(capi.lib().pp_search_mesh(self.mesh.p, self.ps.p, 0, 1, 2, self.ids.ptr, 0, 1,
                                                 self.xface.ptr, self.xpts.ptr, 2000, None, None))
        elif self.name == "c2":
            capi.push_search(self.mesh, self.ps, self.h, self.k, self.d, self.deg, self.ids,
                             seeded=not self.first, looplimit=200, want_found=False)
        elif self.name in ("c3", "c5"):
            # rebuilt every step: every particle sits in its row's element, no seed ids needed
            capi.push_search(self.mesh, self.ps, self.h, self.k, self.d, self.deg, self.ids,
                             seeded=False, looplimit=200, want_found=False)
        elif self.name == "2dc3":
            # rebuilt every step: no seeds = every particle starts in its row's element (the -1 seeds
            # search_mesh_2d would read, adjacency.hpp:1051-1056) -- no fill of the ids per step
            capi.push_search(self.mesh, self.ps, self.h, self.k, self.d, self.deg, self.ids,
                             seeded=False, looplimit=200, want_found=False)
        else:
            capi.push_search(self.mesh, self.ps, self.h, self.k, self.d, self.deg, self.ids,
                             seeded=True, looplimit=200, want_found=False)
        if timed and (not whole or self.name in ("c2", "2d", "c2mt")):
            e1.record()
        else:
            e1 = None
        if self.first and self.w.get("origin_trust", False) and self.w["dim"] == 3 and self.name not in ("c4", "c2mt"):
            # From the second step on every origin is the destination the previous walk accepted in the
            # element the walk starts from (c3 / c5: the structure was rebuilt from the ids; c2: the ids are
            # the seeds): pp_ps_set_origin_trust skips check_initial_parents (include/pumipic_hip.h).
            self.ps.set_origin_trust(True)
        self.first = False
        self.steps_done += 1
        beat("step %d: rebuild / migration / scatter" % (self.steps_done - 1))
        self._rest_of_step()
        if timed:
            if whole:
                e2.record()
            else:
                e2 = None
            self.kernel_ms.append((e0, e1, e2))

    def _rest_of_step(self):
        capi = self.capi
        if self.name == "c2mt":
            pass  # the rays end at the wall: nothing is committed, every step follows the same rays
        elif self.name == "c2":
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
            # the search's `found` (every search of the reference returns it, the driver asserts it:
            # test/pseudoXGCm.cpp:153-154): it came to the host with the rebuild's totals -- no wait of its own
            if not capi.last_search_found(self.ps):
                raise RuntimeError("search: particles were cut off by the loop limit")
            cap = max(self.ps.capacity(), 1)
            if cap > self.ids.n:  # 10% slack: the capacity wanders by a few chunk widths per rebuild
                self.ids = capi.DevArray(cap + cap // 10, np.int32)
            # unseeded (an "empty elem_ids", adjacency.tpp:504-515; 2-D: every seed -1) (an "empty elem_ids", adjacency.tpp:504-515): the search writes every
            # slot itself, -1 into the masked ones -- no fill
        elif self.name == "c5":
            # updatePtclPositions rides in the records / the rebuild, the two scatters behind it
            scat = (self.mesh, [self.fwd, self.bkwd], [self.w_f, self.w_b])
            if self.comm is not None:
                # migrate_lb_ptcls + gyroSync behind the C-ABI (RCCL): setUnsafeProcs is the routing rule
                # of the pack, SellCSigma::migrate, reduceCommArray
                sent, _ = capi.migrate_ptcls(self.ps, self.ids, self.safe, self.owners, self.comm, commit=True,
                                             scatter=scat)
                if self.world > 1:  # gyroSync: SUM over ranks of the interleaved fields
                    self.sync_d = capi.gyro_sync_pack(self.mesh.nverts, self.w_f, self.w_b,
                                                      out=getattr(self, "sync_d", None))
                    self.comm.allreduce_sum(self.sync_d)
            else:
                self.route = capi.set_unsafe_procs(self.ps, self.ids, self.safe, self.owners, self.rank,
                                                   out=getattr(self, "route", None))
                ne_, npr = self.route
                sent, _ = self.ppdist.migrate(capi, self.ps, ne_, npr, self.rank, self.world, commit=True,
                                              scatter=scat)
                if self.world > 1:
                    self._allreduce_fields()
            self.moved += sent
            if not capi.last_search_found(self.ps):  # (as above: delivered with the migration's rebuild totals)
                raise RuntimeError("search: particles were cut off by the loop limit")
            cap = max(self.ps.capacity(), 1)
            if cap > self.ids.n:  # 10% slack: the capacity wanders by a few chunk widths per rebuild
                self.ids = capi.DevArray(cap + cap // 10, np.int32)
            if self.w["dim"] == 2:
                self.ids.fill_bytes(0xff)  # (dim 3, unseeded: the search writes every slot itself)
        # "2d": search_mesh_2d re-seeds from the previous ids as given

    def _selftest_everywhere(self):
        """pp_comm_selftest on this rank, then the verdict of ALL ranks over the host-side control plane"""
        import torch
        import torch.distributed as dist
        ok = 1
        try:
            self.comm.selftest(5)
        except self.capi.PPError as e:
            sys.stderr.write("bench.py: rank %d: pre-flight exchange failed: %s\n" % (self.rank, e))
            ok = 0
        if not (dist.is_available() and dist.is_initialized()):
            if not ok:
                raise RuntimeError("pre-flight exchange failed and there is no control plane to agree on a fallback")
            return True
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def _make_comm(self):
        """pp_comm over RCCL: rank 0 draws the id, torch.distributed (the host-side control plane) broadcasts
        it -- the MPI_Bcast a PUMI-PIC build would do.  Every rank reaches the same verdict: a failure on ANY
        rank (no id, communicator not formed) raises on ALL of them, so the fallback is taken together."""
        capi = self.capi
        if self.world == 1:
            return capi.Comm.env()
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.zeros(129, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            try:
                t[1:] = torch.tensor(list(capi.Comm.unique_id()), dtype=torch.uint8)
                t[0] = 1
            except capi.PPError as e:
                sys.stderr.write("bench.py: rank 0 could not draw an RCCL id: %s\n" % e)
        dist.broadcast(t, 0)
        t = t.cpu()
        comm, err = None, None
        if int(t[0]) == 1:
            try:
                comm = capi.Comm.rccl(bytes(t[1:].tolist()), self.rank, self.world)
            except capi.PPError as e:
                err = e
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.destroy()
            raise RuntimeError("the RCCL communicator was not formed on every rank (this rank: %s)"
                               % (err if err is not None else "ok" if comm is not None else "no id"))
        return comm

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
        """median HIP-event time of one pp_push_search call (both kernels) over the sampled steps"""
        ms = [a.elapsed_ms(b) for a, b, _ in self.kernel_ms if b is not None]
        return float(np.median(ms)) if ms else None

    def step_avg_ms(self):
        """median HIP-event time of the whole step on the library stream (the sampled steps)"""
        ms = [a.elapsed_ms(c) for a, _, c in self.kernel_ms if c is not None]
        return float(np.median(ms)) if ms else None


def cpu_baseline(pp, w, name, deg, sample, steps=20):
    """The oracle (restated reference, Kokkos::Serial semantics: C=1, unfused kernels, one pass
    per kernel per walk iteration, reshuffle-then-rebuild as SCS_rebuild.h:160-189) timed on one host
    core on a bounded sample of the workload: the same step the GPU line times (c2: push + search;
    c3 / 2dc3 / c5: + updatePtclPositions + rebuild + gyroScatter x2)."""
    ppo = pumipic_amd_loader.load_oracle()
    s = pp.synth
    idx = np.sort(np.random.default_rng(0).choice(len(w["elem"]), size=sample, replace=False))
    elem = w["elem"][idx]
    ppe = np.bincount(elem, minlength=w["ne"]).astype(np.int32)
    info = [np.ascontiguousarray(a[..., idx]) for a in w["info"]]
    mesh = ppo.Mesh(w["dim"], w["coords"], w["e2v"], w["cls"])
    full = name in ("c3", "2dc3", "c5")
    maps = ppo.create_gyro_ring_mappings(mesh) if full else None

    def make_ps():
        return ppo.PS.scs(ppo.PARTICLE_XGCM, w["ne"], ppe, C_max=1, pad_strat=0, shuffle_padding=0.1,
                          extra_padding=0.0, particle_elements=elem, particle_info=info)

    def run(ps):
        ids = None
        t0 = time.perf_counter()
        for _ in range(steps):
            if name == "c2mt":
                ppo.toroidal_push(ps, mesh, s.XGC_H, s.XGC_K, s.XGC_D, deg, trig=0)
                ppo.search_mesh(mesh, ps, require_intersection=True, looplimit=2000)
                continue
            if w["dim"] == 3:
                ppo.toroidal_push(ps, mesh, s.XGC_H, s.XGC_K, s.XGC_D, deg, trig=0)
                ids = ppo.search_mesh(mesh, ps, elem_ids=None if full else ids, looplimit=200)["elem_ids"]
            else:
                ppo.elliptical_push(ps, mesh, s.XGC_H, s.XGC_K, s.XGC_D, deg, trig=0)
                _, ids, _ = ppo.search_mesh_2d(mesh, ps, elem_ids=None if full else ids, looplimit=200)
            if full:  # the drivers' rebuild() + the two scatters (pseudoXGCm.cpp:116-140, 529-530)
                ppo.update_positions(ps)
                ps.rebuild(ids)
                ppo.gyro_scatter(mesh, ps, maps[0])
                ppo.gyro_scatter(mesh, ps, maps[1])
            elif w["dim"] == 3:
                a, b = ps.member(0), ps.member(1)
                tmp = a.copy()
                a[:] = b
                b[:] = tmp
        return time.perf_counter() - t0

    dt = run(make_ps())
    what = ("push + search + updatePtclPositions + rebuild + gyroScatter x2" if full else
            "push + search_mesh in intersection mode (rays to the wall)" if name == "c2mt" else "push + search")
    out = dict(value=sample * steps / dt, unit="particles/s", cores=1, kind="port",
               sample="%d particles x %d steps (%s) of the same mesh/push, oracle (C=1 Serial semantics, "
                      "libm trig), 1 core" % (sample, steps, what))
    # the same loop with its per-particle loops spread over all host cores (OpenMP)
    nthr = ppo.max_threads()
    if nthr > 1:
        ppo.set_threads(nthr)
        try:
            dt = run(make_ps())
        finally:
            ppo.set_threads(1)
        out["all_cores"] = dict(value=sample * steps / dt, unit="particles/s", cores=nthr,
                                note="same sample, per-particle loops under OpenMP; the structure "
                                     "rebuild and the slot tables stay serial")
    return out


# ---------------------------------------------------------------------------------------------------
# `also`: the other single-GPU configurations of BASELINE.json and the pieces the headline's step does not
# exercise, measured in the same run (N = 1, default c3 line only).  Each returns a small dict; bench.py's
# main() guards every call.
def clock_prewarm(capi, seconds):
    """~0.1 s of FP64 vector load settles the shader clock; an unrelated kernel on scratch data"""
    if seconds <= 0:
        return
    n = 1 << 22
    rng = np.random.default_rng(0)
    tri = capi.DevArray.from_host(rng.normal(size=9))
    pts = capi.DevArray.from_host(rng.normal(size=3 * n))
    scratch = capi.DevArray(3 * n, np.float64)
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(20):
            capi.check(capi.lib().pp_closest_point_on_triangle(n, tri.ptr, 0, pts.ptr, 0, scratch.ptr, None))
        capi.sync()


def _time_steps(capi, st, warm, k):
    import gc
    gc.collect()  # (finalisers of set-up temporaries -- hipFree -- now, not inside the timed steps)
    clock_prewarm(capi, float(os.environ.get("PP_BENCH_PREWARM", "0.3")))  # (the host just built a population)
    for _ in range(warm):
        st.step()
    capi.sync()
    t0 = time.perf_counter()
    for _ in range(k):
        st.step()
    capi.sync()
    return (time.perf_counter() - t0) / k


def also_c2(pp, capi, a, w_main, st_main):
    """BASELINE configs[1]: push + search only (fused toroidal push + BCC walk), same mesh and population"""
    w = build_workload(pp, capi, "c2", a.particles, 0, 1, a.deg, a.remainder, "100k", a.sigma)
    w["origin_trust"] = w_main.get("origin_trust", False)
    st = Stepper(pp, capi, w, "c2", a.deg)
    dt = _time_steps(capi, st, 12, 20)
    n = w["ps"].nPtcls()
    out = {"workload": "configs[1]: %s, %d particles, push+search only, ids re-used as seeds; the search's `found` flag is "
                       "NOT read in the timed step (want_found=False and config 2 has no rebuild whose totals could carry it "
                       "to the host: a host read would add one sync per step)" % (w["label"], n),
           "ms_per_step": dt * 1e3, "value": n / dt, "unit": "particles/s", "steps": 20, "warmup": 12,
           "bytes_per_particle": BYTES["c2"], "roofline_frac": BYTES["c2"] * n / dt / 1e9 / HBM_PEAK_GBS}
    # Config 2 never rebuilds: with every step more particles have left the element of their row, the walk starts
    # from per-particle seed records instead of the row's one, and the SAME call gets slower.  HIP events around
    # every step of a fresh structure show the whole curve (step 1: every particle in its row's element, no seeds).
    del st, w
    w = build_workload(pp, capi, "c2", a.particles, 0, 1, a.deg, a.remainder, "100k", a.sigma)
    st = Stepper(pp, capi, w, "c2", a.deg)
    st.sample_every = 1
    clock_prewarm(capi, float(os.environ.get("PP_BENCH_PREWARM", "0.3")))
    for _ in range(32):
        st.step(timed=True)
    capi.sync()
    ms = [e0.elapsed_ms(e1) for e0, e1, _ in st.kernel_ms]
    out["ms_of_step_on_a_fresh_structure"] = {str(i): ms[i - 1] for i in (1, 2, 3, 4, 8, 16, 24, 32)}
    out["roofline_frac_fastest_step"] = BYTES["c2"] * n / min(ms) / 1e6 / HBM_PEAK_GBS
    out["note"] = ("no rebuild in config 2: the structure loses its element locality step by step (the curve above: "
                   "a HIP event pair around every step, which adds ~10 us to each); ms_per_step is wall clock over "
                   "steps 13-32 of the first structure")
    return out


def also_c2mt(pp, capi, a, w_main, st_main):
    """configs[1] in the reference's intersection mode (Moeller-Trumbore, rays followed to the wall)"""
    w = build_workload(pp, capi, "c2mt", a.particles, 0, 1, a.deg, a.remainder, "100k", a.sigma)
    st = Stepper(pp, capi, w, "c2mt", a.deg)
    dt = _time_steps(capi, st, 2, 4)
    n = w["ps"].nPtcls()
    nsteps = capi.search_walk_steps()
    return {"workload": "configs[1], search_mesh with requireIntersection: %s, %d particles, toroidal push + "
                        "Moeller-Trumbore walk of every ray to the domain boundary" % (w["label"], n),
            "ms_per_step": dt * 1e3, "value": n / dt, "unit": "particles/s", "steps": 4, "warmup": 2,
            "elements_visited_per_particle": nsteps / max(n, 1),
            "visited_elements_per_s": nsteps / dt,
            "bytes_per_particle": BYTES["c2mt"], "roofline_frac": BYTES["c2mt"] * n / dt / 1e9 / HBM_PEAK_GBS,
            "note": "walk-bound: one visit = one 128-B record (L2 / Infinity Cache) + 3-4 ray/triangle tests in "
                    "FP64; the HBM fraction is reported for completeness, it is not what bounds this search"}


def also_c4(pp, capi, a, w_main, st_main):
    """BASELINE configs[3]: ps_combo160 largeE_smallP, 1 M elements / 1 M particles, Sell-64-ne"""
    w = build_c4(pp, capi, 1_000_000, 1_000_000, 0, "scs", 1)
    st = StepperC4(capi, w)
    dt = _time_steps(capi, st, 8, 30)
    n = w["ps"].nPtcls()
    st.kernel_ms, st.ntimed, st.sample_every = [], 0, 1
    for _ in range(8):
        st.step(timed=True)
    kms = st.kernel_avg_ms()
    return {"workload": "configs[3]: %s, %d particles, pseudo-push + redistribute(0.5) + rebuild per step"
                        % (w["label"], n),
            "ms_per_step": dt * 1e3, "value": n / dt, "unit": "particles/s", "steps": 30, "warmup": 8,
            "pseudo_push_ms": kms,
            "pseudo_push_roofline_frac": BYTES["c4"] * n / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms else None,
            "step_roofline_frac": (BYTES["c4"] + 328.0) * n / dt / 1e9 / HBM_PEAK_GBS,
            "note": "step fraction on 161 (pseudo-push) + 328 (rebuild: 160 read + 160 written + new element + "
                    "mask) algorithmic B/particle; at one particle per element the step is a chain of kernels "
                    "that sweep 10^6 rows, not a stream"}


def measure_c4ref(pp, capi, ne, nptcl, structure="scs", dist=1, k=100):
    """ps_combo160 in the REFERENCE's own shape (performance_tests/ps_combo160.cpp:152-188, 205-232): K pseudo-pushes
    back to back, timed; then K x (redistribute_particles + rebuild), timed, with NO member access between the
    rebuilds -- the second pass of a re-layout (records -> member arrays) is deferred for records wider than 64 B
    and the next rebuild moves records to records (k_move_pack_rec: 192 B read + 192 B written per particle)."""
    import gc
    w = build_c4(pp, capi, ne, nptcl, 0, structure, dist)
    ps, parent = w["ps"], w["parent"]
    gc.collect()
    clock_prewarm(capi, float(os.environ.get("PP_BENCH_PREWARM", "0.3")))
    for _ in range(3):
        capi.pseudo_push160(ps, parent)
    capi.sync()
    t0 = time.perf_counter()
    for _ in range(k):
        capi.pseudo_push160(ps, parent)
    capi.sync()
    push_ms = (time.perf_counter() - t0) / k * 1e3
    new_elems = None

    def round_(r):
        nonlocal new_elems
        new_elems = capi.redistribute_particles(ps, 0.5, seed=r, out=new_elems, strat=dist)
        ps.rebuild(new_elems)
        cap = max(ps.capacity(), 1)
        if cap > new_elems.n:
            new_elems = capi.DevArray(cap + cap // 10, np.int32)
    for r in range(4):
        round_(r)
    capi.sync()
    t0 = time.perf_counter()
    for r in range(k):
        round_(4 + r)
    capi.sync()
    loop_ms = (time.perf_counter() - t0) / k * 1e3
    # the redistribution alone (reads the mask and the slot's element, writes 4 B per slot)
    t0 = time.perf_counter()
    for r in range(20):
        new_elems = capi.redistribute_particles(ps, 0.5, seed=r, out=new_elems, strat=dist)
    capi.sync()
    redis_ms = (time.perf_counter() - t0) / 20 * 1e3
    n = ps.nPtcls()
    rebuild_ms = loop_ms - redis_ms
    return {"workload": "%s, %d particles: %d pseudo-pushes, then %d x (redistribute 0.5 + rebuild) with no member "
                        "access in between (performance_tests/ps_combo160.cpp:152-188, 205-232)" % (w["label"], n, k, k),
            "particles": n, "elements": ne, "pseudo_push_ms": push_ms,
            "pseudo_push_roofline_frac": BYTES["c4"] * n / (push_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "redistribute_plus_rebuild_ms": loop_ms, "redistribute_ms": redis_ms, "rebuild_ms": rebuild_ms,
            "rebuild_roofline_frac": 328.0 * n / (rebuild_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "loop_roofline_frac": 328.0 * n / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "bytes_per_particle_rebuild": 328.0,
            "note": "fractions on SURVEY 8(d)'s 328 algorithmic B per kept-or-moved particle (160 read + 160 written + "
                    "8 index); the record-to-record pass moves 2 x 192 B of padded records for them"}


def also_c4ref(pp, capi, a, w_main, st_main):
    """configs[3] in the reference's two-loop shape: the stress point 1 M elements / 1 M particles and the script's
    own point 50 k elements / 50 M particles (performance_tests/test_largeE_smallP.sh:9-19), Sell-64-ne"""
    out = {"1Me_1Mp": measure_c4ref(pp, capi, 1_000_000, 1_000_000, "scs", 1, 100)}
    try:
        out["50ke_50Mp"] = measure_c4ref(pp, capi, 50_000, 50_000_000, "scs", 1, 20)
    except Exception as e:  # noqa: BLE001 -- (memory / time on a loaded box: the stress point stands on its own)
        out["50ke_50Mp"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def also_general_scatter(pp, capi, a, w_main, st_main):
    """pp_gyro_scatter_radius on the headline's structure: per-particle radius (and weight 1), the kernel the
    reference's gyroScatter would be without its constant radius (test/gyroScatter.hpp:182-204: 2*(dim+1) FP64
    atomics per particle there; here one per (element, ring) touched by a row's run)"""
    ps, mesh = w_main["ps"], w_main["mesh"]
    cap = max(ps.capacity(), 1)
    rng = np.random.default_rng(1)
    radius = capi.DevArray.from_host(rng.uniform(0.0, 0.038 * 0.6, size=cap))
    out = capi.DevArray(mesh.nverts, np.float64)
    clock_prewarm(capi, float(os.environ.get("PP_BENCH_PREWARM", "0.3")))
    for _ in range(3):
        capi.gyro_scatter_radius(mesh, ps, radius, st_main.fwd, out=out, want_clipped=False)
    capi.sync()
    k = 10
    t0 = time.perf_counter()
    for _ in range(k):
        capi.gyro_scatter_radius(mesh, ps, radius, st_main.fwd, out=out, want_clipped=False)
    capi.sync()
    dt = (time.perf_counter() - t0) / k
    n = ps.nPtcls()
    bpp = 8.0 + 1.0  # radius + mask per particle; the element's ring sums and the vertex field are O(mesh)
    return {"workload": "gyroScatter with a per-particle radius (pp_gyro_scatter_radius) on the headline's "
                        "structure: %d particles, one field" % n,
            "ms_per_call": dt * 1e3, "value": n / dt, "unit": "particles/s", "calls": k,
            "bytes_per_particle": bpp, "roofline_frac": bpp * n / dt / 1e9 / HBM_PEAK_GBS,
            "reference_atomics_per_particle": 8, "note": "the timed step's gyroScatter is the constant-radius form "
            "(a function of per-element counts, O(vertices)); this is the per-particle form beside it"}


def also_parallel_for(pp, capi, a, w_main, st_main):
    """the operator API itself: the reference driver's USER lambda through ps::parallel_for (C++ mirror header,
    drivers/ps_combo160.cpp) against the library's restatement of the same pass, 1 M / 1 M and 50 k / 50 M"""
    import subprocess
    drv = os.path.join(ROOT, "pumi-pic_amd", "drivers", "ps_combo160")
    if not os.path.exists(drv):
        subprocess.check_call(["make", "-C", os.path.dirname(drv), "-s"])
    out = {}
    for label, ne, npt in (("1Me_1Mp", 1_000_000, 1_000_000), ("50ke_50Mp", 50_000, 50_000_000)):
        env = dict(os.environ, PS_COMBO_CMP="10")
        r = subprocess.run([drv, str(ne), str(npt), "1", "0", "-i", "1"], env=env, capture_output=True, text=True,
                           timeout=120)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("PUSHCMP")]
        if r.returncode != 0 or not line:
            out[label] = {"error": "driver exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
            continue
        tok = line[0].split()
        val = lambda key: float(tok[tok.index(key) + 1])  # noqa: E731
        lam, libk, once = val("parallel_for_lambda_ms"), val("library_kernel_ms"), val("parallel_for_single_store_lambda_ms")
        n = int(tok[tok.index("particles") + 1])
        out[label] = {"particles": n, "elements": ne, "parallel_for_lambda_ms": lam, "library_kernel_ms": libk,
                      "parallel_for_single_store_lambda_ms": once,
                      "lambda_over_library": lam / libk if libk else None,
                      "single_store_lambda_over_library": once / libk if libk else None,
                      "lambda_roofline_frac": BYTES["c4"] * n / (lam * 1e-3) / 1e9 / HBM_PEAK_GBS if lam else None}
    out["note"] = ("pseudo-push of performance_tests/ps_combo160.cpp:158-183 as a user lambda through the mirror's "
                   "ps::parallel_for vs pp_pseudo_push160, HIP events around 10 back-to-back passes, Sell-64-ne; "
                   "the reference's lambda stores every double twice (10.3, then the value: the compiler must keep "
                   "the first store, parentElmData(e) is read in between) -- `single_store` is the same lambda "
                   "writing each value once, i.e. what ps::parallel_for itself costs")
    return out


def also_driver_pseudoxgcm(pp, capi, a, w_main, st_main):
    """The DROP-IN path: drivers/pseudoXGCm (the reference's test/pseudoXGCm.cpp:504-534 step loop on the mirror
    headers -- user-lambda push with the device libm through ps::parallel_for, search_mesh_2d, the
    updatePtclPositions lambda, migrate_lb_ptcls, tagParentElements, two gyroScatter calls, gyroSync) as a child
    process on the 2-D literal of configs[2] (100 352 triangles, 10 M particles), next to the fused two-call step
    of the same configuration (`2dc3`) measured here in the same process."""
    import re
    import subprocess
    import tempfile
    drv = os.path.join(ROOT, "pumi-pic_amd", "drivers", "pseudoXGCm")
    if not os.path.exists(drv):
        subprocess.check_call(["make", "-C", os.path.dirname(drv), "-s"])
    nptcl, iters = a.particles, 100  # (the first iterations grow the library's buffers and the pool: amortised)
    coords, e2v, cls = pp.synth.annulus_tri()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        mesh_file = os.path.join(tmp, "annulus.bin")
        pp.synth.write_mesh_bin(mesh_file, 2, coords, e2v, cls)
        cmd = [drv, mesh_file, str(nptcl), "12", str(iters), str(a.deg), "0"]
        for label, env in (("loop", {}), ("fenced", {"PP_TIMER_FENCE": "1"})):
            r = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
            m = re.search(r"(\d+) iterations of pseudopush \(seconds\) (\S+)", r.stderr)
            res = re.search(r"RESULT particles (\d+)", r.stdout)
            if r.returncode != 0 or not m or not res:
                return {"error": "driver exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
            ms = float(m.group(2)) / iters * 1e3
            if label == "loop":
                out["ms_per_step"] = ms
                out["particles"] = int(res.group(1))
                out["value"] = int(res.group(1)) / (ms * 1e-3)
                out["unit"] = "particles/s"
            else:
                out["ms_per_step_with_a_fence_per_recorded_operation"] = ms
                table = {}
                for ln in r.stderr.splitlines():
                    t = re.match(r"(\S.*?)\s+(\d+\.\d+) \(\s*\d+\)\s+\d+\.\d+ \(\s*\d+\)\s+\d+\.\d+\s+(\d+)\s*$", ln)
                    if t:
                        table[t.group(1).strip()] = {"ms_per_step": float(t.group(2)) / iters * 1e3, "calls": int(t.group(3))}
                out["record_time_ms_per_step"] = table
        # the reference's OWN source (test/pseudoXGCm.cpp with its ellipticalPush.hpp / gyroScatter.hpp, compiled
        # unchanged against the mirror headers by tools/ref_conformance.py where the reference tree exists): its
        # gyroScatter is the reference's user lambdas (24 double atomics per particle and map), not the library's
        ref_exe = os.path.join(ROOT, "tests", "_refdrivers", "pseudoXGCm")
        if os.path.exists(ref_exe):
            r = subprocess.run([ref_exe] + cmd[1:], capture_output=True, text=True, timeout=300)
            m = re.search(r"(\d+) iterations of pseudopush \(seconds\) (\S+)", r.stderr)
            if r.returncode == 0 and m:
                out["reference_source_ms_per_step"] = float(m.group(2)) / iters * 1e3
                out["reference_source"] = ("tests/_refdrivers/pseudoXGCm = /root/reference/test/pseudoXGCm.cpp byte for "
                                           "byte (user-lambda push AND user-lambda gyroScatter through ps::parallel_for)")
            else:
                out["reference_source_ms_per_step"] = None
                out["reference_source"] = "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-200:])
    # the fused step of the same configuration, same process
    w = build_workload(pp, capi, "2dc3", a.particles, 0, 1, a.deg, a.remainder, "100k", a.sigma)
    st = Stepper(pp, capi, w, "2dc3", a.deg)
    dt = _time_steps(capi, st, 6, 20)
    out["fused_2dc3_ms_per_step"] = dt * 1e3
    out["driver_over_fused"] = out["ms_per_step"] / (dt * 1e3)
    # what the UNFUSED loop must move per slot through user lambdas on the member arrays (algorithmic, 2-D):
    # push 8+1+4(class via element) read, 20 written; search 16+1+4 read, 4 written; updatePtclPositions 24 read, 48
    # written; setUnsafeProcs 4+1 read, 8 written; rebuild 60+4 read, 60+1 written (it takes two passes through 64-B
    # records: + 128 B that are not counted here); tagParentElements 1+4
    out["unfused_bytes_per_particle"] = 33 + 25 + 72 + 13 + 125 + 5
    out["roofline_frac_on_unfused_bytes"] = out["unfused_bytes_per_particle"] * out["particles"] / (out["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    tf = os.path.join(ROOT, "profiles", "traffic_driver.json")
    if os.path.exists(tf):
        try:  # HBM bytes of one step of this very command from the PMC counters (tools/r05_driver_pmc.sh)
            out["traffic_bytes_per_step"] = json.load(open(tf))["traffic_bytes_per_step"]
            out["traffic_provenance"] = "profiles/traffic_driver.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except (ValueError, KeyError):
            pass
    out["workload"] = ("drivers/pseudoXGCm <100352-tri annulus> %d 12 %d %g 0: whole-loop wall clock / iterations "
                       "(Kokkos::Timer semantics: no fence per operation); `record_time_ms_per_step` is the RecordTime "
                       "table of a second run with PP_TIMER_FENCE=1 (device time per operation, one extra host wait each)"
                       % (nptcl, iters, a.deg))
    return out


def run_virtual_ranks(pp, capi, a, V):
    """BASELINE configs[4] at its FULL size on ONE GPU: V virtual ranks (element-block owners) of a.particles each on a
    `local` communicator -- real migration traffic between the blocks (pack, exchange through device memory, unpack
    into the receiver's rebuild), gyroSync over the V ranks -- all on one stream.  The time of a step divided by V
    is the per-GPU COMPUTE of an N = V run (everything but the xGMI transfer itself): the like-for-like N = 1 point
    of the scaling curve."""
    from pumipic_amd import dist as ppdist
    s = pp.synth
    t_set = time.perf_counter()
    ws = []
    for r in range(V):
        w = build_workload(pp, capi, "c5", a.particles, r, V, a.deg, a.remainder, "1m", a.sigma)
        for k in ("elem", "info", "ppe"):  # (host copies of 32 M particles per rank: not needed again)
            w.pop(k, None)
        ws.append(w)
        beat("virtual rank %d built" % r)
    mesh, ne = ws[0]["mesh"], ws[0]["ne"]  # (every rank of a node holds the full mesh; here they share one copy)
    owners = ppdist.element_block_owners(ne, V)
    owners_d = capi.DevArray.from_host(owners)
    safes = [capi.DevArray.from_host((owners == r).astype(np.uint8)) for r in range(V)]
    comms = capi.Comm.local(V)
    fwd, bkwd = capi.create_gyro_ring_mappings(mesh)
    ids, wf, wb, packed = [], [], [], [None] * V
    for w in ws:
        cap = w["ps"].capacity()
        ids.append(capi.DevArray.from_host(np.full(cap + cap // 4, -1, dtype=np.int32)))
        wf.append(capi.DevArray(mesh.nverts, np.float64))
        wb.append(capi.DevArray(mesh.nverts, np.float64))
    t_set = time.perf_counter() - t_set
    moved = [0]

    def step():
        for r, w in enumerate(ws):
            ps = w["ps"]
            capi.push_search(mesh, ps, s.XGC_H, s.XGC_K, s.XGC_D, a.deg, ids[r], seeded=False, looplimit=200,
                             want_found=False)
            capi.migrate_ptcls_begin(ps, ids[r], safes[r], owners_d, comms[r], commit=True,
                                     scatter=(mesh, [fwd, bkwd], [wf[r], wb[r]]))
        for r, w in enumerate(ws):
            ns, _ = capi.migrate_end(w["ps"], comms[r])
            moved[0] += ns
            cap = w["ps"].capacity()
            if cap > ids[r].n:
                ids[r] = capi.DevArray(cap + cap // 4, np.int32)
        for r in range(V):
            packed[r] = capi.gyro_sync_pack(mesh.nverts, wf[r], wb[r], out=packed[r])
        for r in range(V):
            comms[r].allreduce_sum(packed[r])

    import gc
    gc.collect()
    clock_prewarm(capi, float(os.environ.get("PP_BENCH_PREWARM", "0.3")))
    for _ in range(a.warmup):
        step()
    capi.sync()
    moved[0] = 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    capi.sync()
    dt = time.perf_counter() - t0
    n_all = sum(w["ps"].nPtcls() for w in ws)
    per_rank_ms = dt / a.steps / V * 1e3
    out = {"metric": METRIC["c5"], "value": n_all * a.steps / dt, "unit": "particles/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "virtual_ranks": V, "per_rank_ms": per_rank_ms, "per_rank_value": n_all / V / (per_rank_ms * 1e-3),
           "per_rank_roofline_frac": 194.0 * (n_all / V) / (per_rank_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "particles_total": n_all, "migrated_per_step": moved[0] / a.steps, "setup_seconds": t_set,
           "config": {"workload": "configs[4] at full size on one GPU: %s, %d virtual ranks x %d particles = %d, element-block "
                                  "owners, push+search+migrate(local exchange)+rebuild+gyroScatter x2+gyroSync; "
                                  "per_rank_ms = step time / %d: the compute one GPU of an N = %d run does per step "
                                  "(the xGMI transfer is not in it)" % (ws[0]["label"], V, a.particles, n_all, V, V),
                      "parallelism": "%d virtual ranks on one GPU (pp_comm_create_local)" % V}}
    for c in comms:
        c.destroy()
    return out


ALSO = {"driver_pseudoxgcm": also_driver_pseudoxgcm, "c2": also_c2, "c2mt": also_c2mt, "c4_1Me_1Mp": also_c4, "c4_reference_shape": also_c4ref, "c3_general_scatter": also_general_scatter,
        "ps_parallel_for": also_parallel_for}


METRIC = {
    "c2": "particles pushed+searched / sec / GPU; achieved HBM GB/s vs peak",
    "2d": "particles pushed+searched / sec / GPU; achieved HBM GB/s vs peak",
    "c2mt": "particles pushed+searched / sec / GPU; achieved HBM GB/s vs peak",
    "c3": "particles pushed+searched+scattered / sec / GPU; achieved HBM GB/s vs peak",
    "2dc3": "particles pushed+searched+scattered / sec / GPU; achieved HBM GB/s vs peak",
    # BASELINE.json's metric string; the migration is part of the step (config.workload says so) and `value`
    # is the whole-job aggregate over all ranks (`value_scope`)
    "c5": "particles pushed+searched+scattered / sec / GPU; achieved HBM GB/s vs peak",
    "c4": "particles pseudo-pushed+redistributed+rebuilt / sec / GPU; achieved HBM GB/s vs peak",
}


class Watchdog:
    """A rank that makes no progress for `limit` seconds EXITS with code 3 (a fresh failure -- never a
    re-exec, never a retry): a hang inside a collective would otherwise hold the whole job until the
    driver's own limit.  The main thread calls beat(label) at every phase boundary; a daemon thread checks
    the age of the last beat (ctypes calls and torch.distributed release the GIL, so it runs while the main
    thread is blocked)."""

    def __init__(self, limit_s, rank):
        import threading
        self.limit, self.rank = float(limit_s), rank
        self.last, self.label = time.monotonic(), "start"
        self.enabled = self.limit > 0
        if self.enabled:
            threading.Thread(target=self._run, name="bench-watchdog", daemon=True).start()

    def beat(self, label):
        self.last, self.label = time.monotonic(), label

    def stop(self):
        self.enabled = False

    def _run(self):
        while self.enabled:
            time.sleep(min(5.0, max(0.2, self.limit / 10)))
            age = time.monotonic() - self.last
            if self.enabled and age > self.limit:
                sys.stderr.write("bench.py: rank %d made no progress for %.0f s (last phase: %s) -- giving up, "
                                 "exit code 3\n" % (self.rank, age, self.label))
                sys.stderr.flush()
                os._exit(3)


WATCHDOG = None


def beat(label):
    if WATCHDOG is not None:
        WATCHDOG.beat(label)


def launch_ranks(n, argv):
    """`--gpus N` without a launcher: start N rank processes (one per GPU) from a parent that never
    touches the GPU -- a process that has initialised HIP must not fork/exec workers -- and relay
    their exit status.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    have = None
    if os.environ.get("PP_BENCH_ASSUME_GPUS"):
        have = int(os.environ["PP_BENCH_ASSUME_GPUS"])
    else:
        import torch  # device_count() reads the driver's topology; it does not create a HIP context
        have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write("bench.py: --gpus %d needs %d GPUs, this box has %d -- not running a smaller job "
                         "under that label\n" % (n, n, have))
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in pending:  # one rank died: the others would wait in a collective for ever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["c2", "c2mt", "c3", "c4", "c5", "2d", "2dc3"],
                    help="default: c3 on one GPU (the configuration the metric names), c5 on several")
    ap.add_argument("--sigma", type=int, default=2**31 - 1,
                    help="SCS sorting window (elements); the pseudoXGCm value is INT_MAX = full sort")
    ap.add_argument("--safe-layers", type=int, default=0,
                    help="c5: breadth-first element layers around the owned block that are still safe "
                         "(0 = BASELINE's rule: a particle migrates as soon as it leaves its owner's block)")
    ap.add_argument("--c4-elems", type=int, default=1_000_000, help="c4: number of elements")
    ap.add_argument("--c4-dist", type=int, default=1, choices=[1, 2, 3, 4],
                    help="c4: distribution strategy of the population and of the redistribution "
                         "(Distribute.cpp: 1 uniform, 2 gaussian, 3 exponential, 4 GITRm approximation)")
    ap.add_argument("--structure", default="scs", choices=["scs", "csr"], help="c4: particle structure")
    ap.add_argument("--particles", type=int, default=None,
                    help="particles per GPU (default 10 M; c5 on several GPUs 32 M; c4 1 M)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="several GPUs: weak = --particles per GPU (default, what the driver's scaling run "
                         "computes efficiencies from), strong = --total-particles over all GPUs "
                         "(BASELINE configs[4]: 256 M)")
    ap.add_argument("--total-particles", type=int, default=256_000_000,
                    help="--scaling strong: particles of the whole job")
    ap.add_argument("--deg", type=float, default=0.5, help="degrees per push (testing.cmake:117)")
    ap.add_argument("--cpu-sample", type=int, default=None,
                    help="particles of the bounded CPU-baseline sample (~10 s on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mesh", default=None, choices=["100k", "1m"],
                    help="3-D mesh: 100 800 tets (configs[1-2]) or 998 400 tets (configs[4], default of c5 "
                         "on several GPUs)")
    ap.add_argument("--remainder", default="last", choices=["last", "spread"],
                    help="where particles left over by the Gaussian draws go: 'last' = literal "
                         "pseudoXGCm rule (one outlier element), 'spread' = evenly")
    ap.add_argument("--origin-trust", action="store_true",
                    help="skip check_initial_parents from the second step on (pp_ps_set_origin_trust: exact, "
                         "the origins are the destinations the previous walk accepted; buys nothing on the "
                         "record-fed kernels, so off by default since round 4)")
    ap.add_argument("--no-origin-trust", action="store_true", help=argparse.SUPPRESS)  # (round 2-3 default was on)
    ap.add_argument("--no-also", action="store_true",
                    help="N = 1, c3: skip the extra measurements the line reports under `also` (configs[1] c2, "
                         "configs[3] c4, the intersection-mode search c2mt, the general scatter, the "
                         "ps::parallel_for lambda)")
    ap.add_argument("--watchdog", type=float, default=None,
                    help="seconds without progress after which a rank exits with code 3 (default: 120 on "
                         "multi-rank runs, off on one rank; 0 = off)")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="c5 on ONE GPU as this many virtual ranks of --particles each (local communicator): "
                         "configs[4] at full size is --workload c5 --virtual-ranks 8 --particles 32000000")
    ap.add_argument("--no-scale-ref", action="store_true",
                    help="N = 1: skip the extra measurement of the multi-GPU workload's one-rank share "
                         "(c5, 998 400 tets, 32 M particles) that the line reports as `scale_ref`")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch", "tcp"],
                    help="c5 exchange: RCCL behind the C-ABI (pp_ps_migrate / pp_allreduce_sum), the "
                         "torch.distributed glue of pumi-pic_amd/dist.py, or the library's host-staged TCP "
                         "transport (rehearsal of the multi-rank line on a box with fewer GPUs than ranks: "
                         "with PP_BENCH_REHEARSAL=1 every rank uses GPU 0 and the timing barrier runs over gloo)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a.gpus, sys.argv[1:])  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and rank == 0 and ("--gpus" in sys.argv or world > 1):
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d\n"
                         % (a.gpus, world, world))
    if a.workload is None:
        a.workload = "c3" if world == 1 else "c5"
    if a.mesh is None:
        a.mesh = "1m" if (a.workload == "c5" and world > 1) else "100k"
    if a.scaling == "strong":
        a.particles = a.total_particles // world
    if a.particles is None:
        a.particles = (1_000_000 if a.workload == "c4" else
                       32_000_000 if (a.workload == "c5" and world > 1 and a.mesh == "1m") else 10_000_000)
    full_step = a.workload in ("c3", "2dc3", "c5")
    if a.cpu_sample is None:
        a.cpu_sample = 2_000_000 if full_step else 100_000 if a.workload == "c2mt" else 4_000_000
    import torch
    # Rehearsal (tests): N ranks on ONE GPU, gloo for the timing barrier, --comm tcp for the data path.  RCCL
    # cannot put two ranks on one device, so this is how the multi-rank line's plumbing runs on a 1-GPU box;
    # the JSON line says so and is not a measurement.
    rehearsal = world > 1 and os.environ.get("PP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        if a.comm != "tcp":
            sys.stderr.write("bench.py: PP_BENCH_REHEARSAL=1 needs --comm tcp\n")
            sys.exit(2)
        local_rank = 0
    if world > 1 and not rehearsal and torch.cuda.device_count() < world:
        sys.stderr.write("bench.py: %d ranks need %d GPUs, this box has %d\n"
                         % (world, world, torch.cuda.device_count()))
        sys.exit(2)
    dist = None
    ctl_cpu = True
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a first-run RCCL failure must be diagnosable from the run's stderr tail: WARN prints nothing on success
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        torch.cuda.set_device(local_rank)
        # The control plane (timing barrier, max-over-ranks, the broadcast of the RCCL id) is gloo on the
        # host; the data path is the library's own RCCL communicator.  PyTorch brings its own ROCm stack
        # (libamdhip64 / libhsa-runtime64 / librccl under torch/lib, next to the system's under /opt/rocm that
        # libpumipic_hip.so is linked against): a second RCCL from the other stack in the same process buys
        # nothing and doubles the channel / proxy resources.  --comm torch (the round-1 Python glue) still
        # needs torch's nccl backend.
        ctl_cpu = rehearsal or a.comm != "torch"
        if ctl_cpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    pp = pumipic_amd_loader.load()
    from pumipic_amd import capi
    capi.init(local_rank)  # raises when the HIP library / GPU is missing: no CPU fallback
    if a.virtual_ranks > 1:  # configs[4] at full size on one GPU (its own, shorter, line)
        if world != 1 or a.workload != "c5":
            sys.stderr.write("bench.py: --virtual-ranks needs --workload c5 on one rank\n")
            sys.exit(2)
        print(json.dumps(run_virtual_ranks(pp, capi, a, a.virtual_ranks)))
        return
    # PyTorch creates its HIP context lazily at the first torch.cuda call -- 50-60 ms on a fresh box, which used
    # to land inside the first timed region (the barrier's torch.cuda.synchronize()): seen as sporadic
    # 3 ms/step `cold_clocks` values in round 3's runs.  Pay it here, before anything is timed.
    torch.cuda.synchronize()
    global WATCHDOG
    wd_limit = a.watchdog if a.watchdog is not None else (120.0 if world > 1 else 0.0)
    # (building a 32 M-particle population is minutes of numpy on the host: the set-up phase gets its own,
    # longer, allowance -- the limit proper starts with the first collective)
    WATCHDOG = Watchdog(max(wd_limit, 900.0) if wd_limit > 0 else 0.0, rank)
    beat("set-up")

    if a.workload == "c4":
        w = build_c4(pp, capi, a.c4_elems, a.particles, rank, a.structure, a.c4_dist)
        st = StepperC4(capi, w)
    else:
        w = build_workload(pp, capi, a.workload, a.particles, rank, world, a.deg, a.remainder, a.mesh,
                           a.sigma)
        w["safe_layers"] = a.safe_layers
        w["comm"] = a.comm
        w["origin_trust"] = bool(a.origin_trust) and not a.no_origin_trust
        beat("workload built")
        if WATCHDOG.enabled:
            WATCHDOG.limit = wd_limit
        st = Stepper(pp, capi, w, a.workload, a.deg)

    # The host must not stall inside a timed region: Python's cyclic collector, when it happens to run there, also
    # runs the finalisers of every device array the set-up left behind (pp_free -> hipFree: a device
    # synchronisation and an unmap each) -- a one-off 35-55 ms pause among the first steps after a population was
    # built (seen as `cold_clocks` 3.49 ms per step in BENCH_r03, as 17 against 12 ms in c2mt runs; cold step traces
    # in DESIGN.md "Round 4").  Collect now, keep the collector off while steps are timed.
    import gc
    gc.collect()
    gc.disable()

    def barrier():
        beat("barrier")
        capi.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        beat("barrier passed")

    def timed_run(steps, trace=None):
        st.ntimed = 0
        st.kernel_ms = []
        # HIP events bracket about seven of the K timed steps (whole step and pp_push_search call alternately, the
        # median of each is reported): an event record is a barrier packet (~6 us) on a 0.6 ms step -- three records
        # on every second step (round 3) cost the line ~1.5 %
        st.sample_every = max(1, steps // 6)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            st.step(timed=True)
            if trace is not None:  # host time of the call (it returns when the rebuild's totals are on the host)
                trace.append(round((time.perf_counter() - t1) * 1e3, 3))
        if trace is not None:
            t1 = time.perf_counter()
        barrier()
        if trace is not None:
            trace.append(round((time.perf_counter() - t1) * 1e3, 3))
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device="cpu" if ctl_cpu else "cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # Clock pre-warm.  The GPU's power management needs ~0.1 s of sustained vector-ALU load before the
    # shader clock settles; W = 3 warm-up steps are a few ms.  Measured on MI355X: c2 0.307 ms per step
    # from idle clocks, 0.270 ms after 0.3 s of ANY FP64 kernel (a memset loop, which only loads the
    # memory side, gets half of that), the same as after 100+ warm-up steps.  The pre-warm runs an
    # unrelated kernel (closest point on a triangle, scratch data): the workload's state is untouched
    # and still gets exactly W warm-up and K timed steps.  PP_BENCH_PREWARM=<seconds> (0 = off).
    # The same W + K steps from idle clocks are timed first and reported as `cold_clocks`.
    prewarm_s = float(os.environ.get("PP_BENCH_PREWARM", "0.3"))
    cold = None
    if prewarm_s > 0 and not os.environ.get("PP_BENCH_NO_COLD"):
        for _ in range(a.warmup):
            st.step()
        cold_trace = []
        cold_dt = timed_run(a.steps, trace=cold_trace)
        cold = {"ms_per_step": cold_dt / a.steps * 1e3,
                "host_ms_of_each_step_then_closing_barrier": cold_trace,
                "note": "the same W warm-up + K timed steps before the clock pre-warm, on the population as built: "
                        "the shader clock is still ramping AND the step gets faster as the population ages (per-element "
                        "counts even out, the layout needs less padding: 0.639 ms in the first ten steps of a fresh c3 "
                        "structure, 0.605 after seventy, tools/history/r04_age_exp.py) -- ms_per_step is measured on steps "
                        "2W+K+1 .. 2W+2K of the structure's life"}
    clock_prewarm(capi, prewarm_s)
    for _ in range(a.warmup):
        st.step()
    dt = timed_run(a.steps)
    kms_main, sms_main = st.kernel_avg_ms(), (st.step_avg_ms() if hasattr(st, "step_avg_ms") else None)
    nlive = w["ps"].nPtcls()
    total_particles = nlive
    gc.enable()
    # ---- everything below is EXTRA: measured after the headline, each piece guarded -- an exception, an
    # out-of-memory or a time budget that runs out in an extra must never cost the line its headline
    # (round-3 advisor finding); a failed piece reports {"error": ...} in its place.
    extras_on = world == 1 and not os.environ.get("PP_BENCH_NO_EXTRAS")
    t_extras = time.perf_counter()
    EXTRA_BUDGET_S = float(os.environ.get("PP_BENCH_EXTRA_BUDGET", "200"))

    def guarded(fn, *args):
        if time.perf_counter() - t_extras > EXTRA_BUDGET_S:
            return {"error": "skipped: the extras' time budget (%g s) was used up" % EXTRA_BUDGET_S}
        try:
            return fn(*args)
        except Exception as e:  # noqa: BLE001
            return {"error": "%s: %s" % (type(e).__name__, e)}

    notrust_ms = None
    # ---- scale_ref: ONE rank's share of the multi-GPU workload (c5: 998 400 tets, 32 M particles, migrating
    # step on a one-rank communicator), measured in this very run, so that the N = 1 point of a 1 -> 8 sweep
    # and the N > 1 points (which run c5) can be compared like with like
    scale_ref = None
    also = None
    default_c3 = (a.workload == "c3" and a.mesh == "100k" and a.particles == 10_000_000 and a.sigma >= 2**31 - 1)

    def measure_scale_ref():
        t_set = time.perf_counter()
        w5 = build_workload(pp, capi, "c5", 32_000_000, 0, 1, a.deg, a.remainder, "1m", a.sigma)
        w5["safe_layers"], w5["comm"], w5["origin_trust"] = 0, "rccl", w.get("origin_trust", False)
        st5 = Stepper(pp, capi, w5, "c5", a.deg)
        t_set = time.perf_counter() - t_set
        clock_prewarm(capi, prewarm_s)  # (the clocks fell back while the host built the population)
        for _ in range(12):
            st5.step()
        k5 = 8
        capi.sync()
        t0 = time.perf_counter()
        for _ in range(k5):
            st5.step()
        capi.sync()
        dt5 = time.perf_counter() - t0
        n5 = w5["ps"].nPtcls()
        out5 = {"workload": "c5 on one rank: %s, %d particles spread over the WHOLE mesh (32 per element), "
                            "push+search+migrate(no peer)+rebuild+gyroScatter x2" % (w5["label"], 32_000_000),
                "ms_per_step": dt5 / k5 * 1e3, "value": n5 * k5 / dt5, "unit": "particles/s", "steps": k5,
                "warmup": 12, "setup_seconds": t_set,
                "roofline_frac": 194.0 * n5 / (dt5 / k5) / 1e9 / HBM_PEAK_GBS,
                "note": "NOT the step a rank of an N = 8 run takes: there a rank's 32 M particles sit in its own "
                        "element block (124 800 tets, 256 per element) and seven eighths of its rows are empty -- "
                        "`rank_of_8_population` below is that population on one rank (no peer, so no pack / unpack of "
                        "migration traffic), `virtual_ranks_8` the committed run of all eight ranks on one GPU "
                        "(bench.py --workload c5 --virtual-ranks 8 --particles 32000000)"}
        del st5, w5
        # rank 0 of a world of 8: the same 32 M particles in the FIRST element block only
        t_set = time.perf_counter()
        w8 = build_workload(pp, capi, "c5", 32_000_000, 0, 8, a.deg, a.remainder, "1m", a.sigma)
        w8["rank"], w8["world"] = 0, 1  # (stepped alone: leavers stay on this rank, nothing arrives)
        w8["safe_layers"], w8["comm"], w8["origin_trust"] = 0, "rccl", w.get("origin_trust", False)
        st8 = Stepper(pp, capi, w8, "c5", a.deg)
        t_set = time.perf_counter() - t_set
        clock_prewarm(capi, prewarm_s)
        for _ in range(12):
            st8.step()
        capi.sync()
        t0 = time.perf_counter()
        for _ in range(k5):
            st8.step()
        capi.sync()
        dt8 = time.perf_counter() - t0
        n8 = w8["ps"].nPtcls()
        out5["rank_of_8_population"] = {
            "workload": "the population of rank 0 of 8 (32 M particles in the first 124 800 tets, 256 per element) "
                        "stepped on one rank: the per-GPU compute of configs[4] without the exchange",
            "ms_per_step": dt8 / k5 * 1e3, "value": n8 * k5 / dt8, "unit": "particles/s", "steps": k5, "warmup": 12,
            "setup_seconds": t_set, "roofline_frac": 194.0 * n8 / (dt8 / k5) / 1e9 / HBM_PEAK_GBS}
        vf = os.path.join(ROOT, "profiles", "r06_c5_virtual8.json")
        if not os.path.exists(vf):
            vf = os.path.join(ROOT, "profiles", "r05_c5_virtual8.json")
        if os.path.exists(vf):
            try:
                vj = json.load(open(vf))
                out5["virtual_ranks_8"] = {k: vj.get(k) for k in ("per_rank_ms", "per_rank_value", "per_rank_roofline_frac",
                                                                 "particles_total", "migrated_per_step", "ms_per_step")}
                out5["virtual_ranks_8"]["provenance"] = "profiles/%s (committed run of this file)" % os.path.basename(vf)
                out5["per_rank_ms"] = {"1": dt5 / k5 * 1e3, "8": vj.get("per_rank_ms")}
            except (ValueError, KeyError):
                pass
        return out5

    if extras_on and default_c3 and not a.no_scale_ref:
        scale_ref = guarded(measure_scale_ref)
    if extras_on and default_c3 and not a.no_also:
        also = {k: guarded(f, pp, capi, a, w, st) for k, f in ALSO.items()}
    if dist is not None:
        t = torch.tensor([nlive], device="cpu" if ctl_cpu else "cuda", dtype=torch.int64)
        dist.all_reduce(t)
        total_particles = int(t.item())

    if rank == 0:
        kms = kms_main
        # HBM bytes per launch from the PMC counters: collected with rocprofv3 in separate --pmc
        # passes of THIS command (tools/history/r03_measure.sh) and calibrated as DESIGN.md section 4 says;
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
                    # (c4's roofline object is about ONE kernel, the pseudo-push: its own bytes per launch)
                    kp = (tj.get("kernels") or {}).get("k_pseudo_push160")
                    if a.workload == "c4" and kp and "read_bytes_per_step" in kp:
                        traffic = (kp["read_bytes_per_step"] + kp["write_bytes_per_step"]) / max(kp.get("launches_per_step", 1), 1)
            except (ValueError, KeyError):
                traffic = None
        if a.workload == "c4" and a.structure != "scs":
            traffic = None  # the committed PMC run is the SCS structure
        common = {
            "metric": METRIC[a.workload],
            "value": total_particles * a.steps / dt, "unit": "particles/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "clock_prewarm_s": prewarm_s,
            "value_scope": "whole job: particles of all %d rank(s) x steps / max-over-ranks time" % world,
        }
        if cold is not None:
            common["cold_clocks"] = cold
        if a.workload == "c4":
            bpp = BYTES["c4"]
            out = dict(common)
            out["config"] = {"workload": "%s, %d particles/GPU, pseudo-push + redistribute(0.5, same strategy) + "
                                         "rebuild per step" % (w["label"], a.particles),
                             "parallelism": "%d independent rank(s)" % world}
            out["roofline"] = {"bound": "hbm", "achieved": bpp * nlive / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": bpp * nlive / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "traffic": traffic,
                               "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, calibrated; "
                                               "profiles/traffic_c4.json)",
                               "kernel": "k_pseudo_push160", "kernel_ms": kms,
                               "bytes_per_particle": bpp}
            if not a.no_cpu_baseline and world == 1:
                out["cpu_baseline"] = cpu_baseline_c4(pp, a.c4_elems, a.particles, min(a.particles, 1_000_000))
            print(json.dumps(out))
            if dist is not None:
                dist.destroy_process_group()
            return
        # Roofline.  c2 / 2d: ONE pp_push_search call is the step.  c3 / 2dc3 / c5: the WHOLE step
        # (push+search 69 B + rebuild 125 B per particle) against the median HIP-event time
        # of the sampled steps; the two phases are broken out under "phases".
        bpp_ps = BYTES["2d" if w["dim"] == 2 else "c2mt" if a.workload == "c2mt" else "c2"]
        sms = sms_main
        ps_kernel = ("k_search_mt3 (pp_search_mesh, intersection mode; + k_toroidal_push)" if a.workload == "c2mt" else
                     "k_push_walk_rowsq<3> + k_walk_pending<3> (one pp_push_search call)"
                     if w["dim"] == 3 else "k_push_walk_rows<2>")
        if full_step:
            bpp = bpp_ps + 125.0
            achieved = bpp * nlive / (sms * 1e-3) / 1e9 if sms else None
            rest_ms = (sms - kms) if (sms and kms) else None
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                    "traffic_unit": "bytes per step (rocprofv3 FETCH_SIZE+WRITE_SIZE of every kernel of the "
                                    "step, calibrated; profiles/traffic_%s.json)" % a.workload,
                    "scope": "whole step: every kernel between two steps' first launches (HIP events on the "
                             "library stream)",
                    "kernel": "whole step; its longest launches (profiles/r06_%s_kernel_stats.csv): the record-fed %s, "
                              "then k_move_pack_rm<2> (the re-layout's one data pass: 32-B records + the third member "
                              "beside them, row-major inside a chunk, stored as runs)" % (
                                  "c5_virtual8" if a.workload == "c5" else a.workload,
                                  "k_push_walk_rowsq<3> (+ k_walk_pending<3>)" if w["dim"] == 3 else "k_push_walk_rows<2>"),
                    "kernel_ms": sms, "bytes_per_particle": bpp,
                    "phases": {
                        "push_search": {"ms": kms, "bytes_per_particle": bpp_ps,
                                        "frac": bpp_ps * nlive / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms else None},
                        "rebuild_scatter": {"ms": rest_ms, "bytes_per_particle": 125.0,
                                            "frac": 125.0 * nlive / (rest_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                            if rest_ms else None,
                                            "note": "updatePtclPositions + rebuild (+ migration) + gyroScatter x2"}}}
        else:
            bpp = bpp_ps
            achieved = bpp * nlive / (kms * 1e-3) / 1e9 if kms else None
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                    "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, calibrated; "
                                    "profiles/traffic_%s.json)" % a.workload,
                    "kernel": ps_kernel, "kernel_ms": kms, "bytes_per_particle": bpp}
        # also against the copy ceiling MI355X_MICROARCH.md measures (float4 copy, 6.29 TB/s = 79 % of
        # the spec; tools/ub_copy.hip reaches 6.1-6.4 TB/s here)
        roof["frac_of_measured_copy"] = (achieved / COPY_CEILING_GBS) if achieved else None
        roof["measured_copy_peak"] = COPY_CEILING_GBS
        out = dict(common)
        out["config"] = {"workload": "%s, %d particles/GPU, SCS C=64 sigma=%s V=1024, %s" % (
            w["label"], a.particles, "inf" if a.sigma >= 2**31 - 1 else str(a.sigma),
            {"c2": "push+search only (fused toroidal push + BCC walk), deg/push=%g" % a.deg,
             "c2mt": "toroidal push + search_mesh in intersection mode (Moeller-Trumbore, every ray followed to the "
                     "domain boundary), deg/push=%g" % a.deg,
             "2d": "elliptical push + search_mesh_2d (fused), deg/push=%g" % a.deg,
             "c3": "push+search+rebuild+gyroScatter x2 (tet ring map) every step, deg/push=%g; the second scatter "
                   "field is a device copy of the first (the two ring maps of one createGyroRingMappings call hold "
                   "the same ids); check_initial_parents %s" % (
                       a.deg, "skipped from step 2 on (--origin-trust)" if w.get("origin_trust") else
                       "runs every step") + "; the search's found flag is read every step (it reaches the host with "
                   "the rebuild's totals)",
             "c5": "push+search+migrate(all-to-all-v, %s)+rebuild+gyroScatter x2+gyroSync, deg/push=%g" % (
                 getattr(st, "comm_kind", a.comm), a.deg),
             "2dc3": "push+search+rebuild+gyroScatter x2 every step, deg/push=%g" % a.deg}[a.workload]),
            "parallelism": "element-block partition, %d rank(s), full-mesh replica" % world}
        if scale_ref is not None:
            out["scale_ref"] = scale_ref
        if also is not None:
            out["also"] = also
        if world > 1:
            if getattr(st, "preflight", None):
                out["preflight"] = st.preflight
            ref_file = os.path.join(ROOT, "profiles", "scale_ref.json")
            if os.path.exists(ref_file) and a.workload == "c5" and a.mesh == "1m" and a.scaling == "weak":
                try:
                    ref = json.load(open(ref_file))
                    out["scale_ref"] = {"value": ref["value"], "ms_per_step": ref["ms_per_step"],
                                        "source": "profiles/scale_ref.json (this workload on one rank, from a "
                                                  "driver-style N = 1 run of this file: key scale_ref)"}
                    out["efficiency_vs_scale_ref"] = out["value"] / (world * ref["value"])
                    out["scale_ref"]["caveat"] = ("32 M particles spread over the WHOLE mesh on one rank: a rank of this "
                                                  "run holds its 32 M in its own block of ne / %d elements" % world)
                except (ValueError, KeyError):
                    pass
            v8 = os.path.join(ROOT, "profiles", "r06_c5_virtual8.json")
            if not os.path.exists(v8):
                v8 = os.path.join(ROOT, "profiles", "r05_c5_virtual8.json")
            if os.path.exists(v8) and a.workload == "c5" and a.mesh == "1m" and a.scaling == "weak" and world == 8:
                try:  # like for like: the same eight ranks as virtual ranks of ONE GPU (compute without the transfer)
                    vj = json.load(open(v8))
                    out["scale_ref_virtual_ranks_8"] = {"per_rank_value": vj["per_rank_value"], "per_rank_ms": vj["per_rank_ms"],
                                                        "source": "profiles/" + os.path.basename(v8)}
                    out["efficiency_vs_virtual_ranks_8"] = out["value"] / (world * vj["per_rank_value"])
                except (ValueError, KeyError):
                    pass
            out["watchdog_s"] = WATCHDOG.limit if WATCHDOG is not None and WATCHDOG.enabled else 0
        if rehearsal:
            out["rehearsal"] = ("NOT a measurement: %d ranks share GPU 0, the exchange is the host-staged TCP transport "
                                "and the timing barrier runs over gloo (PP_BENCH_REHEARSAL=1)" % world)
        if a.workload == "c5":
            out["rank0_sent_per_step"] = st.moved / max(1, st.steps_done)
        if a.workload == "c2mt":
            nsteps = capi.search_walk_steps()
            out["walk"] = {"elements_visited_per_particle": nsteps / max(nlive, 1),
                           "ns_per_visited_element": (kms * 1e6 / nsteps) if (kms and nsteps) else None,
                           "visited_elements_per_s": nsteps / (kms * 1e-3) if kms else None,
                           "note": "intersection mode follows the RAY to the boundary (adjacency.tpp:488, "
                                   "'trajectories are considered as rays'): cost scales with elements visited; "
                                   "one visit = one 128-B record + up to 4 ray/triangle tests"}
        elif w["dim"] == 3:
            nf, nie, unm = capi.push_search_counters()
            out["origin_trust"] = {"on": bool(w.get("origin_trust", False)), "unmoved_without_test_last_step": unm,
                                   "not_in_elem_last_step": nie,
                                   "note": "on: check_initial_parents skipped from step 2 on (the origins are the "
                                           "destinations the previous walk accepted; 0 unmoved finishes = the "
                                           "skipped test would have passed for every particle); off (default): "
                                           "the test runs for every particle every step"}
        if full_step:
            ip, fl, rm = w["ps"].rebuild_stats()
            out["rebuilds"] = {"kept_layout": ip, "full_relayout": fl, "rows_traded": rm,
                               "note": "how the structure's rebuilds ended (warm-up and cold-clock steps included)"}
        out["roofline"] = roof
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pp, w, a.workload, a.deg, min(a.cpu_sample, a.particles),
                                               steps=10 if full_step else 20)
        print(json.dumps(tail_safe(out)))
    if dist is not None:
        dist.destroy_process_group()


def tail_safe(out):
    """The driver keeps the LAST 8 KB of the line: the bulky extras go first, the contract's keys after them, and a
    <= 1 KB `digest` of what the extras measured closes the line (round-5 verdict, weak #12)."""
    bulky = [k for k in ("cold_clocks", "also", "scale_ref") if k in out]
    ordered = {k: out[k] for k in bulky}
    ordered.update({k: v for k, v in out.items() if k not in bulky})

    def pick(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return round(d, 4) if isinstance(d, float) else d

    also, sr = out.get("also") or {}, out.get("scale_ref") or {}
    digest = {
        "driver_pseudoxgcm": {"ms": pick(also, "driver_pseudoxgcm", "ms_per_step"),
                              "x_fused": pick(also, "driver_pseudoxgcm", "driver_over_fused"),
                              "reference_source_unchanged_ms": pick(also, "driver_pseudoxgcm", "reference_source_ms_per_step")},
        "c2": {"ms": pick(also, "c2", "ms_per_step"), "frac": pick(also, "c2", "roofline_frac")},
        "c2mt": {"ms": pick(also, "c2mt", "ms_per_step"), "rays_per_s": pick(also, "c2mt", "value")},
        "c4_1Me_1Mp": {"step_ms": pick(also, "c4_1Me_1Mp", "ms_per_step"),
                       "rebuild_ms": pick(also, "c4_reference_shape", "1Me_1Mp", "rebuild_ms"),
                       "rebuild_frac": pick(also, "c4_reference_shape", "1Me_1Mp", "rebuild_roofline_frac")},
        "c4_50ke_50Mp": {"rebuild_ms": pick(also, "c4_reference_shape", "50ke_50Mp", "rebuild_ms"),
                         "rebuild_frac": pick(also, "c4_reference_shape", "50ke_50Mp", "rebuild_roofline_frac")},
        "scatter_radius": {"ms": pick(also, "c3_general_scatter", "ms_per_call"),
                           "frac": pick(also, "c3_general_scatter", "roofline_frac")},
        "scale_ref": {"ms": pick(sr, "ms_per_step"), "frac": pick(sr, "roofline_frac"),
                      "rank_of_8_ms": pick(sr, "rank_of_8_population", "ms_per_step"),
                      "rank_of_8_frac": pick(sr, "rank_of_8_population", "roofline_frac"),
                      "virtual8_per_rank_ms": pick(sr, "virtual_ranks_8", "per_rank_ms")},
        "phases_frac": {"push_search": pick(out, "roofline", "phases", "push_search", "frac"),
                        "rebuild_scatter": pick(out, "roofline", "phases", "rebuild_scatter", "frac")},
    }
    if any(v is not None for d in digest.values() for v in d.values()):
        ordered["digest"] = {k: {kk: vv for kk, vv in d.items() if vv is not None} for k, d in digest.items()
                             if any(v is not None for v in d.values())}
    return ordered


if __name__ == "__main__":
    main()
