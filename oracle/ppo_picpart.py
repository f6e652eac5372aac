"""CPU restatement (numpy) of PUMI-PIC's PICpart construction and comm-array reduction -- TEST
INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the product never
imports it.

Follows, function by function,
  src/pumipic_part_construct.cpp:75-275   Mesh::Mesh(Input&), constructPICPart
  src/pumipic_part_construct.cpp:278-386  setOwnerByClassification, defineOwners, calculateOwnerOffset,
                                          createGlobalNumbering, rankLidNumbering
  src/pumipic_part_construct.cpp:470-507  setSafeEnts, sumPositives
  src/pumipic_comm.cpp:11-191             Mesh::setupComm
  src/pumipic_comm.cpp:249-440            Mesh::reduceCommArray (fan-in through the owners, fan-out)
All ranks of the job are simulated inside one process (`PicParts` holds every rank's part).

PARITY UNPINNED for this file: the reference's tests for these functions (test_comm_array.cpp,
test_input_construct.cpp, test_full_mesh.cpp) need Omega_h meshes from the empty pumipic-data submodule
and hold no golden numbers; what they assert -- minOwnership, sumEntities, fullBufferTest, the BFS safe
zone -- is restated in tests/test_picpart_oracle.py on synthetic meshes.

Two places where the reference leaves the result open, and what is fixed here (and in the HIP library):
  * renumberBoundaryLids numbers the entities of a partially buffered part with atomics, "the order
    doesn't need to be consistent" (pumipic_comm.cpp:66-76): here in increasing picpart entity id;
  * the owner adds the fan-in contributions in arrival order (MPI_Waitany, :311-318): here in increasing
    rank, the owner's own value first.
Entity dimensions: every one from 0 (vertices) to dim (elements) -- sides (dim-1) and, for tets, edges (1) --
as the reference loops them.  A part's vertices and elements are the kept
ones in full-mesh order, as in the reference (:181-194).  Its SIDES and EDGES are numbered by the part's own mesh
(the mesh derives them from its elements -- `Mesh(part arrays)` here, pp_mesh_create in the library), not
in full-mesh order as Omega_h's set_ents would leave them; `full_ids[dim-1]` maps them.  Everything that
travels between ranks (global ids, the order inside a partially held part) is defined on full-mesh ids,
so the difference stays local to the part.
"""
import numpy as np

FULL, BFS, MINIMUM, NONE = 0, 1, 2, 3          # Input::Method, src/pumipic_input.hpp:33-39
SUM_OP, MAX_OP, MIN_OP, BCAST_OP = 0, 1, 2, 3  # Mesh::Op, src/pumipic_mesh.hpp:64-69


def set_owner_by_classification(class_id, class_owners):
    """setOwnerByClassification, pumipic_part_construct.cpp:278-302"""
    return np.asarray(class_owners, dtype=np.int32)[np.asarray(class_id)]


def define_owners(up_off, up, elem_owner, comm_size):
    """defineOwners :305-323 -- a lower-dimension entity belongs to the smallest owner among the elements
    around it"""
    n = len(up_off) - 1
    out = np.full(n, comm_size, dtype=np.int32)
    own = np.asarray(elem_owner)[up]
    ent = np.repeat(np.arange(n), np.diff(up_off))
    np.minimum.at(out, ent, own)
    return out


def calculate_owner_offset(owner, comm_size):
    """calculateOwnerOffset :325-334 -> exclusive scan of the entities per rank (comm_size + 1)"""
    cnt = np.bincount(owner, minlength=comm_size)[:comm_size]
    return np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)


def create_global_numbering(owner, comm_size):
    """createGlobalNumbering + GlobalNumberer :336-376: gid = offset of the owner + number of earlier
    entities with the same owner"""
    off = calculate_owner_offset(owner, comm_size)
    gids = np.zeros(len(owner), dtype=np.int64)
    for r in range(comm_size):
        sel = np.flatnonzero(owner == r)
        gids[sel] = off[r] + np.arange(len(sel))
    return off, gids


def rank_lid_numbering(owner, offset, gids):
    """rankLidNumbering :378-386"""
    return (gids - offset[owner]).astype(np.int32)


def bfs_sweep(up_off, up, visited):
    """BFS :388-405: every bridge entity with a visited element marks all its elements"""
    ent = np.repeat(np.arange(len(up_off) - 1), np.diff(up_off))
    hit = np.zeros(len(up_off) - 1, dtype=bool)
    np.logical_or.at(hit, ent, visited[up].astype(bool))
    nxt = visited.copy()
    nxt[up[hit[ent]]] = 1
    return nxt


def bfs_buffer_layers(up_off, up, owner, rank, comm_size, safe_layers, ghost_layers):
    """bfsBufferLayers :407-437 -> (is_safe, has_part)"""
    visited = (owner == rank).astype(np.int32)
    is_safe = visited.copy()
    has_part = np.zeros(comm_size, dtype=np.int32)
    has_part[rank] = 1
    i = 0
    while i < ghost_layers or i < safe_layers:
        visited = bfs_sweep(up_off, up, visited)
        if i == safe_layers - 1:
            is_safe = visited.copy()
        if i < ghost_layers:
            has_part[np.unique(owner[visited.astype(bool)])] = 1
        i += 1
    return is_safe, has_part


def bfs_safe_inward(up_off, up, owner, rank, safe_layers, has_part):
    """bfsSafeInward :439-468"""
    visited = (has_part[owner] == 0).astype(np.int32)
    for _ in range(safe_layers):
        visited = bfs_sweep(up_off, up, visited)
    return ((visited == 0) | (owner == rank)).astype(np.int32)


def input_safe_and_buffer(up_off, up, owner, rank, comm_size, buffer_method, safe_method, buffer_layers,
                          safe_layers):
    """Mesh::Mesh(Input&) :75-118 -> (is_safe, has_part, is_full_mesh)"""
    ne = len(owner)
    is_safe = np.full(ne, 1 if safe_method == FULL else 0, dtype=np.int32)
    has_part = np.ones(comm_size, dtype=np.int32)
    if (safe_method not in (NONE, FULL)) or buffer_method != FULL:
        safe, part = bfs_buffer_layers(up_off, up, owner, rank, comm_size, safe_layers, buffer_layers)
        if safe_method in (BFS, MINIMUM):
            is_safe = safe
        if buffer_method in (BFS, MINIMUM):
            has_part = part
    if buffer_method == BFS and safe_method == FULL:
        is_safe = bfs_safe_inward(up_off, up, owner, rank, safe_layers, has_part)
    return is_safe, has_part, buffer_method == FULL


TET_EDGE_VERTS = np.array([[0, 1], [1, 2], [2, 0], [0, 3], [1, 3], [2, 3]])  # Omega_h simplex_down_template(3, 1)


def tet_edges(elem2verts):
    """edges of a tet mesh -- what the reference asks Omega_h for (ask_down(3, 1), ask_up(1, 3)): numbered in
    first-seen order over (element, local edge), the vertex pair kept in the orientation first seen.
    Returns (edge2verts [n,2], elem2edges [ne,6], up_off [n+1], up)."""
    e2v = np.asarray(elem2verts, dtype=np.int64).reshape(-1, 4)
    ne = len(e2v)
    pairs = e2v[:, TET_EDGE_VERTS].reshape(-1, 2)
    key = np.minimum(pairs[:, 0], pairs[:, 1]) * (int(e2v.max()) + 1 if ne else 1) + np.maximum(pairs[:, 0], pairs[:, 1])
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # unique keys in the order they are first met
    new_id = np.empty(len(uniq), dtype=np.int64)
    new_id[order] = np.arange(len(uniq))
    elem2edges = new_id[inv].reshape(ne, 6).astype(np.int32)
    edge2verts = pairs[first[order]].astype(np.int32)
    flat = elem2edges.ravel()
    cnt = np.bincount(flat, minlength=len(uniq))
    up_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    up = (np.argsort(flat, kind="stable") // 6).astype(np.int32)  # elements around every edge, ascending
    return edge2verts, elem2edges, up_off, up


class Part:
    """one rank's PICpart (constructPICPart :120-275 + setupComm)"""


def make_mesh(dim, coords, elem2verts, class_id):
    """the oracle's mesh type (oracle/ppo.py), loaded lazily: derives the sides of a part's mesh"""
    import os
    import sys
    if "ppo" not in sys.modules:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import pumipic_amd_loader
        pumipic_amd_loader.load_oracle()
    return sys.modules["ppo"].Mesh(dim, coords, elem2verts, class_id)


class PicParts:
    def __init__(self, mesh, elem_owner, comm_size, buffer_method=FULL, safe_method=FULL, bridge_dim=0,
                 buffer_layers=3, safe_layers=1):
        """mesh: oracle/ppo.py Mesh (full mesh, loaded on every rank as in the reference's drivers);
        MINIMUM is BFS with 0 layers (pumipic_input.cpp sets the layer counts that way)"""
        self.mesh, self.comm_size = mesh, comm_size
        dim = mesh.dim
        owner = np.asarray(elem_owner, dtype=np.int32)
        if buffer_method == NONE:  # pumipic_input.cpp:96-100: "bufferMethod given as NONE, setting to MINIMUM"
            buffer_method = MINIMUM
        if buffer_method == MINIMUM:
            buffer_layers = 0
        if safe_method == MINIMUM:
            safe_layers = 0
        if bridge_dim == 0:
            up_off, up = mesh.vert2elems_off, mesh.vert2elems
        elif bridge_dim == dim - 1:
            up_off, up = mesh.side2elems_off, mesh.side2elems
        else:
            raise ValueError("bridge_dim must be 0 or dim-1")
        # ---- ownership and global numbering of the full mesh (constructPICPart :141-163)
        # entities between the vertices and the elements: the sides, and for tets the edges as well (the
        # reference numbers every dimension 0..dim, pumipic_part_construct.cpp:141-163)
        mid = {dim - 1: (np.asarray(mesh.side2verts).reshape(-1, dim), mesh.side2elems_off, mesh.side2elems)}
        if dim == 3:
            ev, self.elem2edges, eoff, eup = tet_edges(mesh.elem2verts)
            mid[1] = (ev, eoff, eup)
        self.mid = mid
        self.owner = {0: define_owners(mesh.vert2elems_off, mesh.vert2elems, owner, comm_size), dim: owner}
        for d, (_, ent_off, ent_up) in mid.items():
            self.owner[d] = define_owners(ent_off, ent_up, owner, comm_size)
        self.dims = tuple(sorted(self.owner))
        self.offsets, self.gids, self.rank_lids = {}, {}, {}
        for d in self.dims:
            self.offsets[d], self.gids[d] = create_global_numbering(self.owner[d], comm_size)
            self.rank_lids[d] = rank_lid_numbering(self.owner[d], self.offsets[d], self.gids[d])
        self.parts = []
        for rank in range(comm_size):
            p = Part()
            p.rank = rank
            p.is_safe_full, p.has_part, p.is_full_mesh = input_safe_and_buffer(
                up_off, up, owner, rank, comm_size, buffer_method, safe_method, buffer_layers, safe_layers)
            # setSafeEnts :470-494: an entity stays when an element around it belongs to a buffered part
            keep_e = p.has_part[owner].astype(bool)
            keep_v = np.zeros(mesh.nverts, dtype=bool)
            keep_v[mesh.elem2verts[keep_e].ravel()] = True
            p.ent_ids = {}  # full id -> picpart id, -1 = not in the part (:181-194)
            p.full_ids = {}
            for d, keep in ((0, keep_v), (dim, keep_e)):
                ids = np.full(len(keep), -1, dtype=np.int32)
                ids[keep] = np.arange(int(keep.sum()), dtype=np.int32)
                p.ent_ids[d] = ids
                p.full_ids[d] = np.flatnonzero(keep).astype(np.int32)
            # the picpart mesh (:196-258): kept vertices / elements in full-mesh order
            p.coords = mesh.coords[p.full_ids[0]]
            p.elem2verts = p.ent_ids[0][mesh.elem2verts[p.full_ids[dim]]]
            p.class_id = mesh.class_id[p.full_ids[dim]]
            p.safe = p.is_safe_full[p.full_ids[dim]].astype(np.int32)
            # sides / edges: the part's own mesh numbers them; match part entity <-> full entity by their vertices
            if not p.is_full_mesh:
                p.part_mesh = make_mesh(dim, p.coords, p.elem2verts, p.class_id)
            for d, (verts_full, _, _) in mid.items():
                nfull = len(verts_full)
                if p.is_full_mesh:
                    p.full_ids[d] = np.arange(nfull, dtype=np.int32)
                else:
                    part_verts = (np.asarray(p.part_mesh.side2verts).reshape(-1, dim) if d == dim - 1
                                  else tet_edges(p.elem2verts)[0])
                    key = {tuple(sorted(v)): i for i, v in enumerate(np.asarray(verts_full).tolist())}
                    p.full_ids[d] = np.array([key[tuple(sorted(p.full_ids[0][v].tolist()))] for v in part_verts],
                                             dtype=np.int32)
                ids = np.full(nfull, -1, dtype=np.int32)
                ids[p.full_ids[d]] = np.arange(len(p.full_ids[d]), dtype=np.int32)
                p.ent_ids[d] = ids
            p.owners = {d: self.owner[d][p.full_ids[d]] for d in self.dims}
            p.gids = {d: self.gids[d][p.full_ids[d]] for d in self.dims}
            p.rank_lids = {d: self.rank_lids[d][p.full_ids[d]] for d in self.dims}
            p.nents = {d: len(p.full_ids[d]) for d in self.dims}
            self.parts.append(p)
        for p in self.parts:
            for d in self.dims:
                self._setup_comm(p, d)
        for p in self.parts:  # what the owners receive (the MPI_Ialltoall + Isend/Irecv of :113-190)
            p.bounded_ent_ids, p.bounded_offset, p.boundary_parts, p.complete_from = {}, {}, {}, {}
            for d in self.dims:
                lists, off, bparts, comp = [], [0], [], []
                for q in self.parts:
                    if q.rank == p.rank:
                        off.append(off[-1])
                        continue
                    kind = q.is_complete[d][p.rank]
                    if kind == 1:
                        lists.append(q.boundary_rlids[d][p.rank])
                        bparts.append(q.rank)
                    if kind == 2:
                        comp.append(q.rank)
                    off.append(off[-1] + (len(q.boundary_rlids[d][p.rank]) if kind == 1 else 0))
                p.bounded_ent_ids[d] = (np.concatenate(lists) if lists else np.zeros(0, np.int32)).astype(np.int32)
                p.bounded_offset[d] = np.asarray(off, dtype=np.int32)
                p.boundary_parts[d] = bparts  # parts that hold a boundary of this part
                p.complete_from[d] = comp     # parts that buffer this part completely

    def _setup_comm(self, p, d):
        """Mesh::setupComm, pumipic_comm.cpp:11-111"""
        cs = self.comm_size
        own = p.owners[d]
        goff = self.offsets[d]
        poff = calculate_owner_offset(own, cs)  # picpart_ents_per_rank
        gdiff, pdiff = np.diff(goff), np.diff(poff)
        is_complete = (gdiff == pdiff).astype(np.int32) + (pdiff != 0).astype(np.int32)  # :54-63
        lids = (p.gids[d] - goff[own]).astype(np.int32)                                   # :43-49
        boundary_rlids = {}
        for r in range(cs):
            if is_complete[r] == 1:  # renumberBoundaryLids :66-76, in increasing full-mesh id
                sel = np.flatnonzero(own == r)
                sel = sel[np.argsort(p.full_ids[d][sel], kind="stable")]
                lids[sel] = np.arange(len(sel), dtype=np.int32)
                boundary_rlids[r] = p.rank_lids[d][sel].astype(np.int32)  # gatherBoundedEnts :127-137
        if not hasattr(p, "comm_index"):
            p.comm_index, p.nents_offsets, p.is_complete, p.boundary_rlids, p.buffered_parts = {}, {}, {}, {}, {}
        p.comm_index[d] = (lids + poff[own]).astype(np.int32)  # :80-86
        p.nents_offsets[d] = poff
        p.is_complete[d] = is_complete
        p.boundary_rlids[d] = boundary_rlids
        p.buffered_parts[d] = [r for r in range(cs) if pdiff[r] != 0 and r != p.rank]  # :33-40

    def reduce(self, d, op, arrays):
        """reduceCommArray :249-440 on every rank at once.  arrays[r]: (nents_r * nvals) values of rank r's
        part, entity-major; returns the reduced arrays."""
        cs = self.comm_size
        arrays = [np.array(a) for a in arrays]
        if cs == 1:
            return arrays
        nvals = [len(a) // max(p.nents[d], 1) for a, p in zip(arrays, self.parts)]
        # (full-mesh parts: the reference calls MPI_Allreduce here, :262-277, whose summation order is the MPI
        # library's; the same values come out of the fan-in / fan-out below, in THIS file's order -- own value
        # first, then increasing rank -- which is what the HIP library does for every kind of part)
        # convertToComm :280-288
        comm = []
        for a, p, nv in zip(arrays, self.parts, nvals):
            c = np.zeros_like(a).reshape(-1, nv)
            c[p.comm_index[d]] = a.reshape(-1, nv)
            comm.append(c)
        red = [c.copy() for c in comm]
        if op != BCAST_OP:  # fan in :300-385: the owner combines, own value first, then increasing rank
            for p in self.parts:
                o0, o1 = p.nents_offsets[d][p.rank], p.nents_offsets[d][p.rank + 1]
                seg = red[p.rank][o0:o1]
                for q in self.parts:
                    if q.rank == p.rank:
                        continue
                    kind = q.is_complete[d][p.rank]
                    if kind == 0:
                        continue
                    q0, q1 = q.nents_offsets[d][p.rank], q.nents_offsets[d][p.rank + 1]
                    contrib = comm[q.rank][q0:q1]
                    idx = np.arange(o1 - o0) if kind == 2 else p.bounded_ent_ids[d][
                        p.bounded_offset[d][q.rank]:p.bounded_offset[d][q.rank + 1]]
                    if op == SUM_OP:
                        seg[idx] = seg[idx] + contrib
                    elif op == MAX_OP:
                        seg[idx] = np.maximum(seg[idx], contrib)
                    else:
                        seg[idx] = np.minimum(seg[idx], contrib)
        out = []
        for p in self.parts:  # fan out :386-428: every part takes the owners' segments
            c = red[p.rank].copy()
            for r in range(cs):
                if r == p.rank or p.is_complete[d][r] == 0:
                    continue
                owner_part = self.parts[r]
                o0 = owner_part.nents_offsets[d][r]
                q0, q1 = p.nents_offsets[d][r], p.nents_offsets[d][r + 1]
                if p.is_complete[d][r] == 2:
                    c[q0:q1] = red[r][o0:o0 + (q1 - q0)]
                else:
                    idx = owner_part.bounded_ent_ids[d][
                        owner_part.bounded_offset[d][p.rank]:owner_part.bounded_offset[d][p.rank + 1]]
                    c[q0:q1] = red[r][o0 + idx]
            out.append(c[p.comm_index[d]].reshape(-1))  # convertFromComm :432-438
        return out


# ------------------------------------------------------------------------------------------------
# Particle load balancer, src/pumipic_lb.hpp / pumipic_lb.cpp.
#
# What the reference computes itself is restated as is: the "sbars" (ParticleBalancer::ParticleBalancer,
# pumipic_lb.cpp:23-135: for every element the set of parts on which it is safe, its owner included), the
# weights (addWeights, pumipic_lb.hpp:138-237: particles that stay on this rank by the sbar of their new
# element; particles already sent elsewhere count on the destination) and the selection (selectParticles,
# :239-350: per sbar a list of (target part, weight), consumed particle by particle, particles outside the
# core first).
#
# The balancing step itself is NOT in the reference: ParticleBalancer::balance (pumipic_lb.cpp:495-531)
# hands the weighted N-graph to EnGPar (scorec/EnGPar >= 1.1.0, CMakeLists.txt:89-91:
# engpar::createWeightInput + engpar::balanceWeights), which is absent from /root/reference.  EnGPar's
# weight balancer is an iterative diffusion (Diamond, Smith, Shephard, "Dynamic load balancing of
# massively parallel unstructured meshes", ScalA'17): in every round an overloaded part sends
# step_factor x (its weight - the neighbour's weight) towards every lighter neighbour, through the graph
# edges (here: the sbars) the two parts share, until the imbalance target is met.  `diffuse` below is
# that scheme in integers, with the two constraints a one-step particle migration adds: a part can only
# send what it holds in an sbar at the start (no forwarding of weight it is about to receive), and the
# request towards each neighbour is divided by the number of lighter neighbours (no overshoot).
# PARITY UNPINNED against EnGPar's own numbers; the reference's test (test/test_lb.cpp: imbalance <= 1.3
# after one `partition`, <= 1.5 after two `repartition` + migrate rounds) is what the tests assert.
def element_sbars(pic):
    """bit r of mask[e] (full-mesh element e) = e is safe on part r or owned by r (pumipic_lb.cpp:36-75 +
    buildLocalSbarMap :95-112)"""
    dim = pic.mesh.dim
    owner = pic.owner[dim]
    mask = np.left_shift(np.uint64(1), owner.astype(np.uint64))
    for p in pic.parts:
        held = p.has_part[owner].astype(bool)
        safe = p.is_safe_full.astype(bool) & held
        mask[safe] |= np.uint64(1) << np.uint64(p.rank)
    return mask


def diffuse(masks, weights, forced_in, tol, step_factor, max_iters=50):
    """masks: sorted distinct sbar masks (python ints); weights[r][i]: particles of rank r in sbar i (0 where
    the sbar does not contain r); forced_in[r]: particles other ranks already send to r.
    -> plan[r] = list of (sbar index, target, amount), sbar ascending then target ascending."""
    P = len(weights)
    avail = [list(map(int, w)) for w in weights]
    W = [sum(avail[r]) + int(forced_in[r]) for r in range(P)]
    total = sum(W)
    send = [dict() for _ in range(P)]
    for _ in range(max_iters):
        if max(W) * P <= tol * total:
            break
        W0 = list(W)
        moved = False
        for p in range(P):
            if W0[p] * P <= total:
                continue  # only parts above the average send
            nbrs = [q for q in range(P) if q != p and W0[q] < W0[p] and
                    any(avail[p][i] > 0 and (m >> q) & 1 and (m >> p) & 1 for i, m in enumerate(masks))]
            for q in nbrs:
                want = int(step_factor * (W0[p] - W0[q]) / len(nbrs))
                for i, m in enumerate(masks):
                    if want <= 0:
                        break
                    if not ((m >> q) & 1 and (m >> p) & 1):
                        continue
                    t = min(avail[p][i], want)
                    if t <= 0:
                        continue
                    avail[p][i] -= t
                    send[p][(i, q)] = send[p].get((i, q), 0) + t
                    W[p] -= t
                    W[q] += t
                    want -= t
                    moved = True
        if not moved:
            break
    return [[(i, q, t) for (i, q), t in sorted(s.items())] for s in send], W


class Balancer:
    """ParticleBalancer over all simulated ranks"""

    def __init__(self, pic):
        self.pic = pic
        self.full_mask = element_sbars(pic)
        self.masks = sorted(int(m) for m in np.unique(self.full_mask))
        lut = {m: i for i, m in enumerate(self.masks)}
        self.full_index = np.array([lut[int(m)] for m in self.full_mask], dtype=np.int32)
        dim = pic.mesh.dim
        self.part_index = [self.full_index[p.full_ids[dim]] for p in pic.parts]  # getSbarIDs per part

    def weights(self, new_elems, new_procs):
        """addWeights (pumipic_lb.hpp:138-217) for every rank: new_elems[r] / new_procs[r] per live particle of
        rank r (part-local element ids, -1 = leaving the domain)"""
        P = self.pic.comm_size
        w = np.zeros((P, len(self.masks)), dtype=np.int64)
        forced = np.zeros(P, dtype=np.int64)
        for r in range(P):
            e, pr = np.asarray(new_elems[r]), np.asarray(new_procs[r])
            stay = (pr == r) & (e != -1)
            idx = self.part_index[r][e[stay]]
            has_me = np.array([(m >> r) & 1 for m in self.masks], dtype=bool)
            idx = idx[has_me[idx]]
            np.add.at(w[r], idx, 1)
            np.add.at(forced, pr[pr != r], 1)
        return w, forced

    def plan(self, w, forced, tol, step_factor=0.3):
        return diffuse(self.masks, w, forced, tol, step_factor)

    def partition_counts(self, ptcls_per_elem, tol, step_factor=0.3):
        """ParticleBalancer::partition (pumipic_lb.hpp:364-377): particles per element of every part -> the
        plan and the number of particles every rank ends with"""
        P = self.pic.comm_size
        w = np.zeros((P, len(self.masks)), dtype=np.int64)
        for r in range(P):
            has_me = np.array([(m >> r) & 1 for m in self.masks], dtype=bool)
            idx = self.part_index[r]
            sel = has_me[idx]
            np.add.at(w[r], idx[sel], np.asarray(ptcls_per_elem[r])[sel])
        plan, W = self.plan(w, np.zeros(P, dtype=np.int64), tol, step_factor)
        return plan, W, w
