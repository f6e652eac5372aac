// pp_search.hip -- element-to-element adjacency searches, whole walk inside one kernel.
//
// The reference drives the walk from the host: per iteration three ps::parallel_for kernels, a
// min-reduction of ptcl_done and a D2H scalar read (src/pumipic_adjacency.tpp:558-608,
// adjacency.hpp:1066-1150).  A particle's walk never depends on another particle, so here each
// thread walks its own particle to completion; the result per particle is identical to the
// reference's lock-step loop (a particle that needs more than `looplimit` steps is marked -1
// exactly as ptclsNotFound does, tpp:584-606).
//
//   search_mesh_2d            src/pumipic_adjacency.hpp:1011-1158
//   search_mesh (new)         src/pumipic_adjacency.tpp:72-145,231-416,460-654
//   search_mesh (legacy 3-D)  src/pumipic_adjacency.hpp:558-768
//
// BCC walks read ONE packed record per visited element (tri 64 B / tet 128 B: vertex coords,
// neighbour ids with -1 = exposed side, class id, measure) instead of the reference's chain of
// dependent gathers elem2verts -> coords, elem2sides -> exposed -> side2elems.
#include "pp_geom.hpp"
#include <atomic>
#include <chrono>
#include "pp_internal.hpp"
#include "pp_push_math.hpp"

namespace {
using pp::grid_for;
using pp::kBlock;
using namespace ppg;

constexpr int kHardLoopCap = 1 << 22;  // guards looplimit==0 against a GPU hang

struct Counters {
  int not_found;   // particles cut off by looplimit
  int not_in_elem; // check_initial_parents failures
  int aborted;     // legacy search: origin not in start element at loops==0 (OMEGA_H_CHECK)
  int pending;     // fused kernel: entries in the deferred-walk queue
  int unmoved;     // trusted-origin mode: particles that finished as "unmoved" (no containment test ran)
};
// landing zone of search_mesh_2d's result in host-mapped memory: a one-thread kernel behind the search writes the
// count and then the stamp, the host polls the stamp -- the `bool found` every search returns (adjacency.hpp:1011-1020)
// without a stream synchronisation (~25 us of idle GPU per step of the drop-in loop)
struct FoundPin {
  int stamp;
  int not_found;
};

__device__ __forceinline__ void load_tri(const pp_tri_rec* __restrict__ recs, int e, V2 fc[3],
                                         int nbr[3]) {
  const pp_tri_rec* r = recs + e;
  for (int i = 0; i < 3; ++i) {
    fc[i] = {r->xy[i][0], r->xy[i][1]};
    nbr[i] = r->nbr[i];
  }
}
__device__ __forceinline__ void load_tet(const pp_tet_rec* __restrict__ recs, int e, V3 M[4],
                                         int nbr[4], double& vol) {
  const pp_tet_rec* r = recs + e;
  for (int i = 0; i < 4; ++i) {
    M[i] = {r->xyz[i][0], r->xyz[i][1], r->xyz[i][2]};
    nbr[i] = r->nbr[i];
  }
  vol = r->vol;
}

// one BCC step in a triangle: returns done; sets next (neighbour across the arg-min edge, -1 if
// that edge is exposed).  barycentric_tri + all_positive + min3 (hpp:1072-1081 / tpp:250-260)
__device__ __forceinline__ bool step_tri(const pp_tri_rec* __restrict__ recs, int elem, V2 pos,
                                         int& next) {
  V2 fc[3];
  int nbr[3];
  load_tri(recs, elem, fc, nbr);
  const double area = tri_area(fc);  // == measure_elements_real(elem), same expression
  double bcc[3];
  barycentric_tri(area, fc, pos, bcc);
  const bool done = all_positive3(bcc, kEpsilon);
  next = nbr[min3(bcc)];
  return done;
}
__device__ __forceinline__ bool step_tet(const pp_tet_rec* __restrict__ recs, int elem, V3 pos,
                                         int& next) {
  V3 M[4];
  int nbr[4];
  double vol;
  load_tet(recs, elem, M, nbr, vol);
  double bcc[4];
  barycentric_tet(vol, M, pos, bcc);
  const bool done = all_positive4(bcc, kEpsilon);
  next = nbr[min_index4(bcc)];
  return done;
}

// BCC walk shared by search_mesh_2d, search_mesh(BCC) and the fused kernel.
// Per iteration (reference kernel order): find exit -> exposed? -> next element -> looplimit.
template <int DIM>
__device__ __forceinline__ int bcc_walk(const void* __restrict__ recs, int elem, V3 pos,
                                        int looplimit, Counters* cnt, int loops = 0) {
  bool done = false;
  const int cap = looplimit ? looplimit : kHardLoopCap;
  while (true) {
    int next;
    if (DIM == 2)
      done = step_tri((const pp_tri_rec*)recs, elem, V2{pos.x, pos.y}, next);
    else
      done = step_tet((const pp_tet_rec*)recs, elem, pos, next);
    if (!done) {
      if (next == -1) {  // exposed side: leaves the domain
        elem = -1;
        done = true;
      } else {
        elem = next;
      }
    }
    ++loops;
    if (done) break;
    if (loops >= cap) {
      elem = -1;
      atomicAdd(&cnt->not_found, 1);
      break;
    }
  }
  return elem;
}

// ------------------------------------------------------------------ search_mesh_2d
__global__ void k_search2d(int capacity, const unsigned char* __restrict__ mask,
                           const int* __restrict__ slot_elem, const pp_tri_rec* __restrict__ recs,
                           int nelems, const double* __restrict__ xt, long long stride,
                           int* __restrict__ elem_ids, int looplimit, Counters* cnt) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  const int e = slot_elem[pid];
  if (e < 0) return;
  if (!mask[pid]) {
    elem_ids[pid] = -1;
    return;
  }
  int elem = elem_ids[pid];
  if (elem == -1) elem = e;
  if (elem == -nelems) {  // hpp:1051-1056
    elem_ids[pid] = -1;
    return;
  }
  const V3 pos{xt[pid], xt[stride + pid], 0.0};
  elem_ids[pid] = bcc_walk<2>(recs, elem, pos, looplimit, cnt);
}

// ------------------------------------------------------------------ search_mesh (tpp)
struct MeshArrays {
  const double* coords;
  const int* elem2verts;
  const int* elem2sides;
  const int* side2verts;
  const int* side2elems_off;
  const int* side2elems;
  const signed char* side_exposed;
  const double* elem_measure;
  const int* dual_off;
  const int* dual_elems;
};

template <int DIM>
__device__ __forceinline__ bool origin_inside(const void* __restrict__ recs, int elem, V3 orig,
                                              double tol) {
  if (DIM == 2) {
    V2 fc[3];
    int nbr[3];
    load_tri((const pp_tri_rec*)recs, elem, fc, nbr);
    double bcc[3];
    barycentric_tri(tri_area(fc), fc, V2{orig.x, orig.y}, bcc);
    return all_positive3(bcc, tol);
  } else {
    V3 M[4];
    int nbr[4];
    double vol;
    load_tet((const pp_tet_rec*)recs, elem, M, nbr, vol);
    double bcc[4];
    barycentric_tet(vol, M, orig, bcc);
    return all_positive4(bcc, tol);
  }
}

__device__ __forceinline__ int other_elem(const MeshArrays& m, int bridge, int searchElm) {
  const int first = m.side2elems_off[bridge];
  const int A = m.side2elems[first], B = m.side2elems[first + 1];
  return (A == searchElm) ? B : A;
}

// Intersection-mode exit search in one element (tpp:284-361).  Returns lastExit (side id or -1)
template <int DIM>
__device__ __forceinline__ int exit_by_intersection(const MeshArrays& m, int searchElm, V3 orig,
                                                    V3 dest, double tol, int prevExit,
                                                    double ip[3]) {
  int lastExit = -1;
  if (DIM == 2) {
    int fverts[3];
    for (int i = 0; i < 3; ++i) fverts[i] = m.elem2verts[(size_t)searchElm * 3 + i];
    V2 xpts{0, 0};
    for (int ei = 0; ei < 3; ++ei) {
      const int edge_id = m.elem2sides[(size_t)searchElm * 3 + ei];
      if (edge_id == prevExit) continue;
      int ev2v[2];
      V2 edge[2];
      for (int q = 0; q < 2; ++q) {
        ev2v[q] = m.side2verts[(size_t)edge_id * 2 + q];
        edge[q] = {m.coords[(size_t)ev2v[q] * 2], m.coords[(size_t)ev2v[q] * 2 + 1]};
      }
      const int flip = is_edge_flipped(ev2v, fverts);
      const bool success = line_edge_2d(edge, V2{orig.x, orig.y}, V2{dest.x, dest.y}, xpts, tol, flip);
      if (success) {
        lastExit = edge_id;
        ip[0] = xpts.x;
        ip[1] = xpts.y;
      }
    }
  } else {
    int tetv2v[4];
    for (int i = 0; i < 4; ++i) tetv2v[i] = m.elem2verts[(size_t)searchElm * 4 + i];
    V3 xpts{0, 0, 0};
    double quality = -1;
    int bestFace = -1;
    for (int fi = 0; fi < 4; ++fi) {
      const int face_id = m.elem2sides[(size_t)searchElm * 4 + fi];
      if (face_id == prevExit) continue;
      int fv2v[3];
      V3 face[3];
      for (int q = 0; q < 3; ++q) {
        fv2v[q] = m.side2verts[(size_t)face_id * 3 + q];
        face[q] = {m.coords[(size_t)fv2v[q] * 3], m.coords[(size_t)fv2v[q] * 3 + 1],
                   m.coords[(size_t)fv2v[q] * 3 + 2]};
      }
      const int flip = is_face_flipped(fi, fv2v, tetv2v);
      double dproj, closeness, param;
      const bool success =
          ray_intersects_triangle(face, orig, dest, xpts, tol, flip, dproj, closeness, param);
      if (success) {
        lastExit = face_id;
        ip[0] = xpts.x;
        ip[1] = xpts.y;
        ip[2] = xpts.z;
      }
      if (dproj > -tol && (quality < 0 || closeness < quality) && lastExit == -1) {
        quality = closeness;
        bestFace = face_id;
        ip[0] = xpts.x;
        ip[1] = xpts.y;
        ip[2] = xpts.z;
      }
    }
    if (lastExit == -1) lastExit = bestFace;
  }
  return lastExit;
}

// search_findExitFace_intersect_2d (tpp:287-311) on the packed 64-B triangle record.  Edge ei of a triangle is
// {v_ei, v_ei+1}; the reference hands line_edge_2d the STORED edge and isFaceFlipped (utils.hpp:495-499), which
// together always select (v_ei, v_ei+1) -- flip is true exactly when the side is stored the other way round
// and then swaps it back (line_edge_2d: vtx1 = flip, vtx2 = !flip) -- so the record's own vertices in template
// order with flip = 0 are the reference's operands.  The edge the path came in through (edge_id == prevExit) is
// the one with the previous element behind it (pp_mesh::mt_packed_ok).  Returns the local edge index or -1.
__device__ __forceinline__ int exit_by_intersection_packed2(const pp_tri_rec* __restrict__ recs, int elem, int prev,
                                                            V3 orig, V3 dest, double tol, double ip[3], int nbr[3]) {
  V2 fc[3];
  load_tri(recs, elem, fc, nbr);
  int lastExit = -1;
  V2 xpts{0, 0};
  for (int ei = 0; ei < 3; ++ei) {
    if (prev >= 0 && nbr[ei] == prev) continue;
    const V2 edge[2] = {fc[ei], fc[(ei + 1) % 3]};
    if (line_edge_2d(edge, V2{orig.x, orig.y}, V2{dest.x, dest.y}, xpts, tol, 0)) {
      lastExit = ei;
      ip[0] = xpts.x;
      ip[1] = xpts.y;
    }
  }
  return lastExit;
}

template <int DIM, bool MT>
__global__ void k_search_tpp(int capacity, const unsigned char* __restrict__ mask,
                             const int* __restrict__ slot_elem, const void* __restrict__ recs,
                             MeshArrays m, const double* __restrict__ x,
                             const double* __restrict__ xt, long long stride,
                             int* __restrict__ elem_ids, int seeded, double tol,
                             int* __restrict__ inter_faces, double* __restrict__ inter_points,
                             int looplimit, Counters* cnt, int packed2 = 0) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  const int e = slot_elem[pid];
  if (e < 0) {  // tail slots of a CSR: an elem_ids the search allocates is -1 there (tpp:506)
    if (!seeded) elem_ids[pid] = -1;
    return;
  }
  const bool msk = mask[pid];
  if (MT) {  // initializeIntersection visits every slot (tpp:542-547)
    for (int i = 0; i < DIM; ++i) inter_points[(size_t)DIM * pid + i] = 0;
    inter_faces[pid] = -1;
  }
  int elem;
  bool done;
  if (!seeded) {  // tpp:504-515
    elem = msk ? e : -1;
    done = !msk;
  } else {  // tpp:516-522
    elem = elem_ids[pid];
    done = (msk && elem == -1) || !msk;
  }
  if (!msk) {
    if (!seeded) elem_ids[pid] = -1;
    return;
  }
  const V3 orig{x[pid], x[stride + pid], x[2 * stride + pid]};
  const V3 dest{xt[pid], xt[stride + pid], xt[2 * stride + pid]};
  if (norm(sub(dest, orig)) < tol) done = true;  // finishUnmoved tpp:525-533
  if (!done) {                                   // check_initial_parents tpp:72-145
    if (!origin_inside<DIM>(recs, elem, orig, tol)) {
      atomicAdd(&cnt->not_in_elem, 1);
      elem = -1;
      done = true;
    }
  }
  if (!done) {
    if (!MT) {
      elem = bcc_walk<DIM>(recs, elem, dest, looplimit, cnt);
    } else {
      int lastExit = -1, loops = 0, xface = -1;
      double ip[3] = {0, 0, 0};
      const int cap = looplimit ? looplimit : kHardLoopCap;
      int prev = -1;
      while (DIM == 2 && packed2) {  // triangles: the walk on the packed records (one 64-B record per element)
        int nbr[3];
        const int le = exit_by_intersection_packed2((const pp_tri_rec*)recs, elem, prev, orig, dest, tol, ip, nbr);
        done = (le == -1);
        if (!done) {  // check_model_intersection tpp:372-385 (requireIntersection == true)
          const int nx = le == 0 ? nbr[0] : le == 1 ? nbr[1] : nbr[2];
          done = nx == -1;
          if (done) {
            xface = m.elem2sides[(size_t)elem * 3 + le];
          } else {  // set_new_element tpp:397-414
            prev = elem;
            elem = nx;
          }
        }
        ++loops;
        if (done) break;
        if (loops >= cap) {
          elem = -1;
          atomicAdd(&cnt->not_found, 1);
          break;
        }
      }
      while (!(DIM == 2 && packed2)) {
        lastExit = exit_by_intersection<DIM>(m, elem, orig, dest, tol, lastExit, ip);
        done = (lastExit == -1);
        if (!done) {  // check_model_intersection tpp:372-385 (requireIntersection == true)
          const bool exposed = m.side_exposed[lastExit];
          done = exposed;
          if (exposed) xface = lastExit;
        }
        if (!done) elem = other_elem(m, lastExit, elem);  // set_new_element tpp:397-414
        ++loops;
        if (done) break;
        if (loops >= cap) {
          elem = -1;
          atomicAdd(&cnt->not_found, 1);
          break;
        }
      }
      for (int i = 0; i < DIM; ++i) inter_points[(size_t)DIM * pid + i] = ip[i];
      inter_faces[pid] = xface;
    }
  }
  elem_ids[pid] = elem;
}

// ------------------------------------------------------------------ legacy 3-D search_mesh
__global__ void k_search_legacy3d(int capacity, const unsigned char* __restrict__ mask,
                                  const int* __restrict__ slot_elem, MeshArrays m,
                                  const double* __restrict__ x, const double* __restrict__ xt,
                                  long long stride, int* __restrict__ elem_ids, int seeded,
                                  double* __restrict__ xpoints_d, int* __restrict__ xface_d,
                                  int looplimit, Counters* cnt) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  const int e = slot_elem[pid];
  if (e < 0) {
    if (!seeded) elem_ids[pid] = -1;
    return;
  }
  if (!mask[pid]) {  // hpp:593-596
    elem_ids[pid] = -1;
    return;
  }
  const double tol = 1.0e-10;
  int elmId = seeded ? elem_ids[pid] : e;
  if (elmId == -1) {  // ptcl_done from the start: elem_ids_next stays -1 (hpp:580,746)
    elem_ids[pid] = -1;
    return;
  }
  const V3 orig{x[pid], x[stride + pid], x[2 * stride + pid]};
  const V3 dest{xt[pid], xt[stride + pid], xt[2 * stride + pid]};
  int loops = 0;
  int result = -1;
  bool done = false;
  // the reference breaks when loops > looplimit (hpp:757): looplimit+1 iterations are allowed
  const int cap = looplimit ? looplimit + 1 : kHardLoopCap;
  while (true) {
    int tetv2v[4];
    V3 M[4];
    for (int i = 0; i < 4; ++i) {
      tetv2v[i] = m.elem2verts[(size_t)elmId * 4 + i];
      M[i] = {m.coords[(size_t)tetv2v[i] * 3], m.coords[(size_t)tetv2v[i] * 3 + 1],
              m.coords[(size_t)tetv2v[i] * 3 + 2]};
    }
    double bcc[4];
    if (loops == 0) {
      find_barycentric_tet(M, orig, bcc);
      if (!all_positive4(bcc, tol)) atomicAdd(&cnt->aborted, 1);
    }
    int next = -1;
    find_barycentric_tet(M, dest, bcc);
    if (all_positive4(bcc, tol)) {
      next = elmId;
      done = true;
    } else {
      double dproj[4] = {-1, -1, -1, -1};
      double xps[12] = {0};
      int exposed_faces[4] = {0, 0, 0, 0};
      int xface_ids[4] = {-1, -1, -1, -1};
      int dual_elem_id = m.dual_off[elmId];
      bool intersected = false;
      int findex = 0;
      for (int lf = 0; lf < 4; ++lf) {
        const int face_id = m.elem2sides[(size_t)elmId * 4 + lf];
        const bool exposed = m.side_exposed[face_id];
        exposed_faces[findex] = exposed;
        xface_ids[findex] = face_id;
        int fv2v[3];
        V3 face[3];
        for (int q = 0; q < 3; ++q) {
          fv2v[q] = m.side2verts[(size_t)face_id * 3 + q];
          face[q] = {m.coords[(size_t)fv2v[q] * 3], m.coords[(size_t)fv2v[q] * 3 + 1],
                     m.coords[(size_t)fv2v[q] * 3 + 2]};
        }
        const int m1 = face_map(findex * 2), m2 = face_map(findex * 2 + 1);
        bool flip = true;
        if (fv2v[1] == tetv2v[m1] && fv2v[2] == tetv2v[m2]) flip = false;  // hpp:660-664
        V3 xpoint;
        intersected = line_triangle_intx_simple(face, orig, dest, xpoint, dproj[findex], flip, tol);
        xps[findex * 3] = xpoint.x;
        xps[findex * 3 + 1] = xpoint.y;
        xps[findex * 3 + 2] = xpoint.z;
        if (intersected && exposed) {
          done = true;
          xpoints_d[(size_t)pid * 3] = xpoint.x;
          xpoints_d[(size_t)pid * 3 + 1] = xpoint.y;
          xpoints_d[(size_t)pid * 3 + 2] = xpoint.z;
          xface_d[pid] = face_id;
          next = -1;
          break;
        } else if (intersected && !exposed) {
          next = m.dual_elems[dual_elem_id];
          break;
        }
        if (!exposed) ++dual_elem_id;
        ++findex;
      }
      if (!intersected) {
        const int max_ind = max_index4(dproj);
        if (dproj[max_ind] >= 0) {
          const int fid = xface_ids[max_ind];
          if (exposed_faces[max_ind]) {
            next = -1;
            for (int i = 0; i < 3; ++i) xpoints_d[(size_t)pid * 3 + i] = xps[max_ind * 3 + i];
            xface_d[pid] = fid;
            done = true;
          } else {
            // the reference indexes the dual value array by a face id here
            // (hpp:726); not replicated -- neighbour across the max-dproj face.
            next = other_elem(m, fid, elmId);
          }
        } else {
          next = -1;
          done = true;
        }
      }
    }
    result = next;  // elem_ids <- elem_ids_next after every iteration (hpp:745-748)
    ++loops;
    if (done) break;
    if (loops >= cap) {
      atomicAdd(&cnt->not_found, 1);
      break;  // particle keeps its current element id, as in the reference
    }
    elmId = next;
  }
  elem_ids[pid] = result;
}

// ------------------------------------------------------------------ stepwise walk
// trace_particle_through_mesh (tpp:460-615) accepts a caller-supplied functor that runs between
// find_exit_face and set_new_element of every walk iteration.  A functor cannot be compiled into
// the fused kernels of this library, so the walk is also exposed kernel by kernel; the mirror
// header drives these in the reference's order with the user's functor (any device code) between.
template <int DIM>
__global__ void k_trace_begin(int capacity, const unsigned char* __restrict__ mask,
                              const int* __restrict__ slot_elem, const void* __restrict__ recs,
                              const double* __restrict__ x, const double* __restrict__ xt,
                              long long stride, int* __restrict__ elem_ids, int seeded, double tol,
                              int mt, int* __restrict__ inter_faces,
                              double* __restrict__ inter_points, int* __restrict__ ptcl_done,
                              int* __restrict__ last_exit, Counters* cnt) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  last_exit[pid] = -1;
  const int e = slot_elem[pid];
  if (e < 0) {  // tail slots no parallel_for visits
    ptcl_done[pid] = 1;
    if (!seeded) elem_ids[pid] = -1;
    return;
  }
  const bool msk = mask[pid];
  if (mt) {  // initializeIntersection tpp:542-547
    for (int i = 0; i < DIM; ++i) inter_points[(size_t)DIM * pid + i] = 0;
    inter_faces[pid] = -1;
  }
  int elem;
  bool done;
  if (!seeded) {  // tpp:504-515
    elem = msk ? e : -1;
    done = !msk;
    elem_ids[pid] = elem;
  } else {  // tpp:516-522
    elem = elem_ids[pid];
    done = (msk && elem == -1) || !msk;
  }
  if (msk) {
    const V3 orig{x[pid], x[stride + pid], x[2 * stride + pid]};
    const V3 dest{xt[pid], xt[stride + pid], xt[2 * stride + pid]};
    if (norm(sub(dest, orig)) < tol) done = true;              // finishUnmoved tpp:525-533
    if (!done && !origin_inside<DIM>(recs, elem, orig, tol)) {  // check_initial_parents tpp:72-145
      atomicAdd(&cnt->not_in_elem, 1);
      elem_ids[pid] = -1;
      done = true;
    }
  }
  ptcl_done[pid] = done;
}
// find_exit_face tpp:231-363
template <int DIM>
__global__ void k_trace_find_exit(int capacity, const unsigned char* __restrict__ mask,
                                  const void* __restrict__ recs, MeshArrays m,
                                  const double* __restrict__ x, const double* __restrict__ xt,
                                  long long stride, const int* __restrict__ elem_ids,
                                  int* __restrict__ ptcl_done, int* __restrict__ last_exit,
                                  double* __restrict__ inter_points, int use_bcc, double tol) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid] || ptcl_done[pid]) return;
  const int elem = elem_ids[pid];
  const V3 dest{xt[pid], xt[stride + pid], xt[2 * stride + pid]};
  if (use_bcc) {
    int idx;
    bool done;
    if (DIM == 2) {
      V2 fc[3];
      int nbr[3];
      load_tri((const pp_tri_rec*)recs, elem, fc, nbr);
      double bcc[3];
      barycentric_tri(tri_area(fc), fc, V2{dest.x, dest.y}, bcc);
      done = all_positive3(bcc, kEpsilon);
      idx = min3(bcc);
    } else {
      V3 M[4];
      int nbr[4];
      double vol;
      load_tet((const pp_tet_rec*)recs, elem, M, nbr, vol);
      double bcc[4];
      barycentric_tet(vol, M, dest, bcc);
      done = all_positive4(bcc, kEpsilon);
      idx = min_index4(bcc);
    }
    ptcl_done[pid] = done;
    last_exit[pid] = m.elem2sides[(size_t)elem * (DIM + 1) + idx];
  } else {
    const V3 orig{x[pid], x[stride + pid], x[2 * stride + pid]};
    double ip[3];
    for (int i = 0; i < DIM; ++i) ip[i] = inter_points[(size_t)DIM * pid + i];
    const int ex = exit_by_intersection<DIM>(m, elem, orig, dest, tol, last_exit[pid], ip);
    for (int i = 0; i < DIM; ++i) inter_points[(size_t)DIM * pid + i] = ip[i];
    last_exit[pid] = ex;
    ptcl_done[pid] = (ex == -1);
  }
}
// check_model_intersection tpp:365-387 (the default functor, RemoveParticleOnGeometricModelExit)
__global__ void k_trace_check_model(int capacity, const unsigned char* __restrict__ mask,
                                    const signed char* __restrict__ side_exposed,
                                    int* __restrict__ elem_ids, int* __restrict__ ptcl_done,
                                    const int* __restrict__ last_exit, int require_intersection,
                                    int* __restrict__ inter_faces) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid] || ptcl_done[pid]) return;
  const int bridge = last_exit[pid];
  const bool exposed = side_exposed[bridge];
  ptcl_done[pid] = exposed;
  if (exposed && require_intersection)
    inter_faces[pid] = bridge;
  else if (exposed)
    elem_ids[pid] = -1;
}
// set_new_element tpp:389-416 + the min reduction over ptcl_done (tpp:567-571) as a count of
// unfinished slots.  A functor that leaves a particle unfinished on an exposed side would make the
// reference read past the side's single up-adjacent element; here that particle leaves (-1).
__global__ void k_trace_set_new_element(int capacity, const unsigned char* __restrict__ mask,
                                        MeshArrays m, int* __restrict__ elem_ids,
                                        const int* __restrict__ ptcl_done,
                                        const int* __restrict__ last_exit, int* __restrict__ not_done) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool open = pid < capacity && !ptcl_done[pid];
  if (open && mask[pid]) {
    const int bridge = last_exit[pid];
    const int first = m.side2elems_off[bridge];
    if (m.side2elems_off[bridge + 1] - first < 2) {
      elem_ids[pid] = -1;
    } else {
      const int A = m.side2elems[first], B = m.side2elems[first + 1];
      elem_ids[pid] = (A == elem_ids[pid]) ? B : A;
    }
  }
  const unsigned long long b = __ballot(open);
  if (b && (threadIdx.x & 63) == 0) atomicAdd(not_done, __popcll(b));
}
// ptclsNotFound tpp:583-600
__global__ void k_trace_not_found(int capacity, const unsigned char* __restrict__ mask,
                                  int* __restrict__ elem_ids, const int* __restrict__ ptcl_done,
                                  int* __restrict__ count) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool lost = pid < capacity && mask[pid] && !ptcl_done[pid];
  if (lost) elem_ids[pid] = -1;
  const unsigned long long b = __ballot(lost);
  if (b && (threadIdx.x & 63) == 0) atomicAdd(count, __popcll(b));
}

// ------------------------------------------------------------------ search_mesh_3d
// src/pumipic_adjacency.hpp:314-555.  The reference runs checkCurrentElm / findIntersection /
// processUndetected as three launches per walk iteration; a particle's walk depends on nothing
// but its own state, so one thread carries it to the end.  tol = 1e-20 (hpp:330); neighbours in
// ask_dual order; SURVEY Q3 (dual_elems[face_id], hpp:510) not replicated.
__device__ __forceinline__ bool point_within_tet(const MeshArrays& m, V3 pos, int elem, double tol) {
  V3 M[4];
  for (int i = 0; i < 4; ++i) {
    const int v = m.elem2verts[(size_t)elem * 4 + i];
    M[i] = {m.coords[(size_t)v * 3], m.coords[(size_t)v * 3 + 1], m.coords[(size_t)v * 3 + 2]};
  }
  double bcc[4];
  barycentric_coords_tet(M, pos, bcc, tol);  // isPointWithinElemTet hpp:300-305
  return all_positive4(bcc, tol);
}
__global__ void k_search_mesh3d(int capacity, const unsigned char* __restrict__ mask,
                                const int* __restrict__ slot_elem, MeshArrays m,
                                const double* __restrict__ x, const double* __restrict__ xt,
                                long long stride, int* __restrict__ elem_ids, int seeded,
                                double* __restrict__ xpoints_d, int* __restrict__ xface_d,
                                int looplimit, Counters* cnt) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  const int e = slot_elem[pid];
  if (e < 0 || !mask[pid]) {  // fill hpp:365-368
    elem_ids[pid] = -1;
    return;
  }
  const double tol = 1.0e-20;
  int elm = seeded ? elem_ids[pid] : e;
  if (elm == -1) return;  // ptcl_done = 2 from the start; elem_ids_next stays -1
  const V3 orig{x[pid], x[stride + pid], x[2 * stride + pid]};
  const V3 dest{xt[pid], xt[stride + pid], xt[2 * stride + pid]};
  if (!point_within_tet(m, orig, e, tol)) atomicAdd(&cnt->aborted, 1);  // checkParent hpp:371-382
  const int cap = looplimit ? looplimit : kHardLoopCap;
  int loops = 0;
  while (true) {
    int next = elm;
    bool finished = point_within_tet(m, dest, elm, tol);  // checkCurrentElm
    if (!finished) {                                      // findIntersection
      int tetv2v[4];
      for (int i = 0; i < 4; ++i) tetv2v[i] = m.elem2verts[(size_t)elm * 4 + i];
      int dual_elem_id = m.dual_off[elm];
      int adj_id = -1, ind_exp = -1;
      double projd[4] = {-1, -1, -1, -1};  // processUndetected's initial values (hpp:484)
      double xps[12];
      int face_ids[4];
      for (int fi = 0; fi < 4; ++fi) {
        const int face_id = m.elem2sides[(size_t)elm * 4 + fi];
        face_ids[fi] = face_id;
        int fv2v[3];
        V3 face[3];
        for (int q = 0; q < 3; ++q) {
          fv2v[q] = m.side2verts[(size_t)face_id * 3 + q];
          face[q] = {m.coords[(size_t)fv2v[q] * 3], m.coords[(size_t)fv2v[q] * 3 + 1],
                     m.coords[(size_t)fv2v[q] * 3 + 2]};
        }
        const bool flip = is_face_flipped(fi, fv2v, tetv2v);
        V3 xpoint;
        const bool det = line_triangle_intx_simple(face, orig, dest, xpoint, projd[fi], flip, tol);
        xps[fi * 3] = xpoint.x;
        xps[fi * 3 + 1] = xpoint.y;
        xps[fi * 3 + 2] = xpoint.z;
        const bool exposed = m.side_exposed[face_id];
        if (det && exposed) ind_exp = fi;  // no break: the last detected face wins (hpp:437-445)
        if (det && !exposed) adj_id = dual_elem_id;
        if (!exposed) ++dual_elem_id;
      }
      int done = 0;
      if (ind_exp >= 0) {  // wall collision
        for (int i = 0; i < 3; ++i) xpoints_d[(size_t)pid * 3 + i] = xps[ind_exp * 3 + i];
        xface_d[pid] = face_ids[ind_exp];
        next = -1;
        done = 2;
      }
      if (adj_id >= 0) {  // interior (overrides a wall hit of the same iteration, hpp:462-470)
        next = m.dual_elems[adj_id];
        done = 1;
      }
      if (done < 1) {  // processUndetected hpp:475-519
        const int max_ind = max_index4(projd);
        const int face_id = face_ids[max_ind];
        if (m.side_exposed[face_id]) {
          next = -1;
          for (int i = 0; i < 3; ++i) xpoints_d[(size_t)pid * 3 + i] = xps[max_ind * 3 + i];
          xface_d[pid] = face_id;
          done = 2;
        } else {
          next = other_elem(m, face_id, elm);
        }
      }
      finished = (done == 2);
    }
    elm = next;  // copy_elem_ids hpp:521-524
    ++loops;
    if (finished) break;
    if (loops >= cap) {  // hpp:531-552: the particle keeps its current element id
      atomicAdd(&cnt->not_found, 1);
      break;
    }
  }
  elem_ids[pid] = elm;
}

// ------------------------------------------------------------------ fused push + BCC walk
// DIM 2: ellipticalPush::push + search_mesh_2d.  DIM 3: toroidal push + search_mesh (BCC) with
// finishUnmoved and check_initial_parents.  Particle state is read once; x_tgt, phi, elem_ids
// are written once.
template <int DIM>
__global__ void k_push_walk(int capacity, const unsigned char* __restrict__ mask,
                            const int* __restrict__ slot_elem, const void* __restrict__ recs,
                            const int* __restrict__ class_id, int nelems,
                            const double* __restrict__ x, double* __restrict__ xt,
                            long long stride, const float* __restrict__ pb,
                            float* __restrict__ pphi, double h, double k, double d, double deg,
                            double tol, int* __restrict__ elem_ids, int seeded, int looplimit,
                            Counters* cnt) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  const int e = slot_elem[pid];
  if (e < 0) {  // tail slots of a CSR (see k_search_tpp)
    if (DIM == 3 && !seeded) elem_ids[pid] = -1;
    return;
  }
  if (!mask[pid]) {
    if (DIM == 2 || !seeded) elem_ids[pid] = -1;
    return;
  }
  const int cls = class_id[e];
  const float phi = pphi[pid], b = pb[pid];
  double rad;
  V3 dest;
  int elem = seeded ? elem_ids[pid] : (DIM == 2 ? -1 : e);
  if (DIM == 2) {
    ppm::elliptical_advance(cls, phi, b, h, k, d, deg, dest.x, dest.y, rad);
    dest.z = 0;
    xt[pid] = dest.x;
    xt[stride + pid] = dest.y;
    pphi[pid] = (float)rad;
    if (elem == -1) elem = e;
    if (elem == -nelems) {
      elem_ids[pid] = -1;
      return;
    }
    elem_ids[pid] = bcc_walk<2>(recs, elem, dest, looplimit, cnt);
  } else {
    const V3 orig{x[pid], x[stride + pid], x[2 * stride + pid]};
    ppm::toroidal_advance(cls, phi, b, orig.x, orig.y, h, k, d, deg, dest.x, dest.y, dest.z, rad);
    xt[pid] = dest.x;
    xt[stride + pid] = dest.y;
    xt[2 * stride + pid] = dest.z;
    pphi[pid] = (float)rad;
    bool done = (elem == -1);
    if (norm(sub(dest, orig)) < tol) done = true;
    if (!done && !origin_inside<3>(recs, elem, orig, tol)) {
      atomicAdd(&cnt->not_in_elem, 1);
      elem = -1;
      done = true;
    }
    if (!done) elem = bcc_walk<3>(recs, elem, dest, looplimit, cnt);
    elem_ids[pid] = elem;
  }
}

// ------------------------------------------------------------------ row-tiled fused kernel (SCS)
// Thread = (tile, row): it walks kTileP consecutive particles of ONE row.  Lanes of a wave are
// consecutive rows, so every particle load/store is a coalesced 64-slot run (slot = start + r +
// p*C).  What the flat kernel re-did per particle is hoisted:
//   * per row  : the parent element's class term and the sincos of the toroidal step;
//   * per record: particles of a row start in (or next to) the same element, so the last record
//     read stays in registers together with everything that depends on the element only --
//     3-D: the four face normals cross(c-a, b-a) and 1/vol (barycentric_tet, tpp:51-67);
//     2-D: the three edge vectors (l-k) and the area (barycentric_tri, tpp:28-37).
//   Every value is produced by the same IEEE operations in the same order as the reference
//   (only common sub-expressions are reused), so bcc signs / arg-mins are bit-identical.
//   * the next particle's state is loaded before the current one is processed (software
//     prefetch) so the walk's dependent record loads overlap the streaming loads.
// 2-D decides `all_positive` and `min3` on the NUMERATORS h_i = cross(l-k,pos-k)/2 whenever that
// is provably the same decision as on q_i = h_i/area (division by a positive area is monotone;
// margins of 1e-9 relative are six orders above an ulp); otherwise it falls back to the three
// divisions.  The outcome is identical to the literal code in every case.
template <int DIM>
struct RecCache;
template <>
struct RecCache<2> {
  int id;
  V2 k[3];   // first vertex of edge i
  V2 ev[3];  // l - k of edge i
  double area, lo, hi;  // area ; kEpsilon*area*(1 -/+ 1e-9)
  int nbr[3];
};
template <>
struct RecCache<3> {
  int id;
  V3 a[3];   // M[0], M[1], M[2] (first vertices of faces {0,1},2,3)
  V3 n[4];   // cross(c-a, b-a) of face f
  double inv_vol;  // 1/vol, or NaN-free 0 with ok=false
  bool ok;         // vol > 0
  int nbr[4];
};
__device__ __forceinline__ void build(RecCache<2>& c, const V2 fc[3], int elem) {
  for (int i = 0; i < 3; ++i) {
    c.k[i] = fc[tri_edge_vert(i, 0)];
    c.ev[i] = sub(fc[tri_edge_vert(i, 1)], c.k[i]);
  }
  c.area = tri_area(fc);
  const double T = kEpsilon * c.area;
  c.lo = T * (1.0 - 1e-9);
  c.hi = T * (1.0 + 1e-9);
  c.id = elem;
}
__device__ __forceinline__ void build(RecCache<3>& c, const V3 M[4], double vol, int elem) {
  for (int f = 0; f < 4; ++f) {
    const V3 a = M[tet_face_vert(f, 0)], b = M[tet_face_vert(f, 1)], cc = M[tet_face_vert(f, 2)];
    c.n[f] = cross(sub(cc, a), sub(b, a));
  }
  c.a[0] = M[0];
  c.a[1] = M[1];
  c.a[2] = M[2];
  c.ok = vol > 0;
  c.inv_vol = c.ok ? 1.0 / vol : 0.0;
  c.id = elem;
}
__device__ __forceinline__ void fetch(RecCache<2>& c, const void* __restrict__ recs, int elem) {
  if (c.id == elem) return;
  V2 fc[3];
  load_tri((const pp_tri_rec*)recs, elem, fc, c.nbr);
  build(c, fc, elem);
}
__device__ __forceinline__ void fetch(RecCache<3>& c, const void* __restrict__ recs, int elem) {
  if (c.id == elem) return;
  V3 M[4];
  double vol;
  load_tet((const pp_tet_rec*)recs, elem, M, c.nbr, vol);
  build(c, M, vol, elem);
}
// register-resident select (a dynamic index would push the cached record to scratch)
__device__ __forceinline__ int sel3(const int* a, int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }
__device__ __forceinline__ int sel4(const int* a, int i) {
  return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}
// barycentric_tet on the cached record: bcc[f] = inv_vol * ((pos - a_f) . n_f); -1 when vol <= 0
__device__ __forceinline__ void bcc_cached(const RecCache<3>& c, V3 pos, double bcc[4]) {
  const V3 a0 = c.a[0], a1 = c.a[1], a2 = c.a[2];
  double vals[4];
  vals[0] = dot(sub(pos, a0), c.n[0]);  // faces {0,2,1},{0,1,3} start at M[0]
  vals[1] = dot(sub(pos, a0), c.n[1]);
  vals[2] = dot(sub(pos, a1), c.n[2]);  // {1,2,3}
  vals[3] = dot(sub(pos, a2), c.n[3]);  // {2,0,3}
  for (int i = 0; i < 4; ++i) bcc[i] = c.ok ? c.inv_vol * vals[i] : -1.0;
}
__device__ __forceinline__ bool step_cached(const RecCache<3>& c, V3 pos, int& next) {
  double bcc[4];
  bcc_cached(c, pos, bcc);
  next = sel4(c.nbr, min_index4(bcc));
  return all_positive4(bcc, kEpsilon);
}
__device__ __forceinline__ bool inside_cached(const RecCache<3>& c, V3 orig, double tol) {
  double bcc[4];
  bcc_cached(c, orig, bcc);
  return all_positive4(bcc, tol);
}
// 2-D step: numerators first, exact divisions only when a decision is not provable
__device__ __forceinline__ bool step_cached(const RecCache<2>& c, V3 pos, int& next) {
  const V2 p{pos.x, pos.y};
  double h[3];
  for (int i = 0; i < 3; ++i) h[i] = cross(c.ev[i], sub(p, c.k[i])) / 2.0;
  const double A = c.area;
  // --- provable classification of gtez(h_i / A, kEpsilon)
  bool sure = A > 0;
  bool allpos = true;
  for (int i = 0; i < 3; ++i) {
    const double m = -h[i];
    const bool pos_ = h[i] > 0, in_ = m <= c.lo, out_ = m >= c.hi;
    sure = sure && (pos_ || in_ || out_);
    allpos = allpos && (pos_ || in_);
  }
  // --- provable strict comparisons of the quotients (min3, utils.hpp:88-92)
  auto lt = [&](double x, double y, bool& ok) {
    if (x >= y) return false;  // q_x >= q_y by monotonicity
    const double mx = fmax(fabs(x), fabs(y));
    ok = ok && ((y - x) > 1e-9 * mx) && (mx > A * 1e-250);
    return true;
  };
  int idx = lt(h[0], h[1], sure) ? 0 : 1;
  const double hidx = idx == 0 ? h[0] : h[1];
  idx = lt(hidx, h[2], sure) ? idx : 2;
  if (!sure) {  // literal path
    double q[3];
    for (int i = 0; i < 3; ++i) q[i] = h[i] / A;
    allpos = all_positive3(q, kEpsilon);
    idx = min3(q);
  }
  next = sel3(c.nbr, idx);
  return allpos;
}
template <int DIM>
__device__ __forceinline__ int bcc_walk_cached(RecCache<DIM>& c, const void* __restrict__ recs,
                                               int elem, V3 pos, int looplimit, Counters* cnt, int loops = 0) {
  const int cap = looplimit ? looplimit : kHardLoopCap;
  while (true) {
    fetch(c, recs, elem);
    int next;
    bool done = step_cached(c, pos, next);
    if (!done) {
      if (next == -1) {
        elem = -1;
        done = true;
      } else {
        elem = next;
      }
    }
    ++loops;
    if (done) break;
    if (loops >= cap) {
      elem = -1;
      atomicAdd(&cnt->not_found, 1);
      break;
    }
  }
  return elem;
}

struct PState {
  unsigned char m;
  float phi, b;
  double x, y, z;
  int elem;
  unsigned id;  // record-fed form only: the particle's third member travels through the kernel
};
// NT: non-temporal (streaming) cache policy for the particle streams, so that the 4 MB L2 of an
// XCD keeps the element records instead of particle data it will never see again
template <bool NT, class T>
__device__ __forceinline__ T ld(const T* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT, class T>
__device__ __forceinline__ void stg(T* p, T v) {
  if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}
template <int DIM, bool NT = false>
__device__ __forceinline__ PState load_state(int pid, const unsigned char* __restrict__ mask,
                                             const float* __restrict__ pphi,
                                             const float* __restrict__ pb,
                                             const double* __restrict__ x, long long stride,
                                             const int* __restrict__ elem_ids, bool read_ids) {
  PState s;
  s.m = ld<NT>(mask + pid);
  s.phi = ld<NT>(pphi + pid);
  s.b = ld<NT>(pb + pid);
  s.x = s.y = s.z = 0;
  if (DIM == 3) {
    s.x = ld<NT>(x + pid);
    s.y = ld<NT>(x + stride + pid);
    s.z = ld<NT>(x + 2 * stride + pid);
  }
  s.elem = read_ids ? ld<NT>(elem_ids + pid) : -1;
  return s;
}

// Record-fed form (RECIN).  After a full re-layout with the fused updatePtclPositions the particles are still in
// the 32-B staging records of the move's first pass (pp_ps.hip: k_move_pack; words 0-5 the origin, 6 phi, 7 b) with
// the 4-byte third member beside them in an array in record order (WordTable::side_*) -- the second pass, which would
// copy them into the SoA arrays only for this kernel to read them back, is skipped (pp_ps::lazy_rec).  The members
// the next rebuild packs from the SoA arrays (third member, b; x_tgt and phi are written anyway) are written here.
struct RecIn {
  const char* rec;  // null = SoA input
  const unsigned* side;
  unsigned* id_out;
  float* b_out;
  int rm;      // records row-major inside a chunk (pp_ps::rec_rm): (row r, column p) of chunk c is record
               // pp_rec_row0(chunk_start[c], c, r, chunk_width[c], C) + p; else the record index is the slot
  // 2-D, split records (pp_ps::rec_split): record i's (pad, phi, b, id) is hot[i]; `rec` and `side` are not read
  const uint4* hot = nullptr;
};
template <int DIM = 3>
__device__ __forceinline__ PState load_state_recin(int pid, const unsigned char* __restrict__ mask,
                                                   const char* __restrict__ rec, const unsigned* __restrict__ side,
                                                   long long ri, const uint4* __restrict__ hot = nullptr) {
  PState s;
  s.m = mask[pid];
  if (DIM == 2 && hot) {
    const uint4 hq = hot[ri];
    s.x = s.y = s.z = 0;
    s.phi = __uint_as_float(hq.y);
    s.b = __uint_as_float(hq.z);
    s.id = hq.w;
    s.elem = -1;
    return s;
  }
  const char* rp = rec + ri * 32;
  s.x = s.y = s.z = 0;
  if (DIM == 3) {  // (the 2-D push reads no position)
    const double2 q0 = *(const double2*)rp;
    s.x = q0.x;
    s.y = q0.y;
    s.z = *(const double*)(rp + 16);
  }
  const uint2 q1 = *(const uint2*)(rp + 24);
  s.phi = __uint_as_float(q1.x);
  s.b = __uint_as_float(q1.y);
  s.id = side[ri];
  s.elem = -1;
  return s;
}

// one particle of k_push_walk_rows: push, stores, (3-D) parent check, walk to completion
template <int DIM>
__device__ __forceinline__ void rows_particle(const PState& s, int pid, int e,
                                              const ppm::ClassTerm& ct, RecCache<DIM>& cache,
                                              const void* __restrict__ recs, int nelems,
                                              double* __restrict__ xt, long long stride, float* pphi,
                                              double h, double k, double d, double tol,
                                              double unmoved_sq, int* elem_ids, int seeded,
                                              int looplimit, Counters* cnt, unsigned* id_out = nullptr,
                                              float* b_out = nullptr) {
  if (!s.m) {
    if (DIM == 2 || !seeded) elem_ids[pid] = -1;
    return;
  }
  if (id_out) {  // record-fed form: the members the next rebuild packs from the SoA arrays
    stg<true>(id_out + pid, s.id);
    stg<true>(b_out + pid, s.b);
  }
  double rad;
  V3 dest;
  if constexpr (DIM == 2) {
    int elem = s.elem;
    ppm::elliptical_point(ct, s.phi, s.b, h, k, d, dest.x, dest.y, rad);
    dest.z = 0;
    stg<true>(xt + pid, dest.x);
    stg<true>(xt + stride + pid, dest.y);
    stg<true>(pphi + pid, (float)rad);
    if (elem == -1) elem = e;
    if (elem == -nelems) {
      elem_ids[pid] = -1;
      return;
    }
    elem_ids[pid] = bcc_walk_cached<DIM>(cache, recs, elem, dest, looplimit, cnt);
  } else {
    int elem = seeded ? s.elem : e;
    const V3 orig{s.x, s.y, s.z};
    ppm::toroidal_point(ct, s.phi, s.b, orig.x, orig.y, h, k, d, dest.x, dest.y, dest.z, rad);
    stg<true>(xt + pid, dest.x);
    stg<true>(xt + stride + pid, dest.y);
    stg<true>(xt + 2 * stride + pid, dest.z);
    stg<true>(pphi + pid, (float)rad);
    bool done = (elem == -1);
    // finishUnmoved: norm(dest-orig) < tol  <=>  |dest-orig|^2 < unmoved_sq (sqrt is monotone and
    // correctly rounded; unmoved_sq = min{s : sqrt(s) >= tol} is found on the host)
    const V3 dv = sub(dest, orig);
    if (dot(dv, dv) < unmoved_sq) done = true;
    if (!done) {
      fetch(cache, recs, elem);
      if (!inside_cached(cache, orig, tol)) {
        atomicAdd(&cnt->not_in_elem, 1);
        elem = -1;
        done = true;
      }
    }
    if (!done) elem = bcc_walk_cached<DIM>(cache, recs, elem, dest, looplimit, cnt);
    elem_ids[pid] = elem;
  }
}

template <int DIM, int OCC, bool RECIN = false>
__global__ void __launch_bounds__(256, OCC)
    k_push_walk_rows(const int* __restrict__ ntiles_dev, int C, int TP,
                     const int* __restrict__ tiles, const int* __restrict__ chunk_start,
                     const int* __restrict__ chunk_width, const int* __restrict__ r2e,
                     const unsigned char* __restrict__ mask, const void* __restrict__ recs,
                     const int* __restrict__ class_id, int nelems, const double* __restrict__ x,
                     double* __restrict__ xt, long long stride, const float* __restrict__ pb,
                     float* pphi, double h, double k, double d, double deg, double tol,
                     double unmoved_sq, int* elem_ids, int seeded, int looplimit, Counters* cnt,
                     Counters* cnt_next, RecIn rin = RecIn{}) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *cnt_next = Counters{};  // for the next pp_push_search
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tile = (int)(g / C);
  const int r = (int)(g - (long long)tile * C);
  const bool valid = tile < *ntiles_dev;
  unsigned* const id_out = RECIN ? rin.id_out : nullptr;
  float* const b_out = RECIN ? rin.b_out : nullptr;
  auto load = [&](int pid, long long ri) {
    PState s;
    if constexpr (RECIN) {
      s = load_state_recin<DIM>(pid, mask, rin.rec, rin.side, ri, rin.hot);
      if (seeded) s.elem = elem_ids[pid];
    } else {
      s = load_state<DIM, true>(pid, mask, pphi, pb, x, stride, elem_ids, seeded != 0);
    }
    return s;
  };
  int start = 0, p0 = 0, pend = 0, e = 0;
  long long rb = 0, rs = 1;  // record of column p: rb + p * rs (row-major records: the row's first record, 1)
  if (valid) {
    const int c = tiles[2 * tile];
    p0 = tiles[2 * tile + 1];
    start = chunk_start[c] + r;
    const int cw = chunk_width[c];
    pend = min(p0 + TP, cw);
    e = r2e[c * C + r];
    if (RECIN && rin.rm)
      rb = pp_rec_row0(chunk_start[c], c, r, cw, C);
    else
      rb = start, rs = C;
  }
  const ppm::ClassTerm ct = ppm::class_term((valid && e < nelems) ? class_id[e] : 1, deg, DIM == 3);
  RecCache<DIM> cache;
  cache.id = -1;
  // thin tiles (at most 64/TP live rows): lane l takes (live row l/TP, column l%TP) and the tile
  // is one iteration -- see k_push_walk_rowsq
  {
    const int lane = threadIdx.x & 63;
    const bool alive0 = valid && p0 < pend && mask[start + p0 * C] != 0;
    const unsigned long long live0 = __ballot(alive0);
    const int nlive = __popcll(live0);
    if (nlive * TP <= 64) {
      if ((DIM == 2 || !seeded) && valid && !alive0)  // empty rows: the slots still read -1
        for (int p = p0; p < pend; ++p) elem_ids[start + p * C] = -1;
      const int krow = lane / TP, col = lane - krow * TP;
      unsigned long long m = live0;
      for (int j = 0; j < krow && m; ++j) m &= m - 1;
      const bool have = krow < nlive;
      const int src = have ? __builtin_ctzll(m) : 0;
      const int t_start = __shfl(start, src), t_p0 = __shfl(p0, src), t_pend = __shfl(pend, src);
      const int t_e = __shfl(e, src);
      const long long t_rb = __shfl(rb, src), t_rs = __shfl(rs, src);
      ppm::ClassTerm tct;
      tct.dphi = __shfl(ct.dphi, src);
      tct.st = __shfl(ct.st, src);
      tct.ct = __shfl(ct.ct, src);
      const int p = t_p0 + col;
      if (have && p < t_pend) {
        const int pid = t_start + p * C;
        const PState s = load(pid, t_rb + p * t_rs);
        rows_particle<DIM>(s, pid, t_e, tct, cache, recs, nelems, xt, stride, pphi, h, k, d, tol,
                           unmoved_sq, elem_ids, seeded, looplimit, cnt, id_out, b_out);
      }
      return;
    }
  }
  if (!valid) return;
  // The states of FOUR columns are loaded back to back, the next four before the second column's (dependent) walk.
  // Record-fed, row-major records: the 32-B records of columns (4j .. 4j + 3) of a row are one 128-B line, and the 2-D
  // push reads 8 B of each (+ the side word); asked for one column at a time -- a walk apart -- the line has left the
  // L2 by the next request (round 4, 64-B records: 1.32 GB fetched per step for 10 M particles, every line twice).
  // (p0 is a multiple of the tile width and a row starts on a line: the groups of four are the lines.)
  // (the tet form, which only the lab build's PP_WALK_QUEUE=0 runs, takes two: eight states in flight spill)
  constexpr int G = DIM == 2 ? 4 : 2;
  // the states of columns p .. p + G - 1; record-fed 2-D on row-major records: the four third members as ONE 16-byte
  // load (the side array is in record order, a row's group of four starts on a 16-byte boundary)
  auto loadg = [&](int p, PState* o) {
    if constexpr (RECIN && DIM == 2) {
      if (rin.hot && rs == 1 && p < pend) {  // split records: the four second halves are 64 contiguous bytes
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = PState{};
          if (p + j < pend) {
            const int pid = start + (p + j) * C;
            const uint4 hq = rin.hot[rb + p + j];
            o[j].m = mask[pid];
            o[j].phi = __uint_as_float(hq.y);
            o[j].b = __uint_as_float(hq.z);
            o[j].id = hq.w;
            o[j].elem = seeded ? elem_ids[pid] : -1;
          }
        }
        return;
      }
      if (!rin.hot && rs == 1 && ((rb + p) & 3) == 0 && p < pend) {  // (p >= pend: the prefetch past the row -- no load at all)
        const uint4 ids = *(const uint4*)(rin.side + rb + p);
        const unsigned idv[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = PState{};
          if (p + j < pend) {
            const int pid = start + (p + j) * C;
            const uint2 q1 = *(const uint2*)(rin.rec + (rb + p + j) * 32 + 24);
            o[j].m = mask[pid];
            o[j].phi = __uint_as_float(q1.x);
            o[j].b = __uint_as_float(q1.y);
            o[j].id = idv[j];
            o[j].elem = seeded ? elem_ids[pid] : -1;
          }
        }
        return;
      }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) o[j] = (p + j < pend) ? load(start + (p + j) * C, rb + (p + j) * rs) : PState{};
  };
  PState q[G];
  loadg(p0, q);
  for (int p = p0; p < pend; p += G) {
    const int pid = start + p * C;
    rows_particle<DIM>(q[0], pid, e, ct, cache, recs, nelems, xt, stride, pphi, h, k, d, tol,
                       unmoved_sq, elem_ids, seeded, looplimit, cnt, id_out, b_out);
    PState n[G];
    loadg(p + G, n);
#pragma unroll
    for (int j = 1; j < G; ++j)
      if (p + j < pend)
        rows_particle<DIM>(q[j], pid + j * C, e, ct, cache, recs, nelems, xt, stride, pphi, h, k, d, tol,
                           unmoved_sq, elem_ids, seeded, looplimit, cnt, id_out, b_out);
#pragma unroll
    for (int j = 0; j < G; ++j) q[j] = n[j];
  }
}

// ------------------------------------------------------------------ search_mesh_2d, row-tiled (SCS)
// The stand-alone search_mesh_2d of the drop-in loop (test/pseudoXGCm.cpp:142-156: push by a user lambda, then
// this) with the thread = (tile, row) mapping of k_push_walk_rows: the row's element record is fetched ONCE and
// serves the tile's columns from registers (nine particles in ten end where they started), and a wave's loads
// of a column are 64 consecutive slots.  The flat form (k_search2d, thread = slot) fetches a 64-B record per
// slot -- 64 different lines per wave instruction -- and reads the 4-B slot -> element table on top.
// The inputs of the tile's columns are loaded as one batch.  A particle that does not end in its row's element
// after the first step (or is seeded elsewhere) is a MOVER: its walk needs records nobody has fetched yet, a chain
// of dependent loads that would stall the wave once per column (some lane of 64 moves in every column).  Movers go
// to a per-wave list in LDS and are walked afterwards, one per lane: the chains of a tile's ~50 movers run side by
// side instead of one column after the other (144 -> see DESIGN.md, round 5).
struct Mover2 {
  int pid, elem, loops, pad;
  double x, y;
};
constexpr int kMoverQ = 192;  // list entries per wave; drained when fewer than 64 are free
template <int TPMAX>
__global__ void __launch_bounds__(256, 4)
    k_search2d_rows(const int* __restrict__ ntiles_dev, int C, int TP, const int* __restrict__ tiles,
                    const int* __restrict__ chunk_start, const int* __restrict__ chunk_width,
                    const int* __restrict__ r2e, const unsigned char* __restrict__ mask,
                    const pp_tri_rec* __restrict__ recs, int nelems, const double* __restrict__ xt,
                    long long stride, int* __restrict__ elem_ids, int looplimit, Counters* cnt) {
  __shared__ Mover2 q[4][kMoverQ];
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tile = (int)(g / C);
  const int r = (int)(g - (long long)tile * C);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool valid = tile < *ntiles_dev;
  int start = 0, p0 = 0, pend = 0, e = nelems;
  if (valid) {
    const int c = tiles[2 * tile];
    p0 = tiles[2 * tile + 1];
    start = chunk_start[c] + r;
    pend = min(p0 + TP, chunk_width[c]);
    e = r2e[c * C + r];
  }
  const int cap = looplimit ? looplimit : kHardLoopCap;
  RecCache<2> cache;
  cache.id = -1;
  int nq = 0;  // wave-uniform
  auto drain = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < nq; base += 64) {  // (the row's cached record is given up: re-fetched below)
      const int i = base + lane;
      if (i < nq) {
        const Mover2 m = q[wv][i];
        stg<true>(elem_ids + m.pid, bcc_walk_cached<2>(cache, recs, m.elem, V3{m.x, m.y, 0.0}, looplimit, cnt, m.loops));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    nq = 0;
  };
  // (wave-uniform trip count: the tile's columns are the same in every lane of a wave when C == 64; other chunk
  // heights may mix tiles in a wave, so the bound is the wave's widest)
  int ncol = valid ? pend - p0 : 0;
  for (int o = 32; o > 0; o >>= 1) ncol = max(ncol, __shfl_xor(ncol, o));
  for (int pb = 0; pb < ncol; pb += TPMAX) {
    unsigned char m[TPMAX];
    int seed[TPMAX];
    double x[TPMAX], y[TPMAX];
#pragma unroll
    for (int j = 0; j < TPMAX; ++j) {
      const int pid = start + (p0 + pb + j) * C;
      m[j] = 0;
      seed[j] = -1;
      x[j] = y[j] = 0;
      if (p0 + pb + j < pend) {
        m[j] = ld<true>(mask + pid);
        seed[j] = ld<true>(elem_ids + pid);
        x[j] = ld<true>(xt + pid);
        y[j] = ld<true>(xt + stride + pid);
      }
    }
    // (behind the batch's loads: the record is the end of a chain of four dependent loads)
    if (valid && e < nelems) fetch(cache, recs, e);
#pragma unroll
    for (int j = 0; j < TPMAX; ++j) {
      const bool in = p0 + pb + j < pend;
      const int pid = start + (p0 + pb + j) * C;
      int out = -1, melem = -1, mloops = 0;
      bool mover = false;
      if (in && m[j]) {
        const int elem = seed[j] == -1 ? e : seed[j];  // hpp:1047-1056
        if (elem != -nelems && elem >= 0 && elem < nelems) {
          if (elem == cache.id) {  // the first step of bcc_walk, on the row's record
            int next;
            const bool done = step_cached(cache, V3{x[j], y[j], 0.0}, next);
            if (done) {
              out = elem;
            } else if (next == -1) {
              out = -1;  // exposed side: leaves the domain
            } else if (1 >= cap) {
              atomicAdd(&cnt->not_found, 1);
            } else {
              mover = true;
              melem = next;
              mloops = 1;
            }
          } else {
            mover = true;
            melem = elem;
          }
        }
      }
      if (in && !mover) stg<true>(elem_ids + pid, out);
      const unsigned long long mv = __ballot(mover);
      if (mover) {
        const int at = nq + __popcll(mv & ((1ull << lane) - 1ull));
        q[wv][at] = Mover2{pid, melem, mloops, 0, x[j], y[j]};
      }
      nq += __popcll(mv);
      if (nq > kMoverQ - 64) {  // (rare: more than 128 movers in the tile so far)
        drain();
        if (valid && e < nelems) fetch(cache, recs, e);  // (the drain walked with the row's cache)
      }
    }
  }
  if (nq) drain();
}
// one thread behind the search kernel (stream order): the count, then the stamp, into host-mapped memory.  (A ticket
// counter inside the search kernel -- the last of 48 000 blocks reports -- serialised that many atomics on one
// address: 125 -> 390 us.)
__global__ void k_report_found(const Counters* __restrict__ cnt, FoundPin* pin, int stamp) {
  pin->not_found = cnt->not_found;
  __threadfence_system();
  __hip_atomic_store(&pin->stamp, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------ row-tiled kernel, queued walk
// Same thread = (tile,row) mapping and register-cached seed record as k_push_walk_rows, plus:
//
//  * cooperative record fetch.  Measured on MI355X (tools/ub_gather.hip, profiles/r01_ub_gather):
//    a wave instruction whose 64 lanes read 16 B of 64 DIFFERENT 128-B lines costs ~64 L1 tag
//    cycles, so a lane-private record read (8 x dwordx4) costs 8 lookups per record and the walk
//    was bound by the L1, not by HBM.  Here NP lanes share one record (NP = 8 for the 128-B tet
//    record, 4 for the 64-B tri record): one global_load_lds_dwordx4 touches 64/NP lines, the
//    16-B pieces land in a per-wave LDS staging area (XOR-swizzled through the SOURCE address so
//    the read-back is bank-conflict free) and every lane reads its own record back with
//    ds_read_b128.  One L1 lookup per record instead of eight.
//  * deferred walk.  A particle takes only its FIRST walk step inside the column loop; a particle
//    that crosses into a neighbour is written as a 32-B entry (slot, next element, destination)
//    into the wave's own region of a queue buffer -- no atomics, the wave's entry count goes to
//    wave_cnt[] -- and is finished by k_walk_pending, where wave w walks the entries of region w
//    with all of them stepping together, one cooperative fetch per round.  The column loop keeps
//    no dependent second record load and no 1-of-64-lanes walk; the second pass never gathers
//    x_tgt (a random 8-B gather costs a whole line from HBM).
//  * non-temporal particle streams, so the XCD's L2 keeps element records rather than particle
//    data that is never re-read; thin tiles (<= 64/TP live rows) are transposed to one iteration.
  // 64-slot groups scanned by one wave of k_walk_pending

struct alignas(16) PendEntry {  // deferred-walk queue entry
  int pid, elem;
  double x, y, z;
};
static_assert(sizeof(PendEntry) == 32, "queue entry must be 32 B");

// coop_issue: start the LDS-DMA of the records in `want` (one per lane, -1 = none).
template <int DIM>
__device__ __forceinline__ void coop_issue(const void* __restrict__ recs, int want, double2* st, int lane) {
  constexpr int NP = DIM == 3 ? 8 : 4;  // 16-B pieces per record == lanes per record
  constexpr int RPI = 64 / NP;          // records per wave instruction
  const int sub = lane & (NP - 1);
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int o = RPI * j + lane / NP;  // owner lane of the record this lane helps to fetch
    const int eo = __shfl(want, o);
    const int piece = sub ^ (o & (NP - 1));
    if (eo >= 0)
      __builtin_amdgcn_global_load_lds(
          (const void*)((const char*)recs + (size_t)eo * (NP * 16) + piece * 16),
          (__attribute__((address_space(3))) void*)(st + j * 64), 16, 0, 0);
  }
}
// coop_collect: after the DMA has landed (vmcnt(0)), lanes with take == true read their record
// back from the staging area and rebuild the register cache for element `elem`.
template <int DIM>
__device__ __forceinline__ void coop_collect(RecCache<DIM>& c, bool take, int elem, const double2* st,
                                             int lane) {
  constexpr int NP = DIM == 3 ? 8 : 4;
  const int sub = lane & (NP - 1);
  if (take) {
    const double2* mine = st + lane * NP;
    if constexpr (DIM == 3) {
      double v[12];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double2 t = mine[i ^ sub];
        v[2 * i] = t.x;
        v[2 * i + 1] = t.y;
      }
      const int4 nb = *(const int4*)(mine + (6 ^ sub));
      const double vol = mine[7 ^ sub].x;
      V3 M[4];
      for (int i = 0; i < 4; ++i) M[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
      c.nbr[0] = nb.x;
      c.nbr[1] = nb.y;
      c.nbr[2] = nb.z;
      c.nbr[3] = nb.w;
      build(c, M, vol, elem);
    } else {
      V2 fc[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double2 t = mine[i ^ sub];
        fc[i] = {t.x, t.y};
      }
      const int4 nb = *(const int4*)(mine + (3 ^ sub));
      c.nbr[0] = nb.x;
      c.nbr[1] = nb.y;
      c.nbr[2] = nb.z;
      build(c, fc, elem);
    }
  }
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// synchronous fetch: issue, wait, collect
template <int DIM>
__device__ __forceinline__ void coop_fetch(RecCache<DIM>& c, const void* __restrict__ recs, int want,
                                           double2* st, int lane) {
  coop_issue<DIM>(recs, want, st, lane);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the LDS-DMA pieces have landed
  wave_lds_sync();
  coop_collect<DIM>(c, want >= 0, want, st, lane);
  wave_lds_sync();
}

// ---- record-fed form: the wave's PARTICLE records through the LDS too.  With row-major records (pp_rec_row0) the
// 32-B records of columns (4j .. 4j + 3) of a row are ONE 128-byte line: 8 lanes fetch it (as for an element record)
// and the four columns land in the wave's whole 8 KB staging area -- 8 cache lines per global_load_lds instruction,
// against 64 for each 16-B load a lane-private read needs.  The third member (4 bytes per particle, an array in record
// order beside the records: RecIn::side) comes as ONE 16-byte piece per row into a 1 KB area of its own.  No VGPRs
// hold data in flight: the four columns (p + 4 .. p + 7) are DMA'd while column p + 3 computes.
__device__ __forceinline__ void prec_issue_quad(const char* __restrict__ rec, const unsigned* __restrict__ side,
                                                int my_ri, double2* st, double2* ist, int lane) {
  // One wave-uniform test instead of one exec-mask branch per instruction: the rows of a tile have their four columns
  // or none has (a wave of C = 64 rows is one tile); where a wave holds rows of two tiles (C < 64, the grid's last
  // wave) the rows without re-fetch the line of a row that has them -- into staging nobody reads.
  const unsigned long long have = __ballot(my_ri >= 0);
  if (have == 0ull) return;
  const int any_ri = __shfl(my_ri, (int)__builtin_ctzll(have));
  const int sub = lane & 7;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int o = 8 * j + (lane >> 3);
    int ri = __shfl(my_ri, o);
    ri = ri >= 0 ? ri : any_ri;
    const int piece = sub ^ (o & 7);
    __builtin_amdgcn_global_load_lds((const void*)(rec + (long long)ri * 32 + piece * 16),
                                     (__attribute__((address_space(3))) void*)(st + j * 64), 16, 0, 0);
  }
  const int mine = my_ri >= 0 ? my_ri : any_ri;
  __builtin_amdgcn_global_load_lds((const void*)(side + mine), (__attribute__((address_space(3))) void*)ist, 16, 0, 0);
}
template <int DIM>
__device__ __forceinline__ PState prec_collect_quad(const double2* st, const double2* ist, int lane, int which) {
  const double2* mine = st + lane * 8;
  const int sub = lane & 7, b = 2 * which;
  PState s;
  s.m = 0;
  s.x = s.y = s.z = 0;
  if (DIM == 3) {
    const double2 q0 = mine[(b + 0) ^ sub];
    s.x = q0.x;
    s.y = q0.y;
  }
  const double2 q1 = mine[(b + 1) ^ sub];
  if (DIM == 3) s.z = q1.x;
  const unsigned long long pb = (unsigned long long)__double_as_longlong(q1.y);
  s.phi = __uint_as_float((unsigned)pb);
  s.b = __uint_as_float((unsigned)(pb >> 32));
  s.id = ((const unsigned*)(ist + lane))[which];
  s.elem = -1;
  return s;
}

// per-launch constants of the fused kernel's column arithmetic
struct WalkArgs {
  double* xt;
  long long stride;
  float* pphi;
  double h, k, d, tol, unmoved_sq;
  int* elem_ids;
  int seeded, nelems, cap;
  Counters* cnt;
  int trust;  // pp_ps_set_origin_trust: check_initial_parents is skipped
  unsigned* id_out;  // record-fed form: SoA arrays of the third member and of b (null otherwise)
  float* b_out;
};
// One particle of one column, seed record already in `cache`: push, x_tgt/phi stores,
// check_initial_parents (3-D), first walk step.  Returns true when the particle crossed into
// `elem` and has to be finished by k_walk_pending; otherwise elem_ids[pid] is final.
template <int DIM, bool NT>
__device__ __forceinline__ bool column_math(const WalkArgs& A, const PState& s, bool act, bool live,
                                            int pid, const ppm::ClassTerm& ct,
                                            const RecCache<DIM>& cache, int& elem, V3& dest) {
  if (act && !s.m && (DIM == 2 || !A.seeded)) stg<NT>(A.elem_ids + pid, -1);
  if (!live) return false;
  if (A.id_out) {
    stg<NT>(A.id_out + pid, s.id);
    stg<NT>(A.b_out + pid, s.b);
  }
  double rad;
  bool done = false;
  bool origin_ok = true;
  if constexpr (DIM == 2) {
    ppm::elliptical_point(ct, s.phi, s.b, A.h, A.k, A.d, dest.x, dest.y, rad);
    stg<NT>(A.xt + pid, dest.x);
    stg<NT>(A.xt + A.stride + pid, dest.y);
    if (elem == -1) done = true;  // hpp:1051-1056 (seed == -nelems)
  } else {
    // parent check first (a pure function of the origin): the origin's registers die with the push
    // trust: the caller vouches for the origins (pp_ps_set_origin_trust) -- the test would pass
    origin_ok = (elem == -1) || A.trust || inside_cached(cache, V3{s.x, s.y, s.z}, A.tol);
    ppm::toroidal_point(ct, s.phi, s.b, s.x, s.y, A.h, A.k, A.d, dest.x, dest.y, dest.z, rad);
    stg<NT>(A.xt + pid, dest.x);
    stg<NT>(A.xt + A.stride + pid, dest.y);
    stg<NT>(A.xt + 2 * A.stride + pid, dest.z);
    done = (elem == -1);
    // finishUnmoved: norm(dest-orig) < tol  <=>  |dest-orig|^2 < unmoved_sq (k_push_walk_rows)
    const V3 dv = sub(dest, V3{s.x, s.y, s.z});
    if (dot(dv, dv) < A.unmoved_sq) {
      if (!done && A.trust) atomicAdd(&A.cnt->unmoved, 1);  // (never seen in the pseudoXGCm flows)
      done = true;
    }
  }
  stg<NT>(A.pphi + pid, (float)rad);
  if (!done && !origin_ok) {  // check_initial_parents (tpp:72-145)
    atomicAdd(&A.cnt->not_in_elem, 1);
    elem = -1;
    done = true;
  }
  if (!done) {  // first walk step on the cached record
    int next;
    if (step_cached(cache, dest, next)) {
      done = true;
    } else if (next == -1) {
      elem = -1;
      done = true;
    } else {
      elem = next;
      if (1 >= A.cap) {
        elem = -1;
        atomicAdd(&A.cnt->not_found, 1);
        done = true;
      }
    }
  }
  if (done) stg<NT>(A.elem_ids + pid, elem);
  return !done;
}
// append the crossing particles of this column to the wave's queue region
__device__ __forceinline__ void enqueue(bool need, int pid, int elem, const V3& dest, PendEntry* wq,
                                        int& qn, unsigned long long lt_mask) {
  const unsigned long long bal = __ballot(need);
  if (need) {
    PendEntry en;
    en.pid = pid;
    en.elem = elem;
    en.x = dest.x;
    en.y = dest.y;
    en.z = dest.z;
    wq[qn + __popcll(bal & lt_mask)] = en;
  }
  qn += __popcll(bal);
}
// seed element of a live particle (2-D: -1 = own element, -nelems = skip; 3-D: row element when
// the caller gave no ids)
template <int DIM>
__device__ __forceinline__ int seed_of(const PState& s, int e, int seeded, int nelems) {
  if (DIM == 2) {
    const int el = (s.elem == -1) ? e : s.elem;
    return el == -nelems ? -1 : el;
  }
  return seeded ? s.elem : e;
}

// walks the entries of `nreg` consecutive queue regions (region i holds my_cnt-of-lane-i entries)
template <int DIM>
__device__ __forceinline__ void walk_pending(const PendEntry* __restrict__ regions, long long region_stride,
                                             int nreg, int my_cnt, const void* __restrict__ recs,
                                             int* elem_ids, int cap, Counters* cnt, double2* st, int lane) {
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int reg = 0, off = 0;  // wave-uniform cursor
  int reg_cnt = __shfl(my_cnt, 0);
  bool on = false;
  PendEntry en{};
  int welem = -1, loops = 1;
  RecCache<DIM> cache;
  cache.id = -1;
  while (true) {
    // ---- refill free lanes from the cursor
    unsigned long long free_mask = __ballot(!on);
    while (free_mask != 0ull && reg < nreg) {
      const int avail = reg_cnt - off;
      if (avail <= 0) {
        ++reg;
        off = 0;
        reg_cnt = reg < nreg ? __shfl(my_cnt, reg) : 0;
        continue;
      }
      const int rank = __popcll(free_mask & lt_mask);
      const int take = min(avail, (int)__popcll(free_mask));
      if (!on && rank < take) {
        en = regions[reg * region_stride + off + rank];
        welem = en.elem;
        loops = 1;
        on = true;
      }
      off += take;
      free_mask = __ballot(!on);
    }
    if (__ballot(on) == 0ull) break;
    // ---- one step for every busy lane
    coop_fetch<DIM>(cache, recs, on ? welem : -1, st, lane);
    if (on) {
      int next;
      bool fin = step_cached(cache, V3{en.x, en.y, en.z}, next);
      if (!fin) {
        if (next == -1) {
          welem = -1;
          fin = true;
        } else {
          welem = next;
        }
      }
      ++loops;
      if (!fin && loops >= cap) {
        welem = -1;
        atomicAdd(&cnt->not_found, 1);
        fin = true;
      }
      if (fin) {
        elem_ids[en.pid] = welem;
        on = false;
      }
    }
  }
}
// RECIN: record-fed (else SoA input) -- an unseeded search on row-major 32-B records (RecIn): the element record once
// per tile, four columns (one cache line per row) per fetch; the host materialises the members first when the
// structure's geometry does not allow it (chunk height or tile width not a multiple of four, seeds)
template <int DIM, int OCC, bool NT, bool RECIN = false>
__global__ void __launch_bounds__(256, OCC)
    k_push_walk_rowsq(const int* __restrict__ ntiles_dev, int C, int TP,
                      const int* __restrict__ tiles, const int* __restrict__ chunk_start,
                      const int* __restrict__ chunk_width, const int* __restrict__ r2e,
                      const unsigned char* __restrict__ mask, const void* __restrict__ recs,
                      const int* __restrict__ class_id, int nelems, const double* __restrict__ x,
                      double* xt, long long stride, const float* __restrict__ pb, float* pphi,
                      double h, double k, double d, double deg, double tol, double unmoved_sq,
                      int* elem_ids, int seeded, int looplimit, Counters* cnt, PendEntry* gq,
                      int* wave_cnt, Counters* cnt_next, int trust, RecIn rin = RecIn{}) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *cnt_next = Counters{};  // for the next pp_push_search
  constexpr int NP = DIM == 3 ? 8 : 4;
  extern __shared__ double2 lds_dyn[];
  double2* st = lds_dyn + (size_t)(threadIdx.x >> 6) * 64 * NP;
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int qn = 0;  // wave-uniform number of queue entries written by this wave

  // (blockIdx is dealt round-robin over the 8 XCDs.  Giving every XCD a contiguous range of tiles keeps a
  // chunk's records in one L2 but was measured at +2 % on an even population and -20 % on a skewed one --
  // the over-full element's thin tiles all land on two XCDs -- so the hardware order stays.)
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gwave = g >> 6;
  PendEntry* wq = gq + gwave * 64 * TP;  // this wave's queue region (64*TP entries worst case)
  const int tile = (int)(g / C);
  const int r = (int)(g - (long long)tile * C);
  const bool valid = tile < *ntiles_dev;
  int start = 0, p0 = 0, pend = 0, e = 0;
  int rbase = 0;  // RECIN: record of column p = rbase + p * rstride (row-major records: first record of the row, 1)
  const int rstride = (RECIN && rin.rm) ? 1 : C;
  if (valid) {
    const int c = tiles[2 * tile];
    p0 = tiles[2 * tile + 1];
    start = chunk_start[c] + r;
    const int cw = chunk_width[c];
    pend = min(p0 + TP, cw);
    e = r2e[c * C + r];
    rbase = (RECIN && rin.rm) ? pp_rec_row0(chunk_start[c], c, r, cw, C) : start;
  }
  const ppm::ClassTerm ct = ppm::class_term((valid && e < nelems) ? class_id[e] : 1, deg, DIM == 3);
  const bool read_ids = seeded != 0;  // (2-D without seeds: every seed is -1, the row's element)
  WalkArgs A;
  A.xt = xt;
  A.stride = stride;
  A.pphi = pphi;
  A.h = h;
  A.k = k;
  A.d = d;
  A.tol = tol;
  A.unmoved_sq = unmoved_sq;
  A.elem_ids = elem_ids;
  A.seeded = seeded;
  A.nelems = nelems;
  A.cap = looplimit ? looplimit : kHardLoopCap;
  A.cnt = cnt;
  A.trust = trust;
  A.id_out = RECIN ? rin.id_out : nullptr;
  A.b_out = RECIN ? rin.b_out : nullptr;
  RecCache<DIM> cache;
  cache.id = -1;
  // ---- thin tiles.  Rows fill from column 0, so the rows alive in this tile's first column bound
  // the rows alive in all of it.  When at most 64/TP rows are alive (the tail of a chunk that holds
  // one over-full element: BASELINE's "irregular occupancy") the column loop would run TP
  // iterations with a handful of busy lanes; instead lane l takes (live row l/TP, column l%TP) and
  // the whole tile is ONE iteration.  Loads are strided (one line per lane), which is cheap for
  // the few particles concerned.
  bool thin = false;  // wave-uniform
  {
    const bool alive0 = valid && p0 < pend && mask[start + p0 * C] != 0;
    const unsigned long long live0 = __ballot(alive0);
    const int nlive = __popcll(live0);
    if (nlive * TP <= 64) {
      if ((DIM == 2 || !seeded) && valid && !alive0)  // empty rows: the slots still read -1
        for (int p = p0; p < pend; ++p) stg<NT>(elem_ids + start + p * C, -1);
      const int krow = lane / TP, col = lane - krow * TP;
      unsigned long long m = live0;
      for (int j = 0; j < krow && m; ++j) m &= m - 1;  // drop the krow lowest live rows
      const bool have = krow < nlive;
      const int src = have ? __builtin_ctzll(m) : 0;
      const int t_start = __shfl(start, src), t_p0 = __shfl(p0, src), t_pend = __shfl(pend, src);
      const int t_e = __shfl(e, src);
      const int t_rbase = RECIN ? __shfl(rbase, src) : 0;
      ppm::ClassTerm tct;
      tct.dphi = __shfl(ct.dphi, src);
      tct.st = __shfl(ct.st, src);
      tct.ct = __shfl(ct.ct, src);
      const int p = t_p0 + col;
      const bool act = have && p < t_pend;
      const int pid = t_start + p * C;
      PState s{};
      if (act) {
        if constexpr (RECIN) {
          s = load_state_recin<DIM>(pid, mask, rin.rec, rin.side, (long long)t_rbase + (long long)p * rstride);
          if (read_ids) s.elem = ld<NT>(elem_ids + pid);
        } else {
          s = load_state<DIM, NT>(pid, mask, pphi, pb, x, stride, elem_ids, read_ids);
        }
      }
      const bool live = act && s.m;
      int elem = live ? seed_of<DIM>(s, t_e, seeded, nelems) : -1;
      const int want = elem >= 0 ? elem : -1;
      if (__ballot(want >= 0) != 0ull) coop_fetch<DIM>(cache, recs, want, st, lane);
      V3 dest{0, 0, 0};
      const bool need = column_math<DIM, NT>(A, s, act, live, pid, tct, cache, elem, dest);
      enqueue(need, pid, elem, dest, wq, qn, lt_mask);
      thin = true;
    }
  }
  // Software pipeline over the columns.  At the top of column p everything issued during column
  // p-1 has landed (ONE vmcnt(0) per column): the particle state of p (registers) and the seed
  // records of p (LDS staging, DMA'd from the element id that was read one column earlier).
  // The records are copied to the register cache, then the DMA for column p+1 and the state
  // loads of p+1 are issued and overlap the whole of column p's arithmetic.
  if constexpr (RECIN) {
    static_assert(DIM == 3, "the record-fed queued kernel is the tet form (2-D: k_push_walk_rows)");
    // ---- record-fed form: every live particle starts in its row's element (unseeded search of a structure that
    // was just rebuilt), so the element record is fetched ONCE per tile, up front, and from then on the wave's whole
    // staging area belongs to the particle records: four columns = one cache line per row per fetch
    // (prec_issue_quad).  The columns (p + 4 .. p + 7) are DMA'd while column p + 3 computes.
    double2* const ist = lds_dyn + (size_t)(blockDim.x >> 6) * 64 * NP + (size_t)(threadIdx.x >> 6) * 64;
    if (!thin) {
      const int want0 = (valid && e < nelems && p0 < pend && mask[start + p0 * C] != 0) ? e : -1;
      if (__ballot(want0 >= 0) != 0ull) coop_fetch<DIM>(cache, recs, want0, st, lane);
    }
    auto masks4 = [&](int p) {  // the mask bytes of columns p .. p + 3 (0 beyond the row's end)
      unsigned m = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (p + j < pend) m |= (unsigned)mask[start + (p + j) * C] << (8 * j);
      return m;
    };
    unsigned mq = 0, mq_n = 0;
    {
      const bool first = !thin && p0 < pend;
      prec_issue_quad(rin.rec, rin.side, first ? rbase + p0 : -1, st, ist, lane);
      if (first) mq = masks4(p0);
    }
    for (int i = 0; i < (thin ? 0 : TP); ++i) {  // wave-uniform trip count; ONE column per iteration, one fetch per four
      const int p = p0 + i, pid = start + p * C, which = i & 3;
      const bool act = p < pend;
      if (which == 0) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the four columns' records (and their mask bytes)
        wave_lds_sync();
      }
      PState s = prec_collect_quad<DIM>(st, ist, lane, which);
      s.m = act ? (unsigned char)(mq >> (8 * which)) : 0;
      if (which == 3) {
        wave_lds_sync();  // all four columns are out: the area is free for the next four
        const bool nxt = act && p + 1 < pend && i + 1 < TP;
        prec_issue_quad(rin.rec, rin.side, nxt ? rbase + p + 1 : -1, st, ist, lane);
        if (nxt) mq_n = masks4(p + 1);
      }
      const bool live = act && s.m;
      int elem = live ? e : -1;
      // (the one fetch above relies on rows being left-packed -- a live column implies a live first column, which
      // every rebuild guarantees; should a row ever hold a live slot behind a dead first one, its lane loads the
      // record itself here instead of computing on an unloaded cache: round-4 advisor)
      if (live && cache.id != e) fetch(cache, recs, e);
      V3 dest{0, 0, 0};
      const bool need = column_math<DIM, NT>(A, s, act, live, pid, ct, cache, elem, dest);
      enqueue(need, pid, elem, dest, wq, qn, lt_mask);
      if (which == 3) mq = mq_n;
    }
    if (lane == 0) wave_cnt[gwave] = qn;
    return;
  } else {
  PState cur{};
  int e1 = -1;   // raw elem_ids value of column p+1 (read two columns ahead of its use as a seed)
  int pre = -1;  // element whose record the DMA put into this lane's staging slot for column p
  if (!thin && p0 < pend) {
    cur = load_state<DIM, NT>(start + p0 * C, mask, pphi, pb, x, stride, elem_ids, read_ids);
    if (read_ids && p0 + 1 < pend) e1 = ld<NT>(elem_ids + start + (p0 + 1) * C);
  }
  for (int i = 0; i < (thin ? 0 : TP); ++i) {  // wave-uniform trip count: every lane reaches the wave-level ops
    const int p = p0 + i;
    const int pid = start + p * C;
    const bool act = p < pend;
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): state(p) and the records DMA'd for p
    wave_lds_sync();
    const PState s = cur;
    const bool live = act && s.m;
    // seed element of this column (known before the push; `done` cases just fetch in vain)
    int elem = live ? seed_of<DIM>(s, e, seeded, nelems) : -1;
    const int want = (elem >= 0 && elem != cache.id) ? elem : -1;
    coop_collect<DIM>(cache, want >= 0 && want == pre, want, st, lane);
    wave_lds_sync();
    const int miss = (want >= 0 && want != pre) ? want : -1;  // not prefetched (first column, ...)
    if (__ballot(miss >= 0) != 0ull) coop_fetch<DIM>(cache, recs, miss, st, lane);
    // ---- issue column p+1: record DMA from its (already known) seed, then its state loads
    pre = -1;
    if (act && p + 1 < pend) {
      int seed1 = read_ids ? e1 : e;
      if (DIM == 2 && seed1 == -1) seed1 = e;
      if (seed1 >= 0 && seed1 < nelems && seed1 != cache.id) pre = seed1;
    }
    if (__ballot(pre >= 0) != 0ull) coop_issue<DIM>(recs, pre, st, lane);
    if (act && p + 1 < pend) {
      cur = load_state<DIM, NT>(pid + C, mask, pphi, pb, x, stride, elem_ids, false);
      cur.elem = e1;
      if (read_ids && p + 2 < pend) e1 = ld<NT>(elem_ids + pid + 2 * C);
    }
    V3 dest{0, 0, 0};
    const bool need = column_math<DIM, NT>(A, s, act, live, pid, ct, cache, elem, dest);
    enqueue(need, pid, elem, dest, wq, qn, lt_mask);
  }
  }
  if (lane == 0) wave_cnt[gwave] = qn;
}

// Second pass of the deferred walk.  Walk lengths are long-tailed (a crossing in a tet mesh takes
// 1..8 steps), so a wave that simply stepped 64 entries until the last one finished would idle
// most lanes.  Each wave therefore owns kPendRegions consecutive queue regions and REFILLS lanes
// as they finish: a wave-uniform cursor (region, offset) hands the next unprocessed entries to
// the free lanes, so every round's cooperative fetch + step runs with (nearly) all lanes busy.
constexpr int kPendRegions = 4;
// second pass: wave w owns G consecutive regions
template <int DIM>
__global__ void __launch_bounds__(256, 4)
    k_walk_pending(int nwaves, int TP, int G, const PendEntry* __restrict__ gq,
                   const int* __restrict__ wave_cnt, const void* __restrict__ recs, int* elem_ids,
                   int looplimit, Counters* cnt) {
  constexpr int NP = DIM == 3 ? 8 : 4;
  __shared__ double2 st_all[4 * 64 * NP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long r0 = ((long long)blockIdx.x * 4 + wave) * G;  // first region
  if (r0 >= nwaves) return;
  const int nreg = (int)min((long long)G, nwaves - r0);
  const int my_cnt = lane < nreg ? wave_cnt[r0 + lane] : 0;  // lane i holds the count of region i
  walk_pending<DIM>(gq + r0 * 64 * TP, 64ll * TP, nreg, my_cnt, recs, elem_ids,
                    looplimit ? looplimit : kHardLoopCap, cnt, st_all + wave * 64 * NP, lane);
}

// ------------------------------------------------------------------ Moeller-Trumbore walk on packed records
// search_mesh with requireIntersection (adjacency.tpp:284-361 + the native handler :617-639) follows every
// moving particle's RAY element by element until it meets an exposed face: tens of tets per particle, and per
// tet and face the reference gathers elem2sides -> side2verts -> 3 x coords.  Here a walk step is ONE 128-B
// record (cooperative LDS-DMA fetch, see coop_issue): the four vertices, the neighbours, the volume and
// `mt_code` -- per face the stored side's three vertices as tet-local indices with isFaceFlipped already
// applied (pp_mesh.hip) -- so ray_intersects_triangle sees exactly the operands of the reference, in its
// order; the lane reads the vertices a face names straight out of the staged record (dynamic LDS addresses:
// no register-array indexing).  Ray direction and length are per-particle constants (the reference recomputes
// them per face: same operations, same values).
//  * Persistent waves: a wave draws chunks of 64 * per_lane consecutive slots from ONE device counter and hands
//    them to its lanes AS THEY FALL FREE (wave-uniform cursor): walk lengths differ by the distance to the
//    wall -- a wave that waited for its longest walk would idle most lanes, and waves with fixed shares would
//    finish at different times.
//  * A walking particle tests the THREE faces it did not come in through (the reference skips face_id ==
//    prevExit); only a particle's first element has four candidates, and check_initial_parents on top.  The
//    first steps are therefore batched: fresh lanes wait until `start_batch` of them (or nobody walking) are
//    there, then the wave runs one START round for them; all other rounds are 3-face WALK rounds.  Per round
//    one cooperative fetch and one pass over the faces with (nearly) all participating lanes busy.
// The staged record of a lane is its 128 bytes with the 16-B pieces XOR-swizzled (piece q at slot q ^ (lane & 7),
// see coop_issue): byte b of the record lives at b ^ ((lane & 7) << 4) -- the swizzle touches bits 4-6 only, the
// half-piece bit 3 stays.  A vertex is three doubles at bytes 24 v, 24 v + 8, 24 v + 16: one multiply, two adds,
// three xor-adds per vertex (the generic piece / half arithmetic cost ~7 integer instructions per double).
__device__ __forceinline__ V3 lds_rec_vertex(const double2* mine, int sw, int v) {
  const char* m = (const char*)mine;
  const unsigned swz = (unsigned)sw << 4, b = (unsigned)v * 24u;
  return V3{*(const double*)(m + (b ^ swz)), *(const double*)(m + ((b + 8u) ^ swz)),
            *(const double*)(m + ((b + 16u) ^ swz))};
}
struct MtFace {  // running state of search_findExitFace_intersect_3d over the faces of one element
  int lastExit, bestFace;
  double quality;
  double t_ip;  // ray parameter of the intersection point the reference would hold now (xpoint = orig + dir * t)
  bool upd;     // ... if any face of this element wrote it
  unsigned maybe;  // faces with dproj > -tol: the only ones the no-hit fallback looks at (bit = local face index)
};
// One face of the element staged at `mine`: ray_intersects_triangle (tpp:152-178) on the stored side named by code
// byte `c`, then the reference's bookkeeping (tpp:333-348): the last success wins; without ANY success the face with
// dproj > -tol that is closest in (u, v) -- candidates overwrite the intersection point.
// Round 5: only what the result depends on is evaluated.  A face is a success only if dproj >= tol and a candidate
// only if dproj > -tol, so a back face costs its normal and one dot product -- not u, v, t and the IEEE divide; and
// `closeness` (three min, two max, six abs / subtractions) is consumed only when NO face of the element succeeds
// (then lastExit == -1 for every face, and the candidates are what the loop below makes them) -- mt_face_fallback,
// which nearly never runs.  Every value that IS computed comes from the same operations on the same operands as
// before: parents, exit faces and intersection points stay bit-identical (tests: test_c2_intersection_mode_full_size,
// test_search_mesh_intersection_mode_packed_walk, tools/fuzz_search.py).
struct MtGeom {
  V3 f0, edge1, edge2;
  double dproj;
};
__device__ __forceinline__ MtGeom mt_geom(const double2* mine, int sw, unsigned c, V3 dir) {
  MtGeom g;
  g.f0 = lds_rec_vertex(mine, sw, c & 3);
  const V3 fa = lds_rec_vertex(mine, sw, (c >> 2) & 3), fb = lds_rec_vertex(mine, sw, (c >> 4) & 3);
  g.edge1 = sub(fa, g.f0);
  g.edge2 = sub(fb, g.f0);
  const V3 faceNorm = cross(g.edge2, g.edge1);
  g.dproj = dot(dir, faceNorm);
  return g;
}
__device__ __forceinline__ void mt_uvt(const MtGeom& g, V3 orig, V3 dir, double& u, double& v, double& t) {
  const V3 pvec = cross(dir, g.edge2);
  const double invdet = 1.0 / g.dproj;
  const V3 tvec = sub(orig, g.f0);
  u = invdet * dot(tvec, pvec);
  const V3 qvec = cross(tvec, g.edge1);
  v = invdet * dot(dir, qvec);
  t = invdet * dot(g.edge2, qvec);
}
__device__ __forceinline__ void mt_face(const double2* mine, int sw, unsigned c, int fi, V3 orig, V3 dir, double tol,
                                        MtFace& F) {
  const MtGeom g = mt_geom(mine, sw, c, dir);
  F.maybe |= (g.dproj > -tol) ? (1u << fi) : 0u;
  if (g.dproj >= tol) {
    double u, v, t;
    mt_uvt(g, orig, dir, u, v, t);
    const bool success = (t >= -tol) && (u >= -tol) && (v >= -tol) && (u + v <= 1.0 + 2 * tol);
    // (the reference stores xpoint = orig + dir * t at both places; the point is a function of t alone, so t is
    // kept and the point formed once per element, from the last t written -- the same value)
    F.lastExit = success ? fi : F.lastExit;
    F.t_ip = success ? t : F.t_ip;
    F.upd = F.upd || success;
  }
}
// no face of the element succeeded: the candidates of tpp:340-347 in face order (lastExit == -1 throughout)
__device__ __forceinline__ void mt_face_fallback(const double2* mine, int sw, unsigned c, int fi, V3 orig, V3 dir,
                                                 double tol, MtFace& F) {
  const MtGeom g = mt_geom(mine, sw, c, dir);
  if (!(g.dproj > -tol)) return;
  double u, v, t;
  mt_uvt(g, orig, dir, u, v, t);
  // Kokkos::min(a, b) = (b < a) ? b : a.  In each of the three pairs below the operands are NaN together or not
  // at all (|x| and |1 - x|; |u + v| and |1 - u - v|: inf - inf on one side is inf - inf on the other), and for
  // non-NaN operands the selection equals fmin's -- one v_min_f64 instead of compare + two selects.  The two
  // Kokkos::max stay literal: their operands can be NaN one at a time (u NaN with v finite when dproj == 0).
  const double m1 = __builtin_fmin(fabs(u), fabs(1 - u));
  const double m2 = __builtin_fmin(fabs(v), fabs(1 - v));
  const double m3 = __builtin_fmin(fabs(u + v), fabs(1 - u - v));
  const double mm = PPG_KMAX(m1, m2);
  const double closeness = PPG_KMAX(mm, m3);
  const bool cand = F.quality < 0 || closeness < F.quality;
  F.quality = cand ? closeness : F.quality;
  F.bestFace = cand ? fi : F.bestFace;
  F.t_ip = cand ? t : F.t_ip;
  F.upd = F.upd || cand;
}
__global__ void __launch_bounds__(256, 2)
    k_search_mt3(int capacity, int per_lane, int start_batch, const unsigned char* __restrict__ mask,
                 const int* __restrict__ slot_elem, const void* __restrict__ recs,
                 const int* __restrict__ elem2sides, const double* __restrict__ x,
                 const double* __restrict__ xt, long long stride, int* __restrict__ elem_ids, int seeded,
                 double tol, int* __restrict__ inter_faces, double* __restrict__ inter_points, int looplimit,
                 Counters* cnt, unsigned long long* __restrict__ steps_out) {
  // steps_out[0]: elements visited; steps_out[1]: the chunk counter (both zeroed by the host)
  __shared__ double2 st_all[4 * 64 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sw = lane & 7;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  double2* st = st_all + wave * 64 * 8;
  const double2* mine = st + lane * 8;
  const long long chunk = 64ll * per_lane;
  long long next = 0, wend = 0;  // wave-uniform cursor into the current chunk
  bool more = true;              // the counter has not run past the capacity yet
  const int cap = looplimit ? looplimit : kHardLoopCap;
  bool on = false, chk = false;
  int pid = 0, elem = -1, prev = -1, loops = 0;
  unsigned nsteps = 0;  // elements this lane's particles visited (pp_search_walk_steps)
  V3 orig{0, 0, 0}, dir{0, 0, 0}, ip{0, 0, 0};
  while (true) {
    // ---- refill + START round, batched: the refill (loads, norm, the ray direction's divisions, ~25 registers
    // of per-particle state) and the first element's round (parent check + four faces) cost more than a walk
    // round and would run for one or two lanes nearly every round -- lanes that fall free wait until
    // `start_batch` of them (or nobody walking) are there, then take the next slots together
    unsigned long long free_mask = __ballot(!on);
    if (__popcll(free_mask) >= start_batch || free_mask == ~0ull) {
      while (free_mask != 0ull && (next < wend || more)) {
        if (next >= wend) {  // draw the next chunk
          unsigned long long c0 = 0;
          if (lane == 0) c0 = atomicAdd(steps_out + 1, (unsigned long long)chunk);
          next = (long long)__shfl(c0, 0);
          wend = min((long long)capacity, next + chunk);
          if (next >= capacity) {
            more = false;
            next = wend = 0;
            break;
          }
        }
        const int nfree = __popcll(free_mask);
        const int take = (int)min((long long)nfree, wend - next);
        const int rank = __popcll(free_mask & lt_mask);
        if (!on && rank < take) {
          pid = (int)(next + rank);
          const int e = slot_elem[pid];
          if (e < 0) {  // tail slots of a CSR: an elem_ids the search allocates is -1 there (tpp:506)
            if (!seeded) elem_ids[pid] = -1;
          } else {
            int el = -1;
            bool walk = false;
            V3 o{0, 0, 0}, d{0, 0, 0};
            if (mask[pid]) {
              el = seeded ? elem_ids[pid] : e;  // tpp:504-522
              if (el != -1) {
                o = V3{x[pid], x[stride + pid], x[2 * stride + pid]};
                d = V3{xt[pid], xt[stride + pid], xt[2 * stride + pid]};
                walk = !(norm(sub(d, o)) < tol);  // finishUnmoved tpp:525-533
              }
            }
            if (!walk) {  // initializeIntersection values (tpp:542-547); the parent stays as it is
              inter_points[(size_t)3 * pid] = 0;
              inter_points[(size_t)3 * pid + 1] = 0;
              inter_points[(size_t)3 * pid + 2] = 0;
              inter_faces[pid] = -1;
              if (!seeded || el != -1) elem_ids[pid] = el;
            } else {
              on = true;
              chk = true;
              elem = el;
              prev = -1;
              loops = 0;
              orig = o;
              ip = V3{0, 0, 0};
              const V3 displacement = sub(d, o);
              dir = divs(displacement, norm(displacement));
            }
          }
        }
        next += take;
        free_mask = __ballot(!on);
      }
    }
    const unsigned long long on_mask = __ballot(on);
    if (on_mask == 0ull) break;
    // ---- which kind of round: START (first element: parent check + four faces) for the lanes just filled,
    // else WALK (three faces)
    const bool start_round = __ballot(on && chk) != 0ull;
    const bool act = on && (chk == start_round);
    coop_issue<3>(recs, act ? elem : -1, st, lane);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the LDS-DMA pieces have landed
    wave_lds_sync();
    if (act) {
      ++nsteps;
      const int4 nb = *(const int4*)(mine + (6 ^ sw));
      const double2 tail = mine[7 ^ sw];  // vol | class_id, mt_code
      const unsigned code = (unsigned)(__double_as_longlong(tail.y) >> 32);
      bool fin = false;
      int xf = -1;  // local index of the exposed face the ray leaves through
      MtFace F{-1, -1, -1.0, 0.0, false, 0u};
      if (start_round) {  // (wave-uniform)
        chk = false;
        V3 M[4];
        for (int v = 0; v < 4; ++v) M[v] = lds_rec_vertex(mine, sw, v);
        double bcc[4];
        barycentric_tet(tail.x, M, orig, bcc);
        if (!all_positive4(bcc, tol)) {  // check_initial_parents (tpp:72-145)
          atomicAdd(&cnt->not_in_elem, 1);
          elem = -1;
          fin = true;
        } else {
#pragma unroll
          for (int fi = 0; fi < 4; ++fi) mt_face(mine, sw, code >> (8 * fi), fi, orig, dir, tol, F);
        }
      } else {
        // the face the ray came in through (face_id == prevExit, tpp:320) is the one with `prev` behind it
        const int entry = nb.x == prev ? 0 : nb.y == prev ? 1 : nb.z == prev ? 2 : 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int fi = j + (j >= entry ? 1 : 0);
          mt_face(mine, sw, code >> (8 * fi), fi, orig, dir, tol, F);
        }
      }
      if (!fin && F.lastExit == -1 && F.maybe != 0u) {  // no hit: the closest candidate among the faces just tested
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
          if (F.maybe & (1u << fi)) mt_face_fallback(mine, sw, code >> (8 * fi), fi, orig, dir, tol, F);
      }
      if (!fin) {
        if (F.upd) ip = add(orig, mul(dir, F.t_ip));
        const int lastExit = F.lastExit == -1 ? F.bestFace : F.lastExit;
        fin = lastExit == -1;
        if (!fin) {
          const int nx = lastExit == 0 ? nb.x : lastExit == 1 ? nb.y : lastExit == 2 ? nb.z : nb.w;
          if (nx == -1) {  // check_model_intersection (tpp:372-385): exposed -> done, the parent stays
            fin = true;
            xf = lastExit;
          } else {  // set_new_element (tpp:397-414)
            prev = elem;
            elem = nx;
          }
        }
        ++loops;
        if (!fin && loops >= cap) {
          elem = -1;
          atomicAdd(&cnt->not_found, 1);
          fin = true;
        }
      }
      if (fin) {
        inter_points[(size_t)3 * pid] = ip.x;
        inter_points[(size_t)3 * pid + 1] = ip.y;
        inter_points[(size_t)3 * pid + 2] = ip.z;
        inter_faces[pid] = xf >= 0 ? elem2sides[(size_t)elem * 4 + xf] : -1;
        elem_ids[pid] = elem;
        on = false;
      }
    }
    wave_lds_sync();  // the staging area is free for the next round's DMA
  }
  {
    unsigned long long n = nsteps;
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) atomicAdd(steps_out, n);
  }
}

MeshArrays arrays_of(const pp_mesh* mesh) {
  MeshArrays m;
  m.coords = mesh->d_coords.as<double>();
  m.elem2verts = mesh->d_elem2verts.as<int>();
  m.elem2sides = mesh->d_elem2sides.as<int>();
  m.side2verts = mesh->d_side2verts.as<int>();
  m.side2elems_off = mesh->d_side2elems_off.as<int>();
  m.side2elems = mesh->d_side2elems.as<int>();
  m.side_exposed = mesh->d_side_exposed.as<signed char>();
  m.elem_measure = mesh->d_elem_measure.as<double>();
  m.dual_off = mesh->d_dual_off.as<int>();
  m.dual_elems = mesh->d_dual_elems.as<int>();
  return m;
}

// device-side counters; allocated once and intentionally never freed (library lifetime)
Counters* g_cnt_dev = nullptr;
struct CntRef {
  Counters* get() { return g_cnt_dev; }
} g_cnt;

unsigned long long* g_mt_steps = nullptr;  // elements visited by the last packed Moeller-Trumbore search
pp::DevBuf g_pending_q, g_wave_cnt;  // deferred-walk queue of the fused kernel (grow-only, library lifetime)
Counters* g_last_counters = nullptr;  // the set the last pp_push_search added to (pp_push_search_counters)

// Counters of the deferred-walk kernels: two sets used alternately.  k_walk_pending zeroes the set
// of the NEXT call, so pp_push_search needs no memset launch (8 us per step in rocprof).
// The pair belongs to the STRUCTURE (pp_ps::cnt2): the rebuild that follows a search carries that structure's
// not-found count to the host (pp_ps_last_search_found) even when other structures were searched in between (the
// virtual ranks of one process; round-5 advisor: the pair used to be process-wide).
unsigned long long g_search_serial = 0;  // pp_push_search calls of the process
int pair_counters(pp_ps* ps) {
  if (ps->cnt2) return PP_OK;
  PP_HIP_CHECK(hipMalloc(&ps->cnt2, 2 * sizeof(Counters)));
  PP_HIP_CHECK(hipMemsetAsync(ps->cnt2, 0, 2 * sizeof(Counters), pp::stream()));
  return PP_OK;
}

int reset_counters() {
  if (!g_cnt_dev) PP_HIP_CHECK(hipMalloc((void**)&g_cnt_dev, sizeof(Counters)));
  PP_HIP_CHECK(hipMemsetAsync(g_cnt_dev, 0, sizeof(Counters), pp::stream()));
  return PP_OK;
}
int read_counters(Counters* h) {
  PP_HIP_CHECK(hipMemcpyAsync(h, g_cnt_dev, sizeof(Counters), hipMemcpyDeviceToHost, pp::stream()));
  PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  return PP_OK;
}

FoundPin* g_found_pin = nullptr;
int g_found_stamp = 0;
bool g_found_poll = true;  // (one poll that times out switches back to the stream synchronisation for good)

int member_ok(const pp_ps* ps, int m, int bytes, int ncomp, const char* what) {
  if (int rc = pp::ps_ready(ps)) return rc;  // a member that is only logically zero gets its zeros now
  if (m < 0 || m >= ps->nmembers) {
    pp::set_error(std::string(what) + ": member index out of range");
    return PP_EINVAL;
  }
  const int s = ps->member_map[m];
  if (ps->member_bytes[s] != bytes || ps->member_ncomp[s] < ncomp) {
    pp::set_error(std::string(what) + ": member has the wrong type for this operator");
    return PP_EINVAL;
  }
  return PP_OK;
}
#define PP_MEMBER(ps, m, T) ((T*)(ps)->data[(ps)->member_map[m]].p)

}  // namespace

namespace pp {
const int* search_not_found_dev(const pp_ps* ps) { return ps->last_nf_dev; }
unsigned long long search_serial() { return g_search_serial; }
void search_counters_released(const void* cnt2) {  // a structure is destroyed: pp_push_search_counters must not read its sets
  const Counters* c = (const Counters*)cnt2;
  if (c && g_last_counters >= c && g_last_counters < c + 2) g_last_counters = nullptr;
}
}  // namespace pp

extern "C" {

int pp_search_mesh_2d(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_pid,
                      int* elem_ids_dev, int looplimit, int* found) {
  pp::Range rg_("search_mesh_2d");
  (void)m_x;
  (void)m_pid;
  PP_REQUIRE(mesh && ps && elem_ids_dev, "pp_search_mesh_2d: null argument");
  PP_REQUIRE(mesh->dim == 2, "pp_search_mesh_2d: needs a triangle mesh");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_search_mesh_2d: structure/mesh element mismatch");
  int rc;
  if ((rc = member_ok(ps, m_xtgt, 8, 2, "pp_search_mesh_2d x_tgt"))) return rc;
  if (found) *found = 1;
  if (ps->num_ptcls == 0 || ps->capacity == 0) return PP_OK;
  if ((rc = reset_counters())) return rc;
  if (ps->kind == PP_SCS && ps->ntiles_max > 0 && ps->tile_p <= 8) {
    if (found && g_found_poll && !g_found_pin) {
      if (hipHostMalloc((void**)&g_found_pin, sizeof(FoundPin), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        g_found_pin = nullptr;
        g_found_poll = false;
      } else {
        memset(g_found_pin, 0, sizeof(FoundPin));
      }
    }
    const bool poll = found && g_found_poll && g_found_pin;
    const int stamp = poll ? (g_found_stamp = g_found_stamp % 1000000 + 1) : 0;
    k_search2d_rows<8><<<grid_for((size_t)ps->ntiles_max * ps->C), kBlock, 0, pp::stream()>>>(
        ps->d_ntiles.as<int>(), ps->C, ps->tile_p, ps->d_tiles.as<int>(), ps->d_chunk_start.as<int>(),
        ps->d_chunk_width.as<int>(), ps->d_row_to_element.as<int>(), ps->d_mask.as<unsigned char>(),
        mesh->d_records.as<pp_tri_rec>(), mesh->nelems, PP_MEMBER(ps, m_xtgt, double), ps->stride, elem_ids_dev,
        looplimit, g_cnt.get());
    if (poll) k_report_found<<<1, 1, 0, pp::stream()>>>(g_cnt.get(), g_found_pin, stamp);
    PP_LAUNCH_CHECK();
    if (poll) {
      const auto t0 = std::chrono::steady_clock::now();
      volatile int* flag = &g_found_pin->stamp;
      long spins = 0;
      bool seen = true;
      while (*flag != stamp) {
        if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) {
          PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
          if (*flag != stamp) {  // kernel stores do not reach this memory mid-stream here: the stream's way from now on
            g_found_poll = false;
            seen = false;
          }
          break;
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (seen) {
        const int nf = g_found_pin->not_found;
        *found = (nf == 0);
        if (nf)
          fprintf(stderr, "ERROR: loop limit %d exceeded. %d particles were not found. Deleting them...\n", looplimit, nf);
        return PP_OK;
      }
    }
  } else  // CSR: thread = slot
    k_search2d<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
        ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps),
        mesh->d_records.as<pp_tri_rec>(), mesh->nelems, PP_MEMBER(ps, m_xtgt, double), ps->stride,
        elem_ids_dev, looplimit, g_cnt.get());
  PP_LAUNCH_CHECK();
  if (found) {
    Counters h;
    if ((rc = read_counters(&h))) return rc;
    *found = (h.not_found == 0);
    if (h.not_found)
      fprintf(stderr, "ERROR: loop limit %d exceeded. %d particles were not found. Deleting them...\n",
              looplimit, h.not_found);
  }
  return PP_OK;
}

int pp_search_mesh(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_pid,
                   int* elem_ids_dev, int elem_ids_seeded, int requireIntersection,
                   int* inter_faces_dev, double* inter_points_dev, int looplimit, int* found,
                   int* num_not_in_elem) {
  pp::Range rg_("search_mesh");
  (void)m_pid;
  PP_REQUIRE(mesh && ps && elem_ids_dev, "pp_search_mesh: null argument");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_search_mesh: structure/mesh element mismatch");
  PP_REQUIRE(!requireIntersection || (inter_faces_dev && inter_points_dev),
             "pp_search_mesh: intersection mode needs inter_faces/inter_points");
  int rc;
  if ((rc = member_ok(ps, m_x, 8, 3, "pp_search_mesh x"))) return rc;
  if ((rc = member_ok(ps, m_xtgt, 8, 3, "pp_search_mesh x_tgt"))) return rc;
  if (found) *found = 1;
  if (num_not_in_elem) *num_not_in_elem = 0;
  if (ps->capacity == 0) return PP_OK;
  if (ps->num_ptcls == 0) {
    if (!elem_ids_seeded)
      PP_HIP_CHECK(hipMemsetAsync(elem_ids_dev, 0xff, sizeof(int) * (size_t)ps->capacity, pp::stream()));
    return PP_OK;
  }
  if ((rc = reset_counters())) return rc;
  const MeshArrays m = arrays_of(mesh);
  const unsigned grid = grid_for(ps->capacity);
  hipStream_t st = pp::stream();
#define PP_TPP_ARGS                                                                              \
  ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), mesh->d_records.p, m,  \
      PP_MEMBER(ps, m_x, double), PP_MEMBER(ps, m_xtgt, double), ps->stride, elem_ids_dev,       \
      elem_ids_seeded, mesh->tol, inter_faces_dev, inter_points_dev, looplimit, g_cnt.get()
  // intersection mode: the walk on packed records (tets: k_search_mt3 with lane refill; triangles: the packed
  // branch of k_search_tpp); a mesh whose records cannot be packed (pp_mesh::mt_packed_ok) keeps the form on the
  // Omega_h-style arrays (lab build, PP_MT_PACKED=0: every mesh)
  static const bool mt_packed_off = PP_LAB_ENV("PP_MT_PACKED") != nullptr && atoi(PP_LAB_ENV("PP_MT_PACKED")) == 0;
  if (mesh->dim == 2) {
    if (requireIntersection)
      k_search_tpp<2, true><<<grid, kBlock, 0, st>>>(PP_TPP_ARGS, (mesh->mt_packed_ok && !mt_packed_off) ? 1 : 0);
    else
      k_search_tpp<2, false><<<grid, kBlock, 0, st>>>(PP_TPP_ARGS);
  } else {
    if (requireIntersection && mesh->mt_packed_ok && !mt_packed_off) {
      const int per_lane = 2;  // (chunks of 128 slots; 1 / 2 / 4 / 8: 24.0 / 15.8 / 16.1 / 17.1 ms for c2mt)
      const int start_batch = 12;
      // persistent blocks: as many as are resident at once (4 per CU: 112 VGPRs, 32 KB of LDS each), fewer when the
      // structure is small; the chunks of slots are drawn from a device counter
      static int resident_blocks = 0;
      if (!resident_blocks) {
        int dev = 0, cus = 0;
        PP_HIP_CHECK(hipGetDevice(&dev));
        PP_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident_blocks = std::max(cus, 1) * 4;
      }
      const size_t chunks = ((size_t)ps->capacity + 64 * (size_t)per_lane - 1) / (64 * (size_t)per_lane);
      const unsigned mt_grid = (unsigned)std::min<size_t>((chunks + 3) / 4, (size_t)resident_blocks);
      if (!g_mt_steps) PP_HIP_CHECK(hipMalloc((void**)&g_mt_steps, 2 * sizeof(unsigned long long)));
      PP_HIP_CHECK(hipMemsetAsync(g_mt_steps, 0, 2 * sizeof(unsigned long long), st));
      k_search_mt3<<<mt_grid, kBlock, 0, st>>>(
          ps->capacity, per_lane, start_batch, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), mesh->d_records.p,
          mesh->d_elem2sides.as<int>(), PP_MEMBER(ps, m_x, double), PP_MEMBER(ps, m_xtgt, double), ps->stride,
          elem_ids_dev, elem_ids_seeded, mesh->tol, inter_faces_dev, inter_points_dev, looplimit, g_cnt.get(),
          g_mt_steps);
    } else if (requireIntersection)
      k_search_tpp<3, true><<<grid, kBlock, 0, st>>>(PP_TPP_ARGS);
    else
      k_search_tpp<3, false><<<grid, kBlock, 0, st>>>(PP_TPP_ARGS);
  }
#undef PP_TPP_ARGS
  PP_LAUNCH_CHECK();
  if (found || num_not_in_elem) {
    Counters h;
    if ((rc = read_counters(&h))) return rc;
    if (found) *found = (h.not_found == 0);
    if (num_not_in_elem) *num_not_in_elem = h.not_in_elem;
    if (h.not_in_elem)
      fprintf(stderr,
              "[WARNING] %d particles are not located in their starting elements. Deleting them...\n",
              h.not_in_elem);
    if (h.not_found)
      fprintf(stderr, "ERROR: loop limit %d exceeded. %d particles were not found. Deleting them...\n",
              looplimit, h.not_found);
  }
  return PP_OK;
}

int pp_search_mesh_legacy3d(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_pid,
                            int* elem_ids_dev, int elem_ids_seeded, double* xpoints_dev,
                            int* xface_dev, int looplimit, int* found) {
  (void)m_pid;
  PP_REQUIRE(mesh && ps && elem_ids_dev && xpoints_dev && xface_dev,
             "pp_search_mesh_legacy3d: null argument");
  PP_REQUIRE(mesh->dim == 3, "pp_search_mesh_legacy3d: needs a tet mesh");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_search_mesh_legacy3d: structure/mesh mismatch");
  int rc;
  if ((rc = member_ok(ps, m_x, 8, 3, "pp_search_mesh_legacy3d x"))) return rc;
  if ((rc = member_ok(ps, m_xtgt, 8, 3, "pp_search_mesh_legacy3d x_tgt"))) return rc;
  if (found) *found = 1;
  if (ps->num_ptcls == 0 || ps->capacity == 0) return PP_OK;
  if ((rc = reset_counters())) return rc;
  k_search_legacy3d<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), arrays_of(mesh),
      PP_MEMBER(ps, m_x, double), PP_MEMBER(ps, m_xtgt, double), ps->stride, elem_ids_dev,
      elem_ids_seeded, xpoints_dev, xface_dev, looplimit, g_cnt.get());
  PP_LAUNCH_CHECK();
  if (found) {
    Counters h;
    if ((rc = read_counters(&h))) return rc;
    *found = h.aborted ? -2 : (h.not_found == 0);
    if (h.aborted)
      fprintf(stderr, "Warning: %d particles not in their element at loops=0 (the reference aborts)\n",
              h.aborted);
    if (h.not_found) fprintf(stderr, "ERROR:loop limit %d exceeded\n", looplimit);
  }
  return PP_OK;
}

// ---- stepwise walk (see k_trace_begin)
int pp_trace_begin(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int* elem_ids_dev,
                   int elem_ids_seeded, int requireIntersection, int* inter_faces_dev,
                   double* inter_points_dev, int* ptcl_done_dev, int* last_exit_dev,
                   int* num_not_in_elem) {
  PP_REQUIRE(mesh && ps && elem_ids_dev && ptcl_done_dev && last_exit_dev, "pp_trace_begin: null argument");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_trace_begin: structure/mesh element mismatch");
  PP_REQUIRE(!requireIntersection || (inter_faces_dev && inter_points_dev),
             "pp_trace_begin: intersection mode needs inter_faces/inter_points");
  int rc;
  if ((rc = member_ok(ps, m_x, 8, 3, "pp_trace_begin x"))) return rc;
  if ((rc = member_ok(ps, m_xtgt, 8, 3, "pp_trace_begin x_tgt"))) return rc;
  if (num_not_in_elem) *num_not_in_elem = 0;
  if (ps->capacity == 0) return PP_OK;
  if ((rc = reset_counters())) return rc;
  const unsigned grid = grid_for(ps->capacity);
#define PP_TB_ARGS                                                                                \
  ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), mesh->d_records.p,      \
      PP_MEMBER(ps, m_x, double), PP_MEMBER(ps, m_xtgt, double), ps->stride, elem_ids_dev,        \
      elem_ids_seeded, mesh->tol, requireIntersection, inter_faces_dev, inter_points_dev,         \
      ptcl_done_dev, last_exit_dev, g_cnt.get()
  if (mesh->dim == 2)
    k_trace_begin<2><<<grid, kBlock, 0, pp::stream()>>>(PP_TB_ARGS);
  else
    k_trace_begin<3><<<grid, kBlock, 0, pp::stream()>>>(PP_TB_ARGS);
#undef PP_TB_ARGS
  PP_LAUNCH_CHECK();
  if (num_not_in_elem) {
    Counters h;
    if ((rc = read_counters(&h))) return rc;
    *num_not_in_elem = h.not_in_elem;
    if (h.not_in_elem)
      fprintf(stderr,
              "[WARNING] %d particles are not located in their starting elements. Deleting them...\n",
              h.not_in_elem);
  }
  return PP_OK;
}

int pp_trace_find_exit_face(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt,
                            const int* elem_ids_dev, int* ptcl_done_dev, int* last_exit_dev,
                            double* inter_points_dev, int use_bcc) {
  PP_REQUIRE(mesh && ps && elem_ids_dev && ptcl_done_dev && last_exit_dev,
             "pp_trace_find_exit_face: null argument");
  PP_REQUIRE(use_bcc || inter_points_dev, "pp_trace_find_exit_face: intersection mode needs inter_points");
  int rc;
  if ((rc = member_ok(ps, m_x, 8, 3, "pp_trace_find_exit_face x"))) return rc;
  if ((rc = member_ok(ps, m_xtgt, 8, 3, "pp_trace_find_exit_face x_tgt"))) return rc;
  if (ps->capacity == 0) return PP_OK;
  const unsigned grid = grid_for(ps->capacity);
#define PP_TF_ARGS                                                                               \
  ps->capacity, ps->d_mask.as<unsigned char>(), mesh->d_records.p, arrays_of(mesh),              \
      PP_MEMBER(ps, m_x, double), PP_MEMBER(ps, m_xtgt, double), ps->stride, elem_ids_dev,      \
      ptcl_done_dev, last_exit_dev, inter_points_dev, use_bcc, mesh->tol
  if (mesh->dim == 2)
    k_trace_find_exit<2><<<grid, kBlock, 0, pp::stream()>>>(PP_TF_ARGS);
  else
    k_trace_find_exit<3><<<grid, kBlock, 0, pp::stream()>>>(PP_TF_ARGS);
#undef PP_TF_ARGS
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_trace_check_model_intersection(const pp_mesh* mesh, pp_ps* ps, int* elem_ids_dev,
                                      int* ptcl_done_dev, const int* last_exit_dev,
                                      int requireIntersection, int* inter_faces_dev) {
  PP_REQUIRE(mesh && ps && elem_ids_dev && ptcl_done_dev && last_exit_dev,
             "pp_trace_check_model_intersection: null argument");
  PP_REQUIRE(!requireIntersection || inter_faces_dev,
             "pp_trace_check_model_intersection: intersection mode needs inter_faces");
  if (ps->capacity == 0) return PP_OK;
  k_trace_check_model<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), mesh->d_side_exposed.as<signed char>(),
      elem_ids_dev, ptcl_done_dev, last_exit_dev, requireIntersection, inter_faces_dev);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

static int count_kernel_result(int* out) {
  PP_HIP_CHECK(hipMemcpyAsync(out, &g_cnt_dev->pending, sizeof(int), hipMemcpyDeviceToHost, pp::stream()));
  PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  return PP_OK;
}

int pp_trace_set_new_element(const pp_mesh* mesh, pp_ps* ps, int* elem_ids_dev,
                             const int* ptcl_done_dev, const int* last_exit_dev, int* num_unfinished) {
  PP_REQUIRE(mesh && ps && elem_ids_dev && ptcl_done_dev && last_exit_dev && num_unfinished,
             "pp_trace_set_new_element: null argument");
  *num_unfinished = 0;
  if (ps->capacity == 0) return PP_OK;
  int rc;
  if ((rc = reset_counters())) return rc;
  k_trace_set_new_element<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), arrays_of(mesh), elem_ids_dev, ptcl_done_dev,
      last_exit_dev, &g_cnt_dev->pending);
  PP_LAUNCH_CHECK();
  return count_kernel_result(num_unfinished);
}

int pp_trace_not_found(pp_ps* ps, int* elem_ids_dev, const int* ptcl_done_dev, int* num_not_found) {
  PP_REQUIRE(ps && elem_ids_dev && ptcl_done_dev && num_not_found, "pp_trace_not_found: null argument");
  *num_not_found = 0;
  if (ps->capacity == 0) return PP_OK;
  int rc;
  if ((rc = reset_counters())) return rc;
  k_trace_not_found<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), elem_ids_dev, ptcl_done_dev, &g_cnt_dev->pending);
  PP_LAUNCH_CHECK();
  return count_kernel_result(num_not_found);
}

int pp_search_mesh_3d(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_pid,
                      int* elem_ids_dev, int elem_ids_seeded, double* xpoints_dev, int* xface_dev,
                      int looplimit, int* found) {
  (void)m_pid;
  PP_REQUIRE(mesh && ps && elem_ids_dev && xpoints_dev && xface_dev,
             "pp_search_mesh_3d: null argument");
  PP_REQUIRE(mesh->dim == 3, "pp_search_mesh_3d: needs a tet mesh");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_search_mesh_3d: structure/mesh mismatch");
  int rc;
  if ((rc = member_ok(ps, m_x, 8, 3, "pp_search_mesh_3d x"))) return rc;
  if ((rc = member_ok(ps, m_xtgt, 8, 3, "pp_search_mesh_3d x_tgt"))) return rc;
  if (found) *found = 1;
  if (ps->capacity == 0) return PP_OK;
  if ((rc = reset_counters())) return rc;
  k_search_mesh3d<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), arrays_of(mesh),
      PP_MEMBER(ps, m_x, double), PP_MEMBER(ps, m_xtgt, double), ps->stride, elem_ids_dev,
      elem_ids_seeded, xpoints_dev, xface_dev, looplimit, g_cnt.get());
  PP_LAUNCH_CHECK();
  if (found) {
    Counters h;
    if ((rc = read_counters(&h))) return rc;
    *found = h.aborted ? -2 : (h.not_found == 0);
    if (h.aborted)
      fprintf(stderr, "Search1: %d particles not in their parent element (the reference aborts)\n",
              h.aborted);
    if (h.not_found) fprintf(stderr, "ERROR:loop limit %d exceeded\n", looplimit);
  }
  return PP_OK;
}

int pp_push_search(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_b, int m_phi,
                   double h, double k, double d, double deg, int* elem_ids_dev,
                   int elem_ids_seeded, int looplimit, int* found) {
  pp::Range rg_("pp_push_search");
  PP_REQUIRE(mesh && ps && elem_ids_dev, "pp_push_search: null argument");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_push_search: structure/mesh element mismatch");
  int rc;
  // x_tgt left "logically zero" by the in-place rebuild's fused updatePtclPositions: the 3-D push
  // overwrites all three components of every live particle, so the zeros are never written.  The
  // flag is dropped only for a call that will run (arguments of the right shape, a 3-component
  // target); anything else goes through member_ok -> ps_ready, which writes the zeros, and a call
  // that fails validation leaves the structure as it was.
  const auto shape_ok = [&](int m, int bytes, int ncomp) {
    return m >= 0 && m < ps->nmembers && ps->member_bytes[ps->member_map[m]] == bytes &&
           ps->member_ncomp[ps->member_map[m]] == ncomp;
  };
  if (mesh->dim == 3 && m_x != m_xtgt && shape_ok(m_x, 8, 3) && shape_ok(m_xtgt, 8, 3) && shape_ok(m_b, 4, 1) &&
      shape_ok(m_phi, 4, 1) && ps->zero_pending == ps->member_map[m_xtgt] && ps->zero_pending != ps->member_map[m_x] &&
      ps->capacity > 0 && ps->num_ptcls > 0)
    ps->zero_pending = -1;
  // Record-fed push: the last full re-layout left the particles in its staging records (pp_ps::lazy_rec)
  // and this call can read them there -- the pass that would copy them to the SoA arrays first is skipped.
  // (lab build, PP_WALK_QUEUE=0/1: the column-loop kernel / the queued kernel whatever the dimension)
  const int wq_env = PP_LAB_ENV("PP_WALK_QUEUE") ? atoi(PP_LAB_ENV("PP_WALK_QUEUE")) : -1;  // per call
  const int wq = wq_env >= 0 ? wq_env : (mesh->dim == 3 ? 1 : 0);
  // (the queued tet kernel takes the records four columns = one cache line at a time: row-major records, no seeds,
  // chunk height and tile width multiples of four; anything else runs the deferred pass first and reads the arrays)
  const bool quads_ok = ps->rec_rm && !elem_ids_seeded && (ps->tile_p & 3) == 0 && (ps->C & 3) == 0;
  const bool recin = ((mesh->dim == 3 && wq > 0 && quads_ok) || (mesh->dim == 2 && wq == 0)) &&
                     ps->capacity > 0 && ps->num_ptcls > 0 && ps->ntiles_max > 0 &&
                     pp::lazy_push_ok(ps, m_x, m_xtgt, m_b, m_phi) &&
                     !(mesh->dim == 3 && ps->rec_split);  // (split records hold no third component: 2-D only)
  bool z_stays_zero = false;
  if (recin) {
    // (a 2-D push writes two components of x_tgt: when x_tgt is only logically zero the third one STAYS logically
    // zero -- pp_ps::zero_z_pending, set behind the launch below: no fill of the member, no store per particle, and
    // the next committing re-layout packs a zero instead of reading the plane)
    if (mesh->dim == 2 && ps->zero_pending == ps->member_map[m_xtgt]) {
      z_stays_zero = true;
      ps->zero_pending = -1;
    }
    if ((rc = pp::ps_zeros(ps))) return rc;
  } else {
    if ((rc = member_ok(ps, m_x, 8, 3, "pp_push_search x"))) return rc;
    if ((rc = member_ok(ps, m_xtgt, 8, 3, "pp_push_search x_tgt"))) return rc;
    if ((rc = member_ok(ps, m_b, 4, 1, "pp_push_search b"))) return rc;
    if ((rc = member_ok(ps, m_phi, 4, 1, "pp_push_search phi"))) return rc;
  }
  if (found) *found = 1;
  if (ps->capacity == 0) return PP_OK;
  if (ps->num_ptcls == 0) {
    if (!elem_ids_seeded && mesh->dim == 3)
      PP_HIP_CHECK(hipMemsetAsync(elem_ids_dev, 0xff, sizeof(int) * (size_t)ps->capacity, pp::stream()));
    return PP_OK;
  }
  const unsigned grid = grid_for(ps->capacity);
  hipStream_t st = pp::stream();
  Counters* used = nullptr;  // the counter set this call's kernels add to
  if (ps->kind == PP_SCS) {
    const unsigned rgrid = grid_for((size_t)ps->ntiles_max * ps->C);
#define PP_ROWS_ARGS                                                                             \
  ps->d_ntiles.as<int>(), ps->C, ps->tile_p, ps->d_tiles.as<int>(), ps->d_chunk_start.as<int>(), \
      ps->d_chunk_width.as<int>(), ps->d_row_to_element.as<int>(),                               \
      ps->d_mask.as<unsigned char>(), mesh->d_records.p, mesh->d_class_id.as<int>(),             \
      mesh->nelems, PP_MEMBER(ps, m_x, double), PP_MEMBER(ps, m_xtgt, double), ps->stride,       \
      PP_MEMBER(ps, m_b, float), PP_MEMBER(ps, m_phi, float), h, k, d, deg, mesh->tol,           \
      mesh->unmoved_sq, elem_ids_dev, elem_ids_seeded, looplimit, used, (Counters*)ps->cnt2 + (ps->cnt2_cur ^ 1)
#define PP_ROWSQ_ARGS                                                                            \
  ps->d_ntiles.as<int>(), ps->C, ps->tile_p, ps->d_tiles.as<int>(), ps->d_chunk_start.as<int>(), \
      ps->d_chunk_width.as<int>(), ps->d_row_to_element.as<int>(),                               \
      ps->d_mask.as<unsigned char>(), mesh->d_records.p, mesh->d_class_id.as<int>(),             \
      mesh->nelems, PP_MEMBER(ps, m_x, double), PP_MEMBER(ps, m_xtgt, double), ps->stride,       \
      PP_MEMBER(ps, m_b, float), PP_MEMBER(ps, m_phi, float), h, k, d, deg, mesh->tol,           \
      mesh->unmoved_sq, elem_ids_dev, elem_ids_seeded, looplimit, used,                          \
      g_pending_q.as<PendEntry>(), g_wave_cnt.as<int>(), (Counters*)ps->cnt2 + (ps->cnt2_cur ^ 1), trust
    // (kernels are built for 4 waves per SIMD: measured fastest on MI355X -- 3-D needs 126 VGPRs, a
    // 96-register build for 5 waves spills and runs 2x slower, 3 waves hide less latency)
    // Two row-tiled variants (measured on MI355X, profiles/r01_c_*):
    //   k_push_walk_rows   walks every particle to completion inside the column loop;
    //   k_push_walk_rowsq  one step in the loop, cooperative LDS-DMA record fetch pipelined one
    //                      column ahead, crossing particles finished by k_walk_pending.
    // 3-D (128-B records, 8% of the particles cross per step, walks of 1..8 tets) is 8-30%
    // faster with the deferred walk; 2-D (64-B records, cheap steps) is faster in one kernel.
    // PP_WALK_QUEUE=0/1 forces one or the other (A/B knob).
    const int trust = ps->trust_origins ? 1 : 0;
    // (two counter sets used alternately: the kernels of a call clear the set of the next one -- no fill launch)
    if ((rc = pair_counters(ps))) return rc;
    used = (Counters*)ps->cnt2 + ps->cnt2_cur;
    if (rgrid > 0 && wq > 0) {
      const size_t lds = (size_t)(kBlock / 64) * 64 * (mesh->dim == 3 ? 8 : 4) * sizeof(double2);
      const size_t nwaves = (size_t)rgrid * (kBlock / 64);
      PP_HIP_CHECK(g_pending_q.reserve(nwaves * 64 * ps->tile_p * sizeof(PendEntry)));
      PP_HIP_CHECK(g_wave_cnt.reserve(nwaves * sizeof(int)));
      if (mesh->dim == 2) {
        k_push_walk_rowsq<2, 4, true><<<rgrid, kBlock, lds, st>>>(PP_ROWSQ_ARGS);
      } else if (recin) {
        const RecIn rin{ps->s_aos_live.as<char>(), ps->s_side_live.as<unsigned>(), (unsigned*)ps->data[2].p,
                        (float*)ps->data[3].p, 1};
        // (+ 1 KB per wave: the four columns' third members, prec_issue_quad)
        k_push_walk_rowsq<3, 4, true, true><<<rgrid, kBlock, lds + (size_t)(kBlock / 64) * 1024, st>>>(PP_ROWSQ_ARGS, rin);
        ps->lazy_rec = 2;  // every member but the origin is in the SoA arrays now
      } else {
        k_push_walk_rowsq<3, 4, true><<<rgrid, kBlock, lds, st>>>(PP_ROWSQ_ARGS);
      }
      PP_LAUNCH_CHECK();
      {  // second pass: a wave owns kPendRegions consecutive queue regions and refills its lanes
        constexpr int G = kPendRegions;
        const unsigned pgrid = (rgrid + G - 1) / G;
        if (mesh->dim == 2)
          k_walk_pending<2><<<pgrid, kBlock, 0, st>>>((int)nwaves, ps->tile_p, G, g_pending_q.as<PendEntry>(),
                                                     g_wave_cnt.as<int>(), mesh->d_records.p,
                                                     elem_ids_dev, looplimit, used);
        else
          k_walk_pending<3><<<pgrid, kBlock, 0, st>>>((int)nwaves, ps->tile_p, G, g_pending_q.as<PendEntry>(),
                                                     g_wave_cnt.as<int>(), mesh->d_records.p,
                                                     elem_ids_dev, looplimit, used);
      }
      ps->cnt2_cur ^= 1;
    } else if (rgrid > 0) {
      if (mesh->dim == 2 && recin) {
        const RecIn rin{ps->s_aos_live.as<char>(), ps->s_side_live.as<unsigned>(), (unsigned*)ps->data[2].p,
                        (float*)ps->data[3].p, ps->rec_rm ? 1 : 0,
                        ps->rec_split ? ps->s_side_live.as<uint4>() : nullptr};
        static const bool occ5 = PP_LAB_ENV("PP_PUSH2D_OCC5") != nullptr;  // (lab build: 96 VGPRs + 60 B of scratch, five waves)
        if (occ5)
          k_push_walk_rows<2, 5, true><<<rgrid, kBlock, 0, st>>>(PP_ROWS_ARGS, rin);
        else
          k_push_walk_rows<2, 4, true><<<rgrid, kBlock, 0, st>>>(PP_ROWS_ARGS, rin);
        ps->lazy_rec = 2;
        ps->zero_z_pending = z_stays_zero;
      } else if (mesh->dim == 2)
        k_push_walk_rows<2, 4><<<rgrid, kBlock, 0, st>>>(PP_ROWS_ARGS);
      else
        k_push_walk_rows<3, 4><<<rgrid, kBlock, 0, st>>>(PP_ROWS_ARGS);
      ps->cnt2_cur ^= 1;
    }
#undef PP_ROWS_ARGS
#undef PP_ROWSQ_ARGS
  } else {
    if ((rc = reset_counters())) return rc;
    used = g_cnt.get();
    // DIM 2 follows search_mesh_2d: the caller's elem_ids are always read (-1 = own element)
    if (mesh->dim == 2) {
    k_push_walk<2><<<grid, kBlock, 0, st>>>(
        ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), mesh->d_records.p,
        mesh->d_class_id.as<int>(), mesh->nelems, PP_MEMBER(ps, m_x, double),
        PP_MEMBER(ps, m_xtgt, double), ps->stride, PP_MEMBER(ps, m_b, float),
        PP_MEMBER(ps, m_phi, float), h, k, d, deg, mesh->tol, elem_ids_dev, elem_ids_seeded, looplimit,
        g_cnt.get());
  } else {
    k_push_walk<3><<<grid, kBlock, 0, st>>>(
        ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), mesh->d_records.p,
        mesh->d_class_id.as<int>(), mesh->nelems, PP_MEMBER(ps, m_x, double),
        PP_MEMBER(ps, m_xtgt, double), ps->stride, PP_MEMBER(ps, m_b, float),
        PP_MEMBER(ps, m_phi, float), h, k, d, deg, mesh->tol, elem_ids_dev, elem_ids_seeded,
        looplimit, g_cnt.get());
    }
  }
  PP_LAUNCH_CHECK();
  g_last_counters = used;
  ps->searched_serial = ++g_search_serial;
  // (the structure's own pair; the plain kernels of a CSR structure count in the process-wide set, which the next
  //  search of ANY structure clears: nothing for a rebuild to carry)
  ps->last_nf_dev = (ps->kind == PP_SCS && ps->cnt2) ? &used->not_found : nullptr;
  if (found) {
    Counters hc;
    PP_HIP_CHECK(hipMemcpyAsync(&hc, used, sizeof(Counters), hipMemcpyDeviceToHost, st));
    PP_HIP_CHECK(hipStreamSynchronize(st));
    *found = (hc.not_found == 0);
  }
  return PP_OK;
}

int pp_push_search_counters(int* not_found, int* not_in_elem, int* unmoved_trusted) {
  Counters hc{};
  if (g_last_counters) {
    PP_HIP_CHECK(hipMemcpyAsync(&hc, g_last_counters, sizeof(Counters), hipMemcpyDeviceToHost, pp::stream()));
    PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  }
  if (not_found) *not_found = hc.not_found;
  if (not_in_elem) *not_in_elem = hc.not_in_elem;
  if (unmoved_trusted) *unmoved_trusted = hc.unmoved;
  return PP_OK;
}

int pp_search_walk_steps(unsigned long long* steps) {
  PP_REQUIRE(steps, "pp_search_walk_steps: null argument");
  *steps = 0;
  if (g_mt_steps) {
    PP_HIP_CHECK(hipMemcpyAsync(steps, g_mt_steps, sizeof(unsigned long long), hipMemcpyDeviceToHost, pp::stream()));
    PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  }
  return PP_OK;
}

int pp_ps_set_origin_trust(pp_ps* ps, int on) {
  PP_REQUIRE(ps, "pp_ps_set_origin_trust: null ps");
  ps->trust_origins = on != 0;
  return PP_OK;
}

}  // extern "C"
