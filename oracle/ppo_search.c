/*
 * ppo_search.c -- ORACLE (test infrastructure, NOT product code).
 *
 * Element-to-element adjacency searches, restated with the reference's unfused kernel
 * sequence (one pass over every slot per kernel per walk iteration + a min-reduction of
 * ptcl_done), i.e. Kokkos::Serial semantics:
 *   search_mesh_2d                src/pumipic_adjacency.hpp:1011-1158
 *   search_mesh_2d_pt             src/pumipic_adjacency.hpp:1160-1252
 *   search_mesh (2-D/3-D, new)    src/pumipic_adjacency.tpp:72-145,231-416,460-654
 *   search_mesh_3d                src/pumipic_adjacency.hpp:314-555
 *   search_mesh (3-D, legacy)     src/pumipic_adjacency.hpp:558-768
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "ppo.h"
#include "ppo_geom.h"

static void* xcalloc(size_t n, size_t s) {
  void* p = calloc(n ? n : 1, s ? s : 1);
  if (!p) {
    fprintf(stderr, "ppo: out of memory\n");
    exit(EXIT_FAILURE);
  }
  return p;
}

#define MEMBER_D(ps, m, pid, i) (((const double*)(ps)->data[m])[(size_t)(i) * (ps)->alloc + (pid)])

static ppo_v2 vec2(const ppo_ps* ps, int m, int pid) {
  ppo_v2 v = {{MEMBER_D(ps, m, pid, 0), MEMBER_D(ps, m, pid, 1)}};
  return v;
}
static ppo_v3 vec3(const ppo_ps* ps, int m, int pid) {
  ppo_v3 v = {{MEMBER_D(ps, m, pid, 0), MEMBER_D(ps, m, pid, 1), MEMBER_D(ps, m, pid, 2)}};
  return v;
}
static void gather_tri(const ppo_mesh* mesh, int elm, int verts[3], ppo_v2 fc[3]) {
  for (int i = 0; i < 3; ++i) {
    verts[i] = mesh->elem2verts[(size_t)elm * 3 + i];
    fc[i].v[0] = mesh->coords[(size_t)verts[i] * 2];
    fc[i].v[1] = mesh->coords[(size_t)verts[i] * 2 + 1];
  }
}
static void gather_tet(const ppo_mesh* mesh, int elm, int verts[4], ppo_v3 M[4]) {
  for (int i = 0; i < 4; ++i) {
    verts[i] = mesh->elem2verts[(size_t)elm * 4 + i];
    for (int c = 0; c < 3; ++c) M[i].v[c] = mesh->coords[(size_t)verts[i] * 3 + c];
  }
}
static int min_done(const int* ptcl_done, int n) {
  int mn = 1; /* callers guarantee n>0 */
#pragma omp parallel for schedule(static) reduction(min : mn)
  for (int i = 0; i < n; ++i)
    if (ptcl_done[i] < mn) mn = ptcl_done[i];
  return mn;
}
static int other_elem(const ppo_mesh* mesh, int bridge, int searchElm) {
  const int first = mesh->side2elems_off[bridge];
  const int A = mesh->side2elems[first];
  const int B = mesh->side2elems[first + 1];
  return (A == searchElm) ? B : A;
}

/* ------------------------------------------------------------------ search_mesh_2d */
int ppo_search_mesh_2d(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt, int m_pid,
                       int* elem_ids, int looplimit, int* loops_out) {
  (void)m_x;
  (void)m_pid;
  const int cap = ps->capacity;
  int* slot_elem = (int*)xcalloc((size_t)cap, sizeof(int));
  unsigned char* slot_mask = (unsigned char*)xcalloc((size_t)cap, 1);
  ppo_ps_slot_info(ps, slot_elem, slot_mask);
  int* ptcl_done = (int*)xcalloc((size_t)cap, sizeof(int));
  int* lastEdge = (int*)xcalloc((size_t)cap, sizeof(int));
  for (int i = 0; i < cap; ++i) {
    ptcl_done[i] = 1;
    lastEdge[i] = -1;
  }
  const int nelems = mesh->nelems;
  int loops = 0, found = 0;
  if (ps->num_ptcls == 0 || cap == 0) { /* parallel_for is a no-op; reference would spin on
                                           get_min of an all-ones array -> found immediately */
    found = 1;
    goto done;
  }
  /* hpp:1045-1062 */
  #pragma omp parallel for schedule(static)
  for (int pid = 0; pid < cap; ++pid) {
    if (slot_elem[pid] < 0) continue;
    if (slot_mask[pid]) {
      if (elem_ids[pid] == -1) elem_ids[pid] = slot_elem[pid];
      ptcl_done[pid] = 0;
      if (elem_ids[pid] == -nelems) {
        elem_ids[pid] = -1;
        ptcl_done[pid] = 1;
      }
    } else {
      elem_ids[pid] = -1;
      ptcl_done[pid] = 1;
    }
  }
  while (!found) {
    /* checkCurrentElm hpp:1067-1084 */
    #pragma omp parallel for schedule(static)
    for (int pid = 0; pid < cap; ++pid) {
      if (slot_mask[pid] && !ptcl_done[pid]) {
        const int searchElm = elem_ids[pid];
        int verts[3];
        ppo_v2 fc[3];
        gather_tri(mesh, searchElm, verts, fc);
        double bcc[3];
        ppo_barycentric_tri(mesh->elem_measure[searchElm], fc, vec2(ps, m_xtgt, pid), bcc);
        ptcl_done[pid] = ppo_all_positive(bcc, 3, PPO_EPSILON);
        lastEdge[pid] = mesh->elem2sides[(size_t)searchElm * 3 + ppo_min3(bcc)];
      }
    }
    /* checkExposedEdges hpp:1086-1095 */
    #pragma omp parallel for schedule(static)
    for (int pid = 0; pid < cap; ++pid) {
      if (slot_mask[pid] && !ptcl_done[pid]) {
        const int exposed = mesh->side_exposed[lastEdge[pid]];
        ptcl_done[pid] = exposed;
        elem_ids[pid] = exposed ? -1 : elem_ids[pid];
      }
    }
    /* setNextElm hpp:1099-1117 */
    #pragma omp parallel for schedule(static)
    for (int pid = 0; pid < cap; ++pid) {
      if (slot_mask[pid] && !ptcl_done[pid])
        elem_ids[pid] = other_elem(mesh, lastEdge[pid], elem_ids[pid]);
    }
    found = 1;
    if (min_done(ptcl_done, cap) == 0) found = 0;
    ++loops;
    if (looplimit && loops >= looplimit) {
      for (int pid = 0; pid < cap; ++pid)
        if (slot_mask[pid] && !ptcl_done[pid]) elem_ids[pid] = -1;
      break;
    }
  }
done:
  if (loops_out) *loops_out = loops;
  free(slot_elem);
  free(slot_mask);
  free(ptcl_done);
  free(lastEdge);
  return found;
}

/* ------------------------------------------------------------------ search_mesh_2d_pt */
int ppo_search_mesh_2d_pt(const ppo_mesh* mesh, const double orig[2], const double dest[2],
                          int pid, int initial_elem, int* loops_out, int looplimit) {
  (void)orig;
  (void)pid;
  int ptcl_done = 0, lastEdge = -1, found = 0, loops = 0;
  int elem_id = initial_elem;
  const ppo_v2 d = {{dest[0], dest[1]}};
  while (!found) {
    if (!ptcl_done) {
      int verts[3];
      ppo_v2 fc[3];
      gather_tri(mesh, elem_id, verts, fc);
      double bcc[3];
      ppo_barycentric_tri(mesh->elem_measure[elem_id], fc, d, bcc);
      ptcl_done = ppo_all_positive(bcc, 3, PPO_EPSILON);
      lastEdge = mesh->elem2sides[(size_t)elem_id * 3 + ppo_min3(bcc)];
    }
    if (!ptcl_done) {
      const int exposed = mesh->side_exposed[lastEdge];
      ptcl_done = exposed;
      if (exposed) elem_id = -1;
    }
    if (!ptcl_done) elem_id = other_elem(mesh, lastEdge, elem_id);
    found = ptcl_done ? 1 : 0;
    ++loops;
    if (loops >= looplimit && !ptcl_done) {
      elem_id = -1;
      break;
    }
  }
  if (loops_out) *loops_out = loops;
  return elem_id;
}

/* Single-point BCC walk through tets: the per-particle work of search_mesh in BCC mode
 * (find_exit_face tpp:276-285, check_model_intersection :365-387, set_new_element :389-416)
 * for ONE target point, used by the tet variant of the gyro ring map. */
int ppo_search_mesh_3d_pt(const ppo_mesh* mesh, const double dest[3], int initial_elem,
                          int looplimit) {
  int elem_id = initial_elem, done = 0, loops = 0;
  const ppo_v3 d = {{dest[0], dest[1], dest[2]}};
  while (!done) {
    int verts[4];
    ppo_v3 M[4];
    gather_tet(mesh, elem_id, verts, M);
    double bcc[4];
    ppo_barycentric_tet(mesh->elem_measure[elem_id], M, d, bcc);
    done = ppo_all_positive(bcc, 4, PPO_EPSILON);
    if (!done) {
      const int side = mesh->elem2sides[(size_t)elem_id * 4 + ppo_min_index(bcc, 4)];
      if (mesh->side_exposed[side]) {
        elem_id = -1;
        done = 1;
      } else {
        elem_id = other_elem(mesh, side, elem_id);
      }
    }
    ++loops;
    if (loops >= looplimit && !done) {
      elem_id = -1;
      break;
    }
  }
  return elem_id;
}

/* ------------------------------------------------------------------ search_mesh (tpp) */
/* check_model_intersection tpp:365-387 = RemoveParticleOnGeometricModelExit tpp:617-639 */
static void default_functor(void* ctx, const ppo_mesh* mesh, ppo_ps* ps, int* elem_ids,
                            int* inter_faces, int* lastExit, double* inter_points, int* ptcl_done,
                            int m_x, int m_xtgt) {
  (void)inter_points;
  (void)m_x;
  (void)m_xtgt;
  const int requireIntersection = *(const int*)ctx;
  const int cap = ps->capacity;
  int* slot_elem = (int*)xcalloc((size_t)cap, sizeof(int));
  unsigned char* slot_mask = (unsigned char*)xcalloc((size_t)cap, 1);
  ppo_ps_slot_info(ps, slot_elem, slot_mask);
#pragma omp parallel for schedule(static)
  for (int pid = 0; pid < cap; ++pid) {
    if (slot_mask[pid] && !ptcl_done[pid]) {
      const int bridge = lastExit[pid];
      const int exposed = mesh->side_exposed[bridge];
      ptcl_done[pid] = exposed;
      if (exposed && requireIntersection)
        inter_faces[pid] = lastExit[pid];
      else
        elem_ids[pid] = exposed ? -1 : elem_ids[pid]; /* leaves domain if exposed */
    }
  }
  free(slot_elem);
  free(slot_mask);
}
int ppo_search_mesh(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt, int m_pid,
                    int* elem_ids, int elem_ids_seeded, int requireIntersection, int* inter_faces,
                    double* inter_points, int looplimit, int* loops_out, int* num_not_in_elem) {
  /* search_mesh tpp:641-654: trace_particle_through_mesh with the default functor */
  int ri = requireIntersection;
  return ppo_trace_particle_through_mesh(mesh, ps, m_x, m_xtgt, m_pid, elem_ids, elem_ids_seeded,
                                         requireIntersection, inter_faces, inter_points, looplimit,
                                         loops_out, num_not_in_elem, default_functor, &ri);
}
/* trace_particle_through_mesh tpp:460-615; func runs between find_exit_face and set_new_element
 * (tpp:561-565) with the arrays the reference hands its functor */
int ppo_trace_particle_through_mesh(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt,
                                    int m_pid, int* elem_ids, int elem_ids_seeded,
                                    int requireIntersection, int* inter_faces,
                                    double* inter_points, int looplimit, int* loops_out,
                                    int* num_not_in_elem, ppo_trace_functor func, void* ctx) {
  (void)m_pid;
  const int cap = ps->capacity;
  const int dim = mesh->dim;
  int* slot_elem = (int*)xcalloc((size_t)cap, sizeof(int));
  unsigned char* slot_mask = (unsigned char*)xcalloc((size_t)cap, 1);
  ppo_ps_slot_info(ps, slot_elem, slot_mask);
  int* ptcl_done = (int*)xcalloc((size_t)cap, sizeof(int));
  int* lastExit = (int*)xcalloc((size_t)cap, sizeof(int));
  for (int i = 0; i < cap; ++i) lastExit[i] = -1;
  const int useBcc = !requireIntersection;
  const double tol = ppo_compute_tolerance_from_area(mesh);
  int loops = 0, found = 0, notIn = 0;
  if (ps->num_ptcls == 0 || cap == 0) {
    if (!elem_ids_seeded)
      for (int i = 0; i < cap; ++i) elem_ids[i] = -1;
    found = 1;
    goto done;
  }
  /* setInitial tpp:504-522 */
  if (!elem_ids_seeded) {
    for (int i = 0; i < cap; ++i) elem_ids[i] = -1;
    for (int pid = 0; pid < cap; ++pid) {
      if (slot_elem[pid] < 0) continue;
      if (slot_mask[pid])
        elem_ids[pid] = slot_elem[pid];
      else
        ptcl_done[pid] = 1;
    }
  } else {
    for (int pid = 0; pid < cap; ++pid) {
      if (slot_elem[pid] < 0) continue;
      if ((slot_mask[pid] && elem_ids[pid] == -1) || !slot_mask[pid]) ptcl_done[pid] = 1;
    }
  }
  /* slots never visited by parallel_for keep ptcl_done=0 in the reference only when they do
   * not exist; every slot < capacity is visited for SCS.  For CSR the tail padding is not
   * visited: mark it done so the min-reduction matches a structure without tail slots. */
  for (int pid = 0; pid < cap; ++pid)
    if (slot_elem[pid] < 0) ptcl_done[pid] = 1;
  /* finishUnmoved tpp:525-533 */
  #pragma omp parallel for schedule(static)
  for (int pid = 0; pid < cap; ++pid) {
    if (slot_mask[pid]) {
      const ppo_v3 d = ppo_sub3(vec3(ps, m_xtgt, pid), vec3(ps, m_x, pid));
      if (ppo_norm3(d) < tol) ptcl_done[pid] = 1;
    }
  }
  if (requireIntersection) { /* tpp:535-549 */
    for (int pid = 0; pid < cap; ++pid) {
      for (int i = 0; i < dim; ++i) inter_points[(size_t)dim * pid + i] = 0;
      inter_faces[pid] = -1;
    }
  }
  /* check_initial_parents tpp:72-145 */
  #pragma omp parallel for schedule(static) reduction(+ : notIn)
  for (int pid = 0; pid < cap; ++pid) {
    if (slot_mask[pid] && !ptcl_done[pid]) {
      const int searchElm = elem_ids[pid];
      int inside;
      if (dim == 2) {
        int verts[3];
        ppo_v2 fc[3];
        gather_tri(mesh, searchElm, verts, fc);
        double bcc[3];
        ppo_barycentric_tri(mesh->elem_measure[searchElm], fc, vec2(ps, m_x, pid), bcc);
        inside = ppo_all_positive(bcc, 3, tol);
      } else {
        int verts[4];
        ppo_v3 M[4];
        gather_tet(mesh, searchElm, verts, M);
        double bcc[4];
        ppo_barycentric_tet(mesh->elem_measure[searchElm], M, vec3(ps, m_x, pid), bcc);
        inside = ppo_all_positive(bcc, 4, tol);
      }
      if (!inside) {
        ++notIn;
        elem_ids[pid] = -1;
        ptcl_done[pid] = 1;
      }
    }
  }
  while (!found) {
    /* find_exit_face tpp:231-363 */
    #pragma omp parallel for schedule(static)
    for (int pid = 0; pid < cap; ++pid) {
      if (!(slot_mask[pid] && !ptcl_done[pid])) continue;
      const int searchElm = elem_ids[pid];
      if (useBcc && dim == 2) {
        int verts[3];
        ppo_v2 fc[3];
        gather_tri(mesh, searchElm, verts, fc);
        double bcc[3];
        ppo_barycentric_tri(mesh->elem_measure[searchElm], fc, vec2(ps, m_xtgt, pid), bcc);
        ptcl_done[pid] = ppo_all_positive(bcc, 3, PPO_EPSILON);
        lastExit[pid] = mesh->elem2sides[(size_t)searchElm * 3 + ppo_min3(bcc)];
      } else if (useBcc && dim == 3) {
        int verts[4];
        ppo_v3 M[4];
        gather_tet(mesh, searchElm, verts, M);
        double bcc[4];
        ppo_barycentric_tet(mesh->elem_measure[searchElm], M, vec3(ps, m_xtgt, pid), bcc);
        ptcl_done[pid] = ppo_all_positive(bcc, 4, PPO_EPSILON);
        lastExit[pid] = mesh->elem2sides[(size_t)searchElm * 4 + ppo_min_index(bcc, 4)];
      } else if (dim == 2) {
        int faceVerts[3];
        ppo_v2 fc[3];
        gather_tri(mesh, searchElm, faceVerts, fc);
        const ppo_v2 dest = vec2(ps, m_xtgt, pid), orig = vec2(ps, m_x, pid);
        ppo_v2 xpts = {{0, 0}};
        const int prevExit = lastExit[pid];
        lastExit[pid] = -1;
        for (int ei = 0; ei < 3; ++ei) {
          const int edge_id = mesh->elem2sides[(size_t)searchElm * 3 + ei];
          if (edge_id == prevExit) continue;
          int ev2v[2];
          ppo_v2 edge[2];
          for (int q = 0; q < 2; ++q) {
            ev2v[q] = mesh->side2verts[(size_t)edge_id * 2 + q];
            edge[q].v[0] = mesh->coords[(size_t)ev2v[q] * 2];
            edge[q].v[1] = mesh->coords[(size_t)ev2v[q] * 2 + 1];
          }
          const int flip = ppo_is_edge_flipped(ei, ev2v, faceVerts);
          const int success = ppo_line_edge_2d(edge, orig, dest, &xpts, tol, flip);
          if (success) {
            lastExit[pid] = edge_id;
            inter_points[2 * (size_t)pid] = xpts.v[0];
            inter_points[2 * (size_t)pid + 1] = xpts.v[1];
          }
        }
        ptcl_done[pid] = (lastExit[pid] == -1);
      } else {
        int tetv2v[4];
        ppo_v3 M[4];
        gather_tet(mesh, searchElm, tetv2v, M);
        const ppo_v3 dest = vec3(ps, m_xtgt, pid), orig = vec3(ps, m_x, pid);
        ppo_v3 xpts = {{0, 0, 0}};
        const int prevExit = lastExit[pid];
        lastExit[pid] = -1;
        double quality = -1;
        int bestFace = -1;
        for (int fi = 0; fi < 4; ++fi) {
          const int face_id = mesh->elem2sides[(size_t)searchElm * 4 + fi];
          if (face_id == prevExit) continue;
          int fv2v[3];
          ppo_v3 face[3];
          for (int q = 0; q < 3; ++q) {
            fv2v[q] = mesh->side2verts[(size_t)face_id * 3 + q];
            for (int c = 0; c < 3; ++c) face[q].v[c] = mesh->coords[(size_t)fv2v[q] * 3 + c];
          }
          const int flip = ppo_is_face_flipped(fi, fv2v, tetv2v);
          double dproj, closeness, param;
          const int success = ppo_ray_intersects_triangle(face, orig, dest, &xpts, tol, flip,
                                                          &dproj, &closeness, &param);
          if (success) {
            lastExit[pid] = face_id;
            for (int c = 0; c < 3; ++c) inter_points[3 * (size_t)pid + c] = xpts.v[c];
          }
          if (dproj > -tol && (quality < 0 || closeness < quality) && lastExit[pid] == -1) {
            quality = closeness;
            bestFace = face_id;
            for (int c = 0; c < 3; ++c) inter_points[3 * (size_t)pid + c] = xpts.v[c];
          }
        }
        if (lastExit[pid] == -1) lastExit[pid] = bestFace;
        ptcl_done[pid] = (lastExit[pid] == -1);
      }
    }
    /* the functor (tpp:563) */
    func(ctx, mesh, ps, elem_ids, inter_faces, lastExit, inter_points, ptcl_done, m_x, m_xtgt);
    /* set_new_element tpp:389-416.  A functor that leaves a particle unfinished on an exposed side
     * makes the reference read past the side's single up-adjacent element; here it leaves (-1). */
    #pragma omp parallel for schedule(static)
    for (int pid = 0; pid < cap; ++pid)
      if (slot_mask[pid] && !ptcl_done[pid]) {
        const int b = lastExit[pid];
        if (mesh->side2elems_off[b + 1] - mesh->side2elems_off[b] < 2)
          elem_ids[pid] = -1;
        else
          elem_ids[pid] = other_elem(mesh, b, elem_ids[pid]);
      }
    found = 1;
    if (min_done(ptcl_done, cap) == 0) found = 0;
    ++loops;
    if (looplimit && loops >= looplimit) {
      for (int pid = 0; pid < cap; ++pid)
        if (slot_mask[pid] && !ptcl_done[pid]) elem_ids[pid] = -1;
      break;
    }
  }
done:
  if (loops_out) *loops_out = loops;
  if (num_not_in_elem) *num_not_in_elem = notIn;
  free(slot_elem);
  free(slot_mask);
  free(ptcl_done);
  free(lastExit);
  return found;
}

/* ------------------------------------------------------------------ search_mesh_3d
 * src/pumipic_adjacency.hpp:314-555, kernel by kernel (fill, checkParent, then per iteration
 * checkCurrentElm, findIntersection, processUndetected, copy_elem_ids, min reduction).
 * tol = 1e-20 for the containment and the intersection tests (hpp:330).
 * processUndetected indexes the dual VALUE array by a face id (hpp:510); not
 * replicated -- the neighbour across the max-projection face is taken.
 * Returns found, or -2 when checkParent would abort (hpp:373-379). */
static int point_within_tet(const ppo_mesh* mesh, ppo_v3 pos, int elem, double tol) {
  int verts[4];
  ppo_v3 M[4];
  double bcc[4];
  gather_tet(mesh, elem, verts, M);
  ppo_barycentric_coords_tet(M, pos, bcc, tol); /* isPointWithinElemTet hpp:300-305 */
  return ppo_all_positive(bcc, 4, tol);
}
int ppo_search_mesh_3d(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt, int m_pid,
                       int* elem_ids, int elem_ids_seeded, double* xpoints_d, int* xface_d,
                       int looplimit, int* loops_out) {
  (void)m_pid;
  const double tol = 1.0e-20;
  const int cap = ps->capacity;
  int* slot_elem = (int*)xcalloc((size_t)cap, sizeof(int));
  unsigned char* slot_mask = (unsigned char*)xcalloc((size_t)cap, 1);
  ppo_ps_slot_info(ps, slot_elem, slot_mask);
  int* ptcl_done = (int*)xcalloc((size_t)cap, sizeof(int));
  int* elem_ids_next = (int*)xcalloc((size_t)cap, sizeof(int));
  for (int i = 0; i < cap; ++i) {
    ptcl_done[i] = 1; /* hpp:348 */
    elem_ids_next[i] = -1;
  }
  int loops = 0, found = 0, aborted = 0;
  /* fill hpp:358-369 (slots of rows outside the structure behave as masked-out slots) */
  for (int pid = 0; pid < cap; ++pid) {
    if (slot_elem[pid] >= 0 && slot_mask[pid]) {
      if (!elem_ids_seeded) elem_ids[pid] = slot_elem[pid];
      ptcl_done[pid] = (elem_ids[pid] == -1) * 2;
    } else {
      elem_ids[pid] = -1;
      ptcl_done[pid] = 2;
    }
  }
  /* checkParent hpp:371-382: the ROW element e, not elem_ids[pid] */
  for (int pid = 0; pid < cap; ++pid)
    if (slot_elem[pid] >= 0 && slot_mask[pid] && ptcl_done[pid] != 2)
      if (!point_within_tet(mesh, vec3(ps, m_x, pid), slot_elem[pid], tol)) aborted = 1;
  while (!found) {
    /* checkCurrentElm hpp:397-412 */
    for (int pid = 0; pid < cap; ++pid) {
      if (!(slot_elem[pid] >= 0 && slot_mask[pid] && !ptcl_done[pid])) continue;
      const int searchElm = elem_ids[pid];
      const int inParent = point_within_tet(mesh, vec3(ps, m_xtgt, pid), searchElm, tol);
      ptcl_done[pid] = inParent ? 2 : 0;
      elem_ids_next[pid] = searchElm;
    }
    /* findIntersection hpp:414-473 */
    for (int pid = 0; pid < cap; ++pid) {
      if (!(slot_elem[pid] >= 0 && slot_mask[pid] && ptcl_done[pid] < 2)) continue;
      const int searchElm = elem_ids[pid];
      int tetv2v[4];
      ppo_v3 M[4];
      gather_tet(mesh, searchElm, tetv2v, M);
      const ppo_v3 dest = vec3(ps, m_xtgt, pid), orig = vec3(ps, m_x, pid);
      int dual_elem_id = mesh->dual_off[searchElm];
      int adj_id = -1, ind_exp = -1;
      double projd[4] = {0, 0, 0, 0};
      ppo_v3 xpts = {{0, 0, 0}};
      int face_ids[4];
      for (int fi = 0; fi < 4; ++fi) {
        const int face_id = mesh->elem2sides[(size_t)searchElm * 4 + fi];
        face_ids[fi] = face_id;
        ppo_v3 xpoint = {{0, 0, 0}};
        int fv2v[3];
        ppo_v3 face[3];
        for (int q = 0; q < 3; ++q) {
          fv2v[q] = mesh->side2verts[(size_t)face_id * 3 + q];
          for (int c = 0; c < 3; ++c) face[q].v[c] = mesh->coords[(size_t)fv2v[q] * 3 + c];
        }
        const int flip = ppo_is_face_flipped(fi, fv2v, tetv2v);
        const int det = ppo_line_triangle_intx_simple(face, orig, dest, &xpoint, &projd[fi], flip, tol);
        const int exposed = mesh->side_exposed[face_id];
        if (det && exposed) {
          ind_exp = fi;
          xpts = xpoint;
        }
        if (det && !exposed) adj_id = dual_elem_id;
        if (!exposed) ++dual_elem_id;
      }
      if (ind_exp >= 0) { /* wall collision */
        for (int i = 0; i < 3; ++i) xpoints_d[(size_t)pid * 3 + i] = xpts.v[i];
        xface_d[pid] = face_ids[ind_exp];
        elem_ids_next[pid] = -1;
        ptcl_done[pid] = 2;
      }
      if (adj_id >= 0) { /* interior */
        elem_ids_next[pid] = mesh->dual_elems[adj_id];
        ptcl_done[pid] = 1;
      }
    }
    /* processUndetected hpp:475-519 */
    for (int pid = 0; pid < cap; ++pid) {
      const int done = ptcl_done[pid];
      ptcl_done[pid] = (done < 2) ? 0 : 2;
      if (!(slot_elem[pid] >= 0 && slot_mask[pid] && done < 1)) continue;
      const int searchElm = elem_ids[pid];
      int tetv2v[4];
      ppo_v3 M[4];
      gather_tet(mesh, searchElm, tetv2v, M);
      const ppo_v3 dest = vec3(ps, m_xtgt, pid), orig = vec3(ps, m_x, pid);
      double projd[4] = {-1, -1, -1, -1};
      double xpoints[12] = {0};
      int face_ids[4];
      for (int fi = 0; fi < 4; ++fi) {
        const int face_id = mesh->elem2sides[(size_t)searchElm * 4 + fi];
        face_ids[fi] = face_id;
        ppo_v3 xpoint = {{0, 0, 0}};
        int fv2v[3];
        ppo_v3 face[3];
        for (int q = 0; q < 3; ++q) {
          fv2v[q] = mesh->side2verts[(size_t)face_id * 3 + q];
          for (int c = 0; c < 3; ++c) face[q].v[c] = mesh->coords[(size_t)fv2v[q] * 3 + c];
        }
        const int flip = ppo_is_face_flipped(fi, fv2v, tetv2v);
        ppo_line_triangle_intx_simple(face, orig, dest, &xpoint, &projd[fi], flip, tol);
        for (int i = 0; i < 3; ++i) xpoints[fi * 3 + i] = xpoint.v[i];
      }
      const int max_ind = ppo_max_index(projd, 4);
      const int face_id = face_ids[max_ind];
      if (mesh->side_exposed[face_id]) {
        elem_ids_next[pid] = -1;
        for (int i = 0; i < 3; ++i) xpoints_d[(size_t)pid * 3 + i] = xpoints[max_ind * 3 + i];
        xface_d[pid] = face_id;
        ptcl_done[pid] = 2;
      } else {
        elem_ids_next[pid] = other_elem(mesh, face_id, searchElm); /* Q3: see header */
      }
    }
    found = 1;
    for (int i = 0; i < cap; ++i) elem_ids[i] = elem_ids_next[i]; /* copy_elem_ids hpp:521-524 */
    if (min_done(ptcl_done, cap) == 0) found = 0;
    ++loops;
    if (looplimit && loops >= looplimit) break; /* hpp:531-552 */
  }
  if (loops_out) *loops_out = loops;
  free(slot_elem);
  free(slot_mask);
  free(ptcl_done);
  free(elem_ids_next);
  if (aborted) return -2;
  return found;
}

/* ------------------------------------------------------------------ legacy 3-D search_mesh */
int ppo_search_mesh_legacy3d(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt, int m_pid,
                             int* elem_ids, int elem_ids_seeded, double* xpoints_d, int* xface_d,
                             int looplimit, int* loops_out) {
  (void)m_pid;
  const double tol = 1.0e-10;
  const int cap = ps->capacity;
  int* slot_elem = (int*)xcalloc((size_t)cap, sizeof(int));
  unsigned char* slot_mask = (unsigned char*)xcalloc((size_t)cap, 1);
  ppo_ps_slot_info(ps, slot_elem, slot_mask);
  int* ptcl_done = (int*)xcalloc((size_t)cap, sizeof(int));
  int* elem_ids_next = (int*)xcalloc((size_t)cap, sizeof(int));
  for (int i = 0; i < cap; ++i) elem_ids_next[i] = -1;
  int loops = 0, found = 0, aborted = 0;
  if (ps->num_ptcls == 0 || cap == 0) {
    found = 1;
    goto done;
  }
  /* fill hpp:586-598 */
  for (int pid = 0; pid < cap; ++pid) {
    if (slot_elem[pid] < 0) {
      ptcl_done[pid] = 1;
      if (!elem_ids_seeded) elem_ids[pid] = -1;
      continue;
    }
    if (slot_mask[pid]) {
      if (!elem_ids_seeded) elem_ids[pid] = slot_elem[pid];
      ptcl_done[pid] = (elem_ids[pid] == -1);
    } else {
      elem_ids[pid] = -1;
      ptcl_done[pid] = 1;
    }
  }
  while (!found) {
    for (int pid = 0; pid < cap; ++pid) {
      if (!(slot_mask[pid] && !ptcl_done[pid])) continue;
      const int elmId = elem_ids[pid];
      int tetv2v[4];
      ppo_v3 M[4];
      gather_tet(mesh, elmId, tetv2v, M);
      const ppo_v3 dest = vec3(ps, m_xtgt, pid), orig = vec3(ps, m_x, pid);
      double bcc[4];
      if (loops == 0) {
        ppo_find_barycentric_tet(M, orig, bcc);
        if (!ppo_all_positive(bcc, 4, tol)) aborted = 1; /* OMEGA_H_CHECK(false) hpp:622-626 */
      }
      int intersected = 0;
      ppo_find_barycentric_tet(M, dest, bcc);
      if (ppo_all_positive(bcc, 4, tol)) {
        elem_ids_next[pid] = elmId;
        ptcl_done[pid] = 1;
      } else {
        double dproj[4] = {-1, -1, -1, -1};
        double xpoints[12] = {0};
        int exposed_faces[4] = {0, 0, 0, 0};
        int xface_ids[4] = {-1, -1, -1, -1};
        int dual_elem_id = mesh->dual_off[elmId];
        int findex = 0;
        for (int iface = elmId * 4; iface < elmId * 4 + 4; ++iface) {
          const int face_id = mesh->elem2sides[iface];
          ppo_v3 xpoint = {{0, 0, 0}};
          const int exposed = mesh->side_exposed[face_id];
          exposed_faces[findex] = exposed;
          xface_ids[findex] = face_id;
          int fv2v[3];
          ppo_v3 face[3];
          for (int q = 0; q < 3; ++q) {
            fv2v[q] = mesh->side2verts[(size_t)face_id * 3 + q];
            for (int c = 0; c < 3; ++c) face[q].v[c] = mesh->coords[(size_t)fv2v[q] * 3 + c];
          }
          const int matInd1 = ppo_face_map(findex * 2);
          const int matInd2 = ppo_face_map(findex * 2 + 1);
          int flip = 1;
          if (fv2v[1] == tetv2v[matInd1] && fv2v[2] == tetv2v[matInd2]) flip = 0;
          intersected = ppo_line_triangle_intx_simple(face, orig, dest, &xpoint, &dproj[findex],
                                                      flip, tol);
          for (int i = 0; i < 3; ++i) xpoints[findex * 3 + i] = xpoint.v[i];
          if (intersected && exposed) {
            ptcl_done[pid] = 1;
            for (int i = 0; i < 3; ++i) xpoints_d[(size_t)pid * 3 + i] = xpoint.v[i];
            xface_d[pid] = face_id;
            elem_ids_next[pid] = -1;
            break;
          } else if (intersected && !exposed) {
            elem_ids_next[pid] = mesh->dual_elems[dual_elem_id];
            break;
          }
          if (!exposed) ++dual_elem_id;
          ++findex;
        }
        if (!intersected) {
          const int max_ind = ppo_max_index(dproj, 4);
          if (dproj[max_ind] >= 0) {
            const int fid = xface_ids[max_ind];
            if (exposed_faces[max_ind]) {
              elem_ids_next[pid] = -1;
              for (int i = 0; i < 3; ++i) xpoints_d[(size_t)pid * 3 + i] = xpoints[max_ind * 3 + i];
              xface_d[pid] = fid;
              ptcl_done[pid] = 1;
            } else {
              /* reference indexes the dual VALUE array by a face id (hpp:726);
               * not replicated -- take the neighbour across the max-dproj face. */
              elem_ids_next[pid] = other_elem(mesh, fid, elmId);
            }
          } else {
            elem_ids_next[pid] = -1;
            ptcl_done[pid] = 1;
          }
        }
      }
    }
    found = 1;
    for (int i = 0; i < cap; ++i) elem_ids[i] = elem_ids_next[i]; /* copy_elem_ids hpp:745-748 */
    if (min_done(ptcl_done, cap) == 0) found = 0;
    ++loops;
    if (looplimit && loops > looplimit) break;
  }
done:
  if (loops_out) *loops_out = loops;
  free(slot_elem);
  free(slot_mask);
  free(ptcl_done);
  free(elem_ids_next);
  if (aborted) return -2;
  return found;
}
