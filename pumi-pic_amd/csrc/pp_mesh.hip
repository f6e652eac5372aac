// pp_mesh.hip -- mesh handle: derives the adjacency the hot path reads from (coords,
// elem2verts), packs the per-element walk records and uploads everything once.
//
// Stands in for the Omega_h queries made by the reference's searches
// (src/pumipic_adjacency.tpp:238-241,394-396,489,497-501; adjacency.hpp:568-574,1030-1036).
// Setup cost only: not on the per-step path (the reference recomputes measure_elements_real and
// mark_exposed_sides on every search_mesh call, SURVEY Q13 -- cached here).
//
// Canonical side numbering: walking elements in id order and local sides in Omega_h template
// order ({0,1},{1,2},{2,0} / {0,2,1},{0,1,3},{1,2,3},{2,0,3}), a side receives the next id the
// first time it is met and keeps the vertex order of that first element.
#include <algorithm>
#include <cmath>
#include <limits>
#include <unordered_map>
#include "pp_geom.hpp"
#include "pp_internal.hpp"

namespace {

struct Key3 {
  int a, b, c;
  bool operator==(const Key3& o) const { return a == o.a && b == o.b && c == o.c; }
};
struct Key3Hash {
  size_t operator()(const Key3& k) const {
    uint64_t h = (uint64_t)(uint32_t)k.a * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)(uint32_t)k.b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    h = (h << 23) | (h >> 41);
    h ^= ((uint64_t)(uint32_t)k.c + 0x165667B1ull) * 0xD6E8FEB86659FD93ull;
    return (size_t)(h ^ (h >> 29));
  }
};

template <class T>
int upload(pp::DevBuf& d, const std::vector<T>& h) {
  PP_HIP_CHECK(d.reserve(std::max<size_t>(h.size() * sizeof(T), 16)));
  if (!h.empty())
    PP_HIP_CHECK(hipMemcpy(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return PP_OK;
}

int derive(pp_mesh& m) {
  const int dim = m.dim, nv = dim + 1, ne = m.nelems;
  static const int TF[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {2, 0, 3}};
  static const int TE[3][2] = {{0, 1}, {1, 2}, {2, 0}};
  m.elem2sides.assign((size_t)ne * nv, -1);
  std::unordered_map<Key3, int, Key3Hash> table;
  table.reserve((size_t)ne * nv);
  std::vector<int> cnt;
  for (int e = 0; e < ne; ++e)
    for (int ls = 0; ls < nv; ++ls) {
      int v[3] = {-1, -1, -1};
      for (int j = 0; j < dim; ++j)
        v[j] = m.elem2verts[(size_t)e * nv + (dim == 3 ? TF[ls][j] : TE[ls][j])];
      int s[3] = {v[0], v[1], v[2]};
      if (dim == 2) {
        if (s[1] < s[0]) std::swap(s[0], s[1]);
      } else {
        std::sort(s, s + 3);
      }
      Key3 k{s[0], s[1], s[2]};
      auto it = table.find(k);
      int sid;
      if (it == table.end()) {
        sid = (int)cnt.size();
        table.emplace(k, sid);
        cnt.push_back(0);
        for (int j = 0; j < dim; ++j) m.side2verts.push_back(v[j]);
      } else {
        sid = it->second;
      }
      cnt[sid]++;
      if (cnt[sid] > 2) {
        pp::set_error("pp_mesh_create: non-manifold side (shared by more than two elements)");
        return PP_EINVAL;
      }
      m.elem2sides[(size_t)e * nv + ls] = sid;
    }
  m.nsides = (int)cnt.size();
  m.side2elems_off.assign((size_t)m.nsides + 1, 0);
  for (int s = 0; s < m.nsides; ++s) m.side2elems_off[s + 1] = m.side2elems_off[s] + cnt[s];
  m.side2elems.assign((size_t)m.side2elems_off[m.nsides], -1);
  m.side_exposed.assign((size_t)m.nsides, 0);
  {
    std::vector<int> fill((size_t)m.nsides, 0);
    for (int e = 0; e < ne; ++e)
      for (int ls = 0; ls < nv; ++ls) {
        const int sid = m.elem2sides[(size_t)e * nv + ls];
        m.side2elems[(size_t)m.side2elems_off[sid] + fill[sid]++] = e;
      }
    for (int s = 0; s < m.nsides; ++s) m.side_exposed[s] = (cnt[s] == 1);
  }
  m.dual_off.assign((size_t)ne + 1, 0);
  for (int e = 0; e < ne; ++e) {
    int c = 0;
    for (int ls = 0; ls < nv; ++ls) c += !m.side_exposed[m.elem2sides[(size_t)e * nv + ls]];
    m.dual_off[e + 1] = m.dual_off[e] + c;
  }
  m.dual_elems.assign((size_t)m.dual_off[ne], -1);
  for (int e = 0; e < ne; ++e) {
    int c = m.dual_off[e];
    for (int ls = 0; ls < nv; ++ls) {
      const int sid = m.elem2sides[(size_t)e * nv + ls];
      if (m.side_exposed[sid]) continue;
      const int a = m.side2elems[m.side2elems_off[sid]], b = m.side2elems[m.side2elems_off[sid] + 1];
      m.dual_elems[c++] = (a == e) ? b : a;
    }
  }
  m.vert2elems_off.assign((size_t)m.nverts + 1, 0);
  for (size_t i = 0; i < m.elem2verts.size(); ++i) m.vert2elems_off[m.elem2verts[i] + 1]++;
  for (int v = 0; v < m.nverts; ++v) m.vert2elems_off[v + 1] += m.vert2elems_off[v];
  m.vert2elems.assign(m.elem2verts.size(), -1);
  {
    std::vector<int> fill((size_t)m.nverts, 0);
    for (int e = 0; e < ne; ++e)
      for (int lv = 0; lv < nv; ++lv) {
        const int v = m.elem2verts[(size_t)e * nv + lv];
        m.vert2elems[(size_t)m.vert2elems_off[v] + fill[v]++] = e;
      }
  }
  // measures (host code of this file is compiled with -ffp-contract=off as well)
  m.elem_measure.assign((size_t)ne, 0.0);
  double min_area = std::numeric_limits<double>::infinity();
  for (int e = 0; e < ne; ++e) {
    double meas;
    if (dim == 2) {
      ppg::V2 p[3];
      for (int i = 0; i < 3; ++i) {
        const int v = m.elem2verts[(size_t)e * 3 + i];
        p[i] = {m.coords[(size_t)v * 2], m.coords[(size_t)v * 2 + 1]};
      }
      meas = ppg::tri_area(p);
    } else {
      ppg::V3 p[4];
      for (int i = 0; i < 4; ++i) {
        const int v = m.elem2verts[(size_t)e * 4 + i];
        p[i] = {m.coords[(size_t)v * 3], m.coords[(size_t)v * 3 + 1], m.coords[(size_t)v * 3 + 2]};
      }
      meas = ppg::tet_volume(p);
    }
    m.elem_measure[e] = meas;
    if (meas < min_area) min_area = meas;
  }
  const double t = 1e-15 / min_area;  // adjacency.tpp:425
  m.tol = (t < 1e-8) ? 1e-8 : t;
  // threshold on the SQUARED length equivalent to `sqrt(s) < tol` (sqrt is monotone, IEEE-exact)
  double s2 = m.tol * m.tol;
  while (std::sqrt(std::nextafter(s2, 0.0)) >= m.tol) s2 = std::nextafter(s2, 0.0);
  while (std::sqrt(s2) < m.tol) s2 = std::nextafter(s2, std::numeric_limits<double>::infinity());
  m.unmoved_sq = s2;
  return PP_OK;
}

int pack_and_upload(pp_mesh& m) {
  const int ne = m.nelems;
  if (m.dim == 2) {
    std::vector<pp_tri_rec> rec((size_t)ne);
    for (int e = 0; e < ne; ++e) {
      pp_tri_rec& r = rec[e];
      for (int i = 0; i < 3; ++i) {
        const int v = m.elem2verts[(size_t)e * 3 + i];
        r.xy[i][0] = m.coords[(size_t)v * 2];
        r.xy[i][1] = m.coords[(size_t)v * 2 + 1];
        const int sid = m.elem2sides[(size_t)e * 3 + i];
        if (m.side_exposed[sid])
          r.nbr[i] = -1;
        else {
          const int a = m.side2elems[m.side2elems_off[sid]], b = m.side2elems[m.side2elems_off[sid] + 1];
          r.nbr[i] = (a == e) ? b : a;
        }
      }
      r.class_id = m.class_id[e];
      for (int i = 0; i < 3; ++i)  // (the packed intersection walk names the entry edge by the neighbour behind it)
        for (int j = i + 1; j < 3; ++j)
          if (r.nbr[i] >= 0 && r.nbr[i] == r.nbr[j]) m.mt_packed_ok = false;
    }
    PP_HIP_CHECK(m.d_records.reserve(std::max<size_t>(rec.size() * sizeof(pp_tri_rec), 64)));
    if (ne) PP_HIP_CHECK(hipMemcpy(m.d_records.p, rec.data(), rec.size() * sizeof(pp_tri_rec),
                                   hipMemcpyHostToDevice));
  } else {
    std::vector<pp_tet_rec> rec((size_t)ne);
    for (int e = 0; e < ne; ++e) {
      pp_tet_rec& r = rec[e];
      for (int i = 0; i < 4; ++i) {
        const int v = m.elem2verts[(size_t)e * 4 + i];
        for (int c = 0; c < 3; ++c) r.xyz[i][c] = m.coords[(size_t)v * 3 + c];
        const int sid = m.elem2sides[(size_t)e * 4 + i];
        if (m.side_exposed[sid])
          r.nbr[i] = -1;
        else {
          const int a = m.side2elems[m.side2elems_off[sid]], b = m.side2elems[m.side2elems_off[sid] + 1];
          r.nbr[i] = (a == e) ? b : a;
        }
      }
      r.vol = m.elem_measure[e];
      r.class_id = m.class_id[e];
      // Moeller-Trumbore face codes (pp_search.hip: k_search_mt3).  ray_intersects_triangle works on the STORED
      // side: faceVerts = coords of bridgeVerts[face_id] in the side's own vertex order, flip = isFaceFlipped
      // (adjacency.tpp:322-331,152-157).  Byte fi of `mt_code`: the tet-local indices (2 bits each) of
      // faceVerts[0], faceVerts[2 - flip], faceVerts[flip + 1] -- the vertex the two edges start from and
      // the far ends of edge1 and edge2.
      unsigned code = 0;
      const int* tv = &m.elem2verts[(size_t)e * 4];
      for (int fi = 0; fi < 4; ++fi) {
        const int sid = m.elem2sides[(size_t)e * 4 + fi];
        const int* fv = &m.side2verts[(size_t)sid * 3];
        const int flip = ppg::is_face_flipped(fi, fv, tv) ? 1 : 0;
        auto local = [&](int v) {
          for (int i = 0; i < 4; ++i)
            if (tv[i] == v) return i;
          return 0;
        };
        const unsigned c = (unsigned)local(fv[0]) | ((unsigned)local(fv[2 - flip]) << 2) | ((unsigned)local(fv[flip + 1]) << 4);
        code |= c << (8 * fi);
      }
      r.mt_code = code;
      // the packed walk names the face a particle came through by the neighbour behind it: needs the
      // neighbours of an element to be distinct (true for any simplicial complex; checked, not assumed)
      for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j)
          if (r.nbr[i] >= 0 && r.nbr[i] == r.nbr[j]) m.mt_packed_ok = false;
    }
    PP_HIP_CHECK(m.d_records.reserve(std::max<size_t>(rec.size() * sizeof(pp_tet_rec), 128)));
    if (ne) PP_HIP_CHECK(hipMemcpy(m.d_records.p, rec.data(), rec.size() * sizeof(pp_tet_rec),
                                   hipMemcpyHostToDevice));
  }
  int rc;
  if ((rc = upload(m.d_coords, m.coords))) return rc;
  if ((rc = upload(m.d_elem2verts, m.elem2verts))) return rc;
  if ((rc = upload(m.d_class_id, m.class_id))) return rc;
  if ((rc = upload(m.d_elem2sides, m.elem2sides))) return rc;
  if ((rc = upload(m.d_side2verts, m.side2verts))) return rc;
  if ((rc = upload(m.d_side2elems_off, m.side2elems_off))) return rc;
  if ((rc = upload(m.d_side2elems, m.side2elems))) return rc;
  if ((rc = upload(m.d_side_exposed, m.side_exposed))) return rc;
  if ((rc = upload(m.d_elem_measure, m.elem_measure))) return rc;
  if ((rc = upload(m.d_dual_off, m.dual_off))) return rc;
  if ((rc = upload(m.d_dual_elems, m.dual_elems))) return rc;
  if ((rc = upload(m.d_vert2elems_off, m.vert2elems_off))) return rc;
  if ((rc = upload(m.d_vert2elems, m.vert2elems))) return rc;
  return PP_OK;
}

}  // namespace

namespace pp {
// Edges of a tet mesh (the reference gets them from Omega_h: ask_down(3, 1) / ask_up(1, 3)).  Same rule as
// the sides in derive(): first-seen numbering over (element, local edge), the stored vertex pair in the
// orientation of the first element that has the edge.
int mesh_edges(const pp_mesh* mesh) {
  pp_mesh& m = *const_cast<pp_mesh*>(mesh);
  PP_REQUIRE(m.dim == 3, "edges are a separate entity dimension only for tet meshes (2-D: the sides)");
  if (m.edges_ready) return PP_OK;
  static const int TE3[6][2] = {{0, 1}, {1, 2}, {2, 0}, {0, 3}, {1, 3}, {2, 3}};
  const int ne = m.nelems;
  m.elem2edges.assign((size_t)ne * 6, -1);
  m.edge2verts.clear();
  std::unordered_map<Key3, int, Key3Hash> table;
  table.reserve((size_t)ne * 2);
  std::vector<int> cnt;
  for (int e = 0; e < ne; ++e)
    for (int le = 0; le < 6; ++le) {
      const int a = m.elem2verts[(size_t)e * 4 + TE3[le][0]], b = m.elem2verts[(size_t)e * 4 + TE3[le][1]];
      const Key3 k{std::min(a, b), std::max(a, b), -1};
      auto it = table.find(k);
      int id;
      if (it == table.end()) {
        id = (int)cnt.size();
        table.emplace(k, id);
        cnt.push_back(0);
        m.edge2verts.push_back(a);
        m.edge2verts.push_back(b);
      } else {
        id = it->second;
      }
      ++cnt[(size_t)id];
      m.elem2edges[(size_t)e * 6 + le] = id;
    }
  m.nedges = (int)cnt.size();
  m.edge2elems_off.assign((size_t)m.nedges + 1, 0);
  for (int i = 0; i < m.nedges; ++i) m.edge2elems_off[(size_t)i + 1] = m.edge2elems_off[(size_t)i] + cnt[(size_t)i];
  m.edge2elems.assign((size_t)m.edge2elems_off[(size_t)m.nedges], -1);
  std::vector<int> fill((size_t)m.nedges, 0);
  for (int e = 0; e < ne; ++e)
    for (int le = 0; le < 6; ++le) {
      const int id = m.elem2edges[(size_t)e * 6 + le];
      m.edge2elems[(size_t)m.edge2elems_off[(size_t)id] + fill[(size_t)id]++] = e;
    }
  int rc;
  if ((rc = upload(m.d_elem2edges, m.elem2edges))) return rc;
  if ((rc = upload(m.d_edge2verts, m.edge2verts))) return rc;
  if ((rc = upload(m.d_edge2elems_off, m.edge2elems_off))) return rc;
  if ((rc = upload(m.d_edge2elems, m.edge2elems))) return rc;
  m.edges_ready = true;
  return PP_OK;
}
}  // namespace pp

extern "C" {

pp_mesh* pp_mesh_create(int dim, int nverts, const double* coords_host, int nelems,
                        const int* elem2verts_host, const int* class_id_host) {
  if ((dim != 2 && dim != 3) || nverts < 0 || nelems < 0 || !coords_host || !elem2verts_host) {
    pp::set_error("pp_mesh_create: bad arguments (dim must be 2 or 3)");
    return nullptr;
  }
  if (!pp::initialised() && pp_init(0) != PP_OK) return nullptr;
  pp_mesh* m = new pp_mesh();
  m->uid = pp::next_version();
  m->dim = dim;
  m->nverts = nverts;
  m->nelems = nelems;
  m->coords.assign(coords_host, coords_host + (size_t)nverts * dim);
  m->elem2verts.assign(elem2verts_host, elem2verts_host + (size_t)nelems * (dim + 1));
  if (class_id_host)
    m->class_id.assign(class_id_host, class_id_host + nelems);
  else
    m->class_id.assign((size_t)nelems, 0);
  for (size_t i = 0; i < m->elem2verts.size(); ++i)
    if (m->elem2verts[i] < 0 || m->elem2verts[i] >= nverts) {
      pp::set_error("pp_mesh_create: elem2verts entry out of range");
      delete m;
      return nullptr;
    }
  if (derive(*m) != PP_OK || pack_and_upload(*m) != PP_OK) {
    delete m;
    return nullptr;
  }
  return m;
}

int pp_mesh_destroy(pp_mesh* m) {
  pp::gyro_map_mesh_gone(m);
  delete m;
  return PP_OK;
}

int pp_mesh_num_edges(const pp_mesh* m) {
  if (!m) return PP_EINVAL;
  if (m->dim == 2) return m->nsides;
  const int rc = pp::mesh_edges(m);
  return rc ? rc : m->nedges;
}

int pp_mesh_info(const pp_mesh* m, int* dim, int* nverts, int* nelems, int* nsides) {
  PP_REQUIRE(m, "pp_mesh_info: null mesh");
  if (dim) *dim = m->dim;
  if (nverts) *nverts = m->nverts;
  if (nelems) *nelems = m->nelems;
  if (nsides) *nsides = m->nsides;
  return PP_OK;
}

double pp_mesh_tolerance(const pp_mesh* m) { return m ? m->tol : 0.0; }

static const void* mesh_array(const pp_mesh* m, int which, size_t* count, size_t* item,
                              const void** host) {
  const void* d = nullptr;
  if (which >= PP_MESH_ELEM2EDGES && which <= PP_MESH_EDGE2ELEMS && m->dim == 3) (void)pp::mesh_edges(m);
#define PP_CASE(W, VEC, DEV, T) \
  case W:                       \
    *count = m->VEC.size();     \
    *item = sizeof(T);          \
    *host = m->VEC.data();      \
    d = m->DEV.p;               \
    break;
  switch (which) {
    PP_CASE(PP_MESH_COORDS, coords, d_coords, double)
    PP_CASE(PP_MESH_ELEM2VERTS, elem2verts, d_elem2verts, int)
    PP_CASE(PP_MESH_CLASS_ID, class_id, d_class_id, int)
    PP_CASE(PP_MESH_ELEM2SIDES, elem2sides, d_elem2sides, int)
    PP_CASE(PP_MESH_SIDE2VERTS, side2verts, d_side2verts, int)
    PP_CASE(PP_MESH_SIDE2ELEMS_OFF, side2elems_off, d_side2elems_off, int)
    PP_CASE(PP_MESH_SIDE2ELEMS, side2elems, d_side2elems, int)
    PP_CASE(PP_MESH_SIDE_EXPOSED, side_exposed, d_side_exposed, signed char)
    PP_CASE(PP_MESH_ELEM_MEASURE, elem_measure, d_elem_measure, double)
    PP_CASE(PP_MESH_DUAL_OFF, dual_off, d_dual_off, int)
    PP_CASE(PP_MESH_DUAL_ELEMS, dual_elems, d_dual_elems, int)
    PP_CASE(PP_MESH_VERT2ELEMS_OFF, vert2elems_off, d_vert2elems_off, int)
    PP_CASE(PP_MESH_VERT2ELEMS, vert2elems, d_vert2elems, int)
    PP_CASE(PP_MESH_ELEM2EDGES, elem2edges, d_elem2edges, int)
    PP_CASE(PP_MESH_EDGE2VERTS, edge2verts, d_edge2verts, int)
    PP_CASE(PP_MESH_EDGE2ELEMS_OFF, edge2elems_off, d_edge2elems_off, int)
    PP_CASE(PP_MESH_EDGE2ELEMS, edge2elems, d_edge2elems, int)
    case PP_MESH_ELEM_RECORDS:
      *count = (size_t)m->nelems;
      *item = (m->dim == 2) ? sizeof(pp_tri_rec) : sizeof(pp_tet_rec);
      *host = nullptr;
      d = m->d_records.p;
      break;
    default:
      *count = 0;
      *item = 0;
      *host = nullptr;
  }
#undef PP_CASE
  return d;
}

const void* pp_mesh_array_dev(const pp_mesh* m, int which, size_t* count) {
  if (!m) return nullptr;
  size_t c = 0, item = 0;
  const void* host = nullptr;
  const void* d = mesh_array(m, which, &c, &item, &host);
  if (count) *count = c;
  return d;
}

int pp_mesh_array_to_host(const pp_mesh* m, int which, void* out_host) {
  PP_REQUIRE(m && out_host, "pp_mesh_array_to_host: null argument");
  size_t c = 0, item = 0;
  const void* host = nullptr;
  const void* d = mesh_array(m, which, &c, &item, &host);
  PP_REQUIRE(item != 0, "pp_mesh_array_to_host: unknown array id");
  if (c == 0) return PP_OK;
  // read back from the DEVICE copy so tests see what kernels see
  PP_HIP_CHECK(hipMemcpy(out_host, d, c * item, hipMemcpyDeviceToHost));
  return PP_OK;
}

}  // extern "C"
