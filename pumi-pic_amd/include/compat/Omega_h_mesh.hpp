// compat/Omega_h_mesh.hpp -- NOT Omega_h.  The Omega_h names the reference's drivers spell next to the
// particle_structs API (test/pseudoXGCm.cpp, test/pseudoPushAndSearch.cpp, test/gyroScatter.hpp), on this library's
// mesh handle: Omega_h::Mesh is pumipic::Mesh (pumipic_adjacency.hpp), Write/Read are pumipic::View.  Here: Few /
// Vector / Matrix and the gather_* helpers of device lambdas, get_sum / get_max / get_bounding_box, mark_exposed_sides
// / mark_up, Library + the mesh readers (Gmsh .msh through pumipic_gmsh.hpp; the library's flat container in the
// place of .osh, whose format lives in Omega_h's sources only), a no-op vtk writer, pumipic::Library and
// pumipic::read.
#pragma once
#include <array>
#include <initializer_list>
#include <sstream>
#include "../pumipic_mpi.hpp"
#include "../pumipic_adjacency.hpp"
#include "../pumipic_gmsh.hpp"
#include "Kokkos_Core.hpp"

namespace Omega_h {
typedef pumipic::CommPtr CommPtr;
typedef Read<Byte> Bytes;

// ---- small fixed-size containers of device lambdas (Omega_h_few.hpp, _vector.hpp, _matrix.hpp)
template <class T, int n>
struct Few {
  T a_[n];
  __host__ __device__ T& operator[](int i) { return a_[i]; }
  __host__ __device__ const T& operator[](int i) const { return a_[i]; }
  __host__ __device__ static constexpr int size() { return n; }
  __host__ __device__ T* data() { return a_; }
  __host__ __device__ const T* data() const { return a_; }
};
template <int n>
struct Vector : Few<Real, n> {
  __host__ __device__ Vector() {}
  __host__ __device__ Vector(const Few<Real, n>& f) : Few<Real, n>(f) {}
  __host__ __device__ Vector(std::initializer_list<Real> l) {  // Vector<3> p{0.0, -0.2, -0.5}
    int i = 0;
    for (const Real* q = l.begin(); q != l.end() && i < n; ++q) this->a_[i++] = *q;
    for (; i < n; ++i) this->a_[i] = 0;
  }
};
template <int n>
__host__ __device__ inline Vector<n> operator+(const Vector<n>& a, const Vector<n>& b) {
  Vector<n> c;
  for (int i = 0; i < n; ++i) c[i] = a[i] + b[i];
  return c;
}
template <int n>
__host__ __device__ inline Vector<n> operator-(const Vector<n>& a, const Vector<n>& b) {
  Vector<n> c;
  for (int i = 0; i < n; ++i) c[i] = a[i] - b[i];
  return c;
}
template <int n>
__host__ __device__ inline Vector<n> operator*(const Vector<n>& a, Real b) {
  Vector<n> c;
  for (int i = 0; i < n; ++i) c[i] = a[i] * b;
  return c;
}
template <int n>
__host__ __device__ inline Vector<n> operator*(Real b, const Vector<n>& a) { return a * b; }
template <int n>
__host__ __device__ inline Vector<n> operator/(const Vector<n>& a, Real b) {
  Vector<n> c;
  for (int i = 0; i < n; ++i) c[i] = a[i] / b;
  return c;
}
template <int n>
__host__ __device__ inline Real operator*(const Vector<n>& a, const Vector<n>& b) {  // inner product
  Real s = a[0] * b[0];
  for (int i = 1; i < n; ++i) s += a[i] * b[i];
  return s;
}
template <int n>
__host__ __device__ inline Real norm(const Vector<n>& a) { return ::sqrt(a * a); }
// m rows, n columns, stored as n column vectors (Matrix<3,4>: the four vertices of a tet)
template <int m, int n>
struct Matrix : Few<Vector<m>, n> {
  __host__ __device__ Matrix() {}
  __host__ __device__ Matrix(const Few<Vector<m>, n>& f) : Few<Vector<m>, n>(f) {}
  __host__ __device__ Matrix(std::initializer_list<Vector<m>> l) {  // Matrix<3, 4> M{p1, p2, p3, p4}: the columns
    int j = 0;
    for (const Vector<m>* q = l.begin(); q != l.end() && j < n; ++q) this->a_[j++] = *q;
  }
  __host__ __device__ Matrix(std::initializer_list<Real> l) {  // m*n scalars, column after column
    int k = 0;
    for (const Real* q = l.begin(); q != l.end() && k < m * n; ++q, ++k) this->a_[k / m][k % m] = *q;
  }
};
// ---- vector algebra of device lambdas (Omega_h_vector.hpp, published definitions)
template <int n>
__host__ __device__ inline Real inner_product(const Vector<n>& a, const Vector<n>& b) { return a * b; }
template <int n>
__host__ __device__ inline Vector<n> normalize(const Vector<n>& a) { return a / norm(a); }
template <int n>
__host__ __device__ inline Vector<n> zero_vector() {
  Vector<n> v;
  for (int i = 0; i < n; ++i) v[i] = 0.0;
  return v;
}
__host__ __device__ inline Vector<3> cross(const Vector<3>& a, const Vector<3>& b) {
  Vector<3> c;
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
  return c;
}
__host__ __device__ inline Real cross(const Vector<2>& a, const Vector<2>& b) { return a[0] * b[1] - a[1] * b[0]; }
__host__ __device__ inline Vector<2> perp(const Vector<2>& a) {
  Vector<2> c;
  c[0] = -a[1];
  c[1] = a[0];
  return c;
}
// are_close (Omega_h_scalar.hpp): relative difference with a floor, tol = floor = 1e-10
__host__ __device__ inline bool are_close(Real a, Real b, Real tol = 1e-10, Real floor = 1e-10) {
  const Real am = ::fabs(a), bm = ::fabs(b);
  if (am <= floor && bm <= floor) return true;
  return ::fabs(b - a) / (am < bm ? bm : am) <= tol;
}
// the entity opposite a boundary entity of a simplex (Omega_h_simplex.hpp simplex_opposite_template): for
// (triangle, edge) and (tet, face) the vertex that the boundary entity does not hold, and back
__host__ __device__ inline Int simplex_opposite_template(Int elem_dim, Int bdry_dim, Int which_bdry) {
  if (elem_dim == 2 && bdry_dim == 1) return (which_bdry + 2) % 3;       // edges {0,1},{1,2},{2,0}
  if (elem_dim == 2 && bdry_dim == 0) return (which_bdry + 1) % 3;
  if (elem_dim == 3 && bdry_dim == 2) return which_bdry == 0 ? 3 : which_bdry == 1 ? 2 : which_bdry == 2 ? 0 : 1;
  if (elem_dim == 3 && bdry_dim == 0) return which_bdry == 0 ? 2 : which_bdry == 1 ? 3 : which_bdry == 2 ? 1 : 0;
  return -1;
}
// barycentric coordinates of a point, one per VERTEX of the simplex (Omega_h_shape.hpp barycentric_from_global):
// here by ratios of signed measures -- equal to Omega_h's solve in exact arithmetic, not bit-matched to it
__host__ __device__ inline Vector<4> barycentric_tet_vertex_major(const Vector<3>& p, const Few<Vector<3>, 4>& x) {
  Vector<4> b;
  const Real vol = inner_product(cross(x[1] - x[0], x[2] - x[0]), x[3] - x[0]);
  for (int i = 0; i < 4; ++i) {
    Few<Vector<3>, 4> y = x;
    y[i] = p;
    b[i] = inner_product(cross(y[1] - y[0], y[2] - y[0]), y[3] - y[0]) / vol;
  }
  return b;
}
template <int sdim, int edim>
__host__ __device__ inline Vector<edim + 1> barycentric_from_global(const Vector<sdim>& p,
                                                                     const Few<Vector<sdim>, edim + 1>& x) {
  if constexpr (sdim == 3 && edim == 3) {
    return barycentric_tet_vertex_major(p, x);
  } else {
    static_assert(sdim == 2 && edim == 2, "barycentric_from_global: triangles in 2-D and tets in 3-D");
    Vector<3> b;
    const Real area = cross(x[1] - x[0], x[2] - x[0]);
    b[0] = cross(x[1] - p, x[2] - p) / area;
    b[1] = cross(x[2] - p, x[0] - p) / area;
    b[2] = cross(x[0] - p, x[1] - p) / area;
    return b;
  }
}
// OMEGA_H_CHECK (Omega_h_fail.hpp): always on; prints and stops -- on the device the wave traps, which the host sees
// as a failed synchronisation
__host__ __device__ inline void check_fail(const char* what, const char* file, int line) {
  printf("assertion %s failed at %s +%d\n", what, file, line);
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_trap();
#else
  abort();
#endif
}
#define OMEGA_H_CHECK(cond) ((cond) ? ((void)0) : ::Omega_h::check_fail(#cond, __FILE__, __LINE__))
#define OMEGA_H_CHECK_PRINTF(cond, format, ...)                                  \
  do {                                                                           \
    if (!(cond)) {                                                               \
      printf(format, __VA_ARGS__);                                               \
      ::Omega_h::check_fail(#cond, __FILE__, __LINE__);                          \
    }                                                                            \
  } while (0)
// the mean of the vectors (Omega_h_vector.hpp average: sum in index order, then one division)
template <int dim, int n>
__host__ __device__ inline Vector<dim> average(const Few<Vector<dim>, n>& x) {
  Vector<dim> avg = x[0];
  for (int i = 1; i < n; ++i) avg = avg + x[i];
  return avg / (Real)n;
}
template <int neev, class Arr>
__host__ __device__ inline Few<LO, neev> gather_verts(const Arr& ev2v, Int e) {
  Few<LO, neev> v;
  for (int i = 0; i < neev; ++i) v[i] = ev2v[e * neev + i];
  return v;
}
template <int neev, int dim, class Arr>
__host__ __device__ inline Matrix<dim, neev> gather_vectors(const Arr& a, const Few<LO, neev>& v) {
  Matrix<dim, neev> x;
  for (int i = 0; i < neev; ++i)
    for (int j = 0; j < dim; ++j) x[i][j] = a[v[i] * dim + j];
  return x;
}
template <int neev, class Arr>
__host__ __device__ inline Few<Real, neev> gather_scalars(const Arr& a, const Few<LO, neev>& v) {
  Few<Real, neev> x;
  for (int i = 0; i < neev; ++i) x[i] = a[v[i]];
  return x;
}

// ---- whole-array reductions (Omega_h_array_ops.hpp), on the library stream
template <class T>
inline T get_sum(Read<T> a) {
  T total = T();
  const T* p = a.data();
  Kokkos::parallel_reduce("get_sum", (long long)a.size(), [=] __host__ __device__(const int i, T& u) { u += p[i]; }, total);
  return total;
}
template <class T>
inline T get_max(Read<T> a) {  // (host fold of the copy: setup code of the drivers, not the step loop)
  pumipic::pp_check(pp_sync(), "get_max");
  const std::vector<T> h = a.to_host();
  T m = h.empty() ? T() : h[0];
  for (const T& v : h) m = v > m ? v : m;
  return m;
}
template <class T>
inline T get_min(Read<T> a) {
  pumipic::pp_check(pp_sync(), "get_min");
  const std::vector<T> h = a.to_host();
  T m = h.empty() ? T() : h[0];
  for (const T& v : h) m = v < m ? v : m;
  return m;
}
template <int dim>
struct BBox {
  Vector<dim> min, max;
};
template <int dim>
inline BBox<dim> get_bounding_box(Mesh* mesh) {
  pumipic::pp_check(pp_sync(), "get_bounding_box");
  const std::vector<Real> c = mesh->coords().to_host();
  const int md = mesh->dim();
  BBox<dim> b;
  for (int j = 0; j < dim; ++j) b.min[j] = b.max[j] = (j < md && !c.empty()) ? c[(size_t)j] : 0.0;
  for (size_t v = 0; v < c.size() / (size_t)md; ++v)
    for (int j = 0; j < dim && j < md; ++j) {
      const Real x = c[v * (size_t)md + (size_t)j];
      if (x < b.min[j]) b.min[j] = x;
      if (x > b.max[j]) b.max[j] = x;
    }
  return b;
}

// measure_elements_real (Omega_h_shape.hpp): triangle areas / tet volumes, the array the search divides by
inline Reals measure_elements_real(Mesh* mesh) { return mesh->elem_measures(); }

// ---- marks (Omega_h_mark.hpp)
inline Read<I8> mark_exposed_sides(Mesh* mesh) { return mesh->side_is_exposed(); }
// an entity of dimension `high` is marked when one of its `low`-dimensional bounding entities is
inline Read<I8> mark_up(Mesh* mesh, Int low, Int high, Read<I8> low_marked) {
  const Adj down = mesh->ask_down(high, low);
  const int nhigh = mesh->nents(high);
  const int deg = nhigh > 0 ? (int)(down.ab2b.size() / (size_t)nhigh) : 0;
  Write<I8> out((size_t)nhigh, (I8)0);
  const LO* d = down.ab2b.data();
  const I8* lm = low_marked.data();
  I8* o = out.data();
  parallel_for(nhigh, [=] __host__ __device__(LO h) {
    I8 m = 0;
    for (int k = 0; k < deg; ++k) m |= (I8)(lm[d[h * deg + k]] != 0);
    o[h] = m;
  }, "mark_up");
  return out;
}

// ---- Library / readers / writers
class Library {
 public:
  Library() {}
  Library(int*, char***) {}
  CommPtr world() const { return CommPtr(pumipic::comm_world()); }
  CommPtr self() const { return CommPtr(pumipic::comm_world()); }  // (a reader's comm argument is unused here)
};
// the mesh of a reader's arrays; boundary entities listed by the file become the sides' `class_id` (sides the file does
// not list: -1 -- Omega_h gives those the id of the region around them, which no driver here asks for)
inline Mesh mesh_from_data(pumipic::gmsh::MeshData& m) {
  Mesh mesh(m.dim, m.coords, m.elem2verts, m.class_id);
  if (!m.side_class.empty()) {
    const int d = m.dim, ns = mesh.nsides();
    pumipic::pp_check(pp_sync(), "side classification");
    const std::vector<int> s2v = mesh.ask_verts_of(d - 1).to_host();
    auto key = [d](const int* v) {
      int a[3] = {v[0], v[1], d == 3 ? v[2] : -1};
      std::sort(a, a + 3);
      return std::array<int, 3>{a[0], a[1], a[2]};
    };
    std::map<std::array<int, 3>, int> listed;
    for (size_t i = 0; i < m.side_class.size(); ++i) listed[key(&m.side_verts[i * (size_t)d])] = m.side_class[i];
    std::vector<int> cls((size_t)ns, -1);
    for (int sd = 0; sd < ns; ++sd) {
      auto it = listed.find(key(&s2v[(size_t)sd * (size_t)d]));
      if (it != listed.end()) cls[(size_t)sd] = it->second;
    }
    Write<LO> side_class((size_t)std::max(ns, 1));
    side_class.from_host(cls.data());
    mesh.set_tag(d - 1, "class_id", Write<LO>::wrap(side_class.data(), (size_t)ns));
    mesh.set_tag(d - 1, "class_id:storage", side_class);  // (keeps the allocation alive next to the exact-size view)
  }
  return mesh;
}
namespace gmsh {
inline Mesh read(const std::string& path, CommPtr) {
  pumipic::gmsh::MeshData m;
  std::string err;
  if (!pumipic::gmsh::read(path, m, &err)) {
    fprintf(stderr, "%s\n", err.c_str());
    exit(EXIT_FAILURE);
  }
  return mesh_from_data(m);
}
}  // namespace gmsh
namespace binary {
// The Omega_h binary format (.osh directories) is defined by Omega_h's sources, which are not in the reference
// tree: this reads the library's own flat container (magic "PPM1", dim, nverts, nelems, coords f64, elem2verts i32,
// class_id i32; pumi-pic_amd/meshio.py writes it) whatever the file is called.
inline Mesh read(const std::string& path, CommPtr, bool = false) {
  FILE* f = fopen(path.c_str(), "rb");
  int hdr[4] = {0, 0, 0, 0};
  pumipic::gmsh::MeshData m;
  bool ok = f && fread(hdr, sizeof(int), 4, f) == 4 && hdr[0] == 0x50504D31;
  if (ok) {  // (sizes a file of this length can hold: a garbled header must not become a huge allocation)
    fseek(f, 0, SEEK_END);
    const long long len = ftell(f);
    fseek(f, 4 * (long)sizeof(int), SEEK_SET);
    ok = (hdr[1] == 2 || hdr[1] == 3) && hdr[2] >= 0 && hdr[3] >= 0 &&
         16ll + 8ll * hdr[2] * hdr[1] + 4ll * hdr[3] * (hdr[1] + 2) <= len;
  }
  if (ok) {
    m.dim = hdr[1];
    m.coords.resize((size_t)hdr[2] * m.dim);
    m.elem2verts.resize((size_t)hdr[3] * (m.dim + 1));
    m.class_id.resize((size_t)hdr[3]);
    ok = fread(m.coords.data(), sizeof(double), m.coords.size(), f) == m.coords.size() &&
         fread(m.elem2verts.data(), sizeof(int), m.elem2verts.size(), f) == m.elem2verts.size() &&
         fread(m.class_id.data(), sizeof(int), m.class_id.size(), f) == m.class_id.size();
  }
  if (f) fclose(f);
  if (!ok) {
    fprintf(stderr, "%s: not a mesh container of this library (Omega_h .osh files cannot be read: the format is "
                    "not part of the reference tree)\n", path.c_str());
    exit(EXIT_FAILURE);
  }
  return mesh_from_data(m);
}
}  // namespace binary
// Omega_h::read_mesh_file(path, comm): the reader by extension (Omega_h_file.hpp)
inline Mesh read_mesh_file(const std::string& path, CommPtr comm) {
  const std::string ext = path.substr(path.find_last_of('.') + 1);
  return ext == "msh" ? gmsh::read(path, comm) : binary::read(path, comm);
}
namespace vtk {
// rendering is outside the hot path: the call is accepted and writes nothing
inline void write_parallel(const std::string&, Mesh*, Int = -1) {}
}  // namespace vtk
}  // namespace Omega_h

namespace pumipic {
using Omega_h::mark_up;  // (the drivers call it unqualified on a pumipic::Mesh*: argument-dependent lookup)
using Omega_h::mark_exposed_sides;
using Omega_h::measure_elements_real;  // (called unqualified on an o::Mesh*: test/test_adj.cpp:96)
// src/pumipic_library.hpp:8-18: starts the runtime (device = the launcher's LOCAL_RANK, PP_DEVICE overrides) and the
// process-wide communicator
class Library {
 public:
  Library(int* argc, char*** argv) : oh_lib(argc, argv) {
    const int device = getenv("PP_DEVICE") ? atoi(getenv("PP_DEVICE")) : (getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0);
    pp_check(pp_init(device), "pumipic::Library");
    (void)comm_world();
  }
  Omega_h::Library& omega_h_lib() { return oh_lib; }

 private:
  Omega_h::Library oh_lib;
};
// read(library, comm, prefix, &picparts) (src/pumipic_mesh.hpp:150-151, pumipic_file.cpp:100-190): the reference
// reads one .osh + one .ppm file per rank from <prefix>_<ranks>.ppm/.  Neither format can be pinned here (no
// reference-written file, no Omega_h sources), so `prefix` names a full mesh -- Gmsh .msh or the library's flat
// container -- and the parts are made on the spot: element blocks with the full mesh buffered (the replica of
// BASELINE's config 5), or PICparts from an Input when PP_PARTS=<buffer layers>:<safe layers> is set.
inline void read(Omega_h::Library* library, Omega_h::CommPtr comm, const char* prefix, Mesh* picparts) {
  {  // what pumipic::write(picparts, prefix) left (pumipic_adjacency.hpp): the parts are cut again as they were
    const std::string parts = std::string(prefix) + "_" + std::to_string(comm->size()) + ".pparts";
    if (FILE* f = fopen(parts.c_str(), "rb")) {
      fclose(f);
      read_parts_container(parts, comm, picparts);
      return;
    }
  }
  const std::string fn(prefix);
  const bool msh = fn.size() > 4 && fn.substr(fn.size() - 4) == ".msh";
  Mesh full = msh ? Omega_h::gmsh::read(fn, library->self()) : Omega_h::binary::read(fn, library->self());
  if (const char* spec = getenv("PP_PARTS")) {
    int buffer_layers = 3, safe_layers = 1;
    if (sscanf(spec, "%d:%d", &buffer_layers, &safe_layers) < 1 || buffer_layers < safe_layers) {
      fprintf(stderr, "PP_PARTS=<buffer layers>:<safe layers> with buffer >= safe\n");
      exit(EXIT_FAILURE);
    }
    const int ne = full.nelems(), world = comm->size();
    std::vector<int> owner((size_t)ne);
    for (int e = 0; e < ne; ++e) owner[(size_t)e] = (int)((long long)e * world / (ne > 0 ? ne : 1));
    Input input(full, Input::PARTITION, owner, Input::BFS, Input::BFS, comm);
    input.bufferBFSLayers = buffer_layers;
    input.safeBFSLayers = safe_layers;
    Mesh part(input);
    picparts->swap(part);
    picparts->keep_alive(std::move(full));  // (the part refers to the full mesh it was cut from)
  } else {
    full.partition(comm);
    picparts->swap(full);
  }
}
}  // namespace pumipic
#include "../pumipic_utils.hpp"
