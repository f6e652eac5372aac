// traceWallModel driver: trace_particle_through_mesh with a USER functor on the MI355X-native
// mirror headers (the `Func` template argument of adjacency.tpp:460-476).
//
// The functor below is a wall model on internal material interfaces: exposed sides behave as in
// RemoveParticleOnGeometricModelExit (tpp:617-639); an interior side between elements of
// different class_id stops the particle in the element it came from and records the side in
// inter_faces.  Its device code runs through ps::parallel_for between find_exit_face and
// set_new_element of every walk iteration, exactly where the reference calls `func` (tpp:563).
// Set-up as in pseudoPushAndSearch (particles at the centroids of the elements on the y == ymin
// side, test/pseudoPushAndSearch.cpp:228-298), one push, one traced search.
//
//   usage: traceWallModel <mesh.bin> <numPtcls> <distance> <dx> <dy> <dz> <requireIntersection>
#include <cmath>
#include <climits>
#include "../include/pumipic_adjacency.hpp"

using particle_structs::lid_t;
using particle_structs::MemberTypes;
using particle_structs::SellCSigma;
using pumipic::fp_t;
using pumipic::Vector3d;

typedef MemberTypes<Vector3d, Vector3d, int> Particle;  // position, next position, id
typedef ps::ParticleStructure<Particle> PS;
typedef decltype(((PS*)nullptr)->get<0>()) Seg3d;

static bool readMesh(const char* fn, int& dim, std::vector<double>& coords, std::vector<int>& e2v,
                     std::vector<int>& cls) {
  FILE* f = fopen(fn, "rb");
  if (!f) return false;
  int hdr[4];
  if (fread(hdr, sizeof(int), 4, f) != 4 || hdr[0] != 0x50504D31) {
    fclose(f);
    return false;
  }
  dim = hdr[1];
  coords.resize((size_t)hdr[2] * dim);
  e2v.resize((size_t)hdr[3] * (dim + 1));
  cls.resize((size_t)hdr[3]);
  bool ok = fread(coords.data(), sizeof(double), coords.size(), f) == coords.size() &&
            fread(e2v.data(), sizeof(int), e2v.size(), f) == e2v.size() &&
            fread(cls.data(), sizeof(int), cls.size(), f) == cls.size();
  fclose(f);
  return ok;
}


struct StopAtClassInterface {
  explicit StopAtClassInterface(bool requireIntersection) : requireIntersection_(requireIntersection) {}
  void operator()(p::Mesh& mesh, PS* ptcls, o::Write<o::LO>& elem_ids, o::Write<o::LO>& inter_faces,
                  o::Write<o::LO>& lastExit, o::Write<o::Real>&, o::Write<o::LO>& ptcl_done, Seg3d,
                  Seg3d) const {
    auto exposed = mesh.side_is_exposed();
    auto s2e_off = mesh.sides2elems_offsets();
    auto s2e = mesh.sides2elems();
    auto cls = mesh.class_ids();
    const bool ri = requireIntersection_;
    auto model = PS_LAMBDA(const int&, const int& pid, const int& mask) {
      if (mask > 0 && !ptcl_done[pid]) {
        const int bridge = lastExit[pid];
        const bool exp = exposed[bridge];
        bool iface = false;
        if (!exp) {
          const int first = s2e_off[bridge];
          iface = cls[s2e[first]] != cls[s2e[first + 1]];
        }
        ptcl_done[pid] = exp || iface;
        if (ri) {
          if (exp || iface) inter_faces[pid] = bridge;
        } else {
          if (exp) elem_ids[pid] = -1;
          if (iface) inter_faces[pid] = bridge;
        }
      }
    };
    ps::parallel_for(ptcls, model, "stopAtClassInterface");
  }
  bool requireIntersection_;
};

int main(int argc, char** argv) {
  if (argc != 8) {
    fprintf(stderr, "Usage: %s <mesh.bin> <numPtcls> <distance> <dx> <dy> <dz> <requireIntersection>\n", argv[0]);
    return EXIT_FAILURE;
  }
  p::pp_check(pp_init(0), "pp_init");
  int dim = 0;
  std::vector<double> coords;
  std::vector<int> e2v, cls;
  if (!readMesh(argv[1], dim, coords, e2v, cls) || dim != 3) {
    fprintf(stderr, "cannot read a 3-D mesh container from %s\n", argv[1]);
    return EXIT_FAILURE;
  }
  p::Mesh mesh(dim, coords, e2v, cls);
  const int ne = mesh.nelems();
  const int numPtcls = atoi(argv[2]);
  const fp_t distance = atof(argv[3]), dx = atof(argv[4]), dy = atof(argv[5]), dz = atof(argv[6]);
  const bool requireIntersection = atoi(argv[7]) != 0;
  double ymin = 1e300;
  for (size_t v = 0; v < coords.size() / 3; ++v) ymin = std::min(ymin, coords[3 * v + 1]);
  std::vector<lid_t> ppe_h(ne, 0);
  std::vector<int> marked;
  for (int e = 0; e < ne; ++e) {
    int on = 0;
    for (int i = 0; i < 4; ++i) on += std::fabs(coords[3 * (size_t)e2v[4 * (size_t)e + i] + 1] - ymin) < 1e-12;
    if (on >= 3) marked.push_back(e);
  }
  if (marked.empty()) return EXIT_FAILURE;
  for (int e : marked) ppe_h[e] = numPtcls / (int)marked.size();
  ppe_h[marked.back()] += numPtcls % (int)marked.size();
  PS::kkLidView ptcls_per_elem("ptcls_per_elem", ne);
  ptcls_per_elem.from_host(ppe_h.data());
  PS::kkGidView element_gids("element_gids", ne);
  {
    std::vector<pumipic::gid_t> g(ne);
    for (int i = 0; i < ne; ++i) g[i] = i;
    element_gids.from_host(g.data());
  }
  pumipic::TeamPolicy policy = pumipic::TeamPolicyAuto(10000, 32);
  PS* ptcls = new SellCSigma<Particle>(policy, INT_MAX, 1024, ne, numPtcls, ptcls_per_elem, element_gids);
  {
    auto cells2nodes = mesh.ask_elem_verts();
    auto nodes2coords = mesh.coords();
    auto x_ps_d = ptcls->get<0>();
    auto xt_ps_d = ptcls->get<1>();
    auto pid_d = ptcls->get<2>();
    auto lamb = PS_LAMBDA(const int& e, const int& pid, const int& mask) {
      pid_d(pid) = pid;
      if (mask > 0) {
        const fp_t dir[3] = {distance * dx, distance * dy, distance * dz};
        for (int i = 0; i < 3; i++) {
          double c = nodes2coords[3 * cells2nodes[4 * e] + i];
          c = c + nodes2coords[3 * cells2nodes[4 * e + 1] + i];
          c = c + nodes2coords[3 * cells2nodes[4 * e + 2] + i];
          c = c + nodes2coords[3 * cells2nodes[4 * e + 3] + i];
          x_ps_d(pid, i) = c / 4;
          xt_ps_d(pid, i) = x_ps_d(pid, i) + dir[i];
        }
      }
    };
    ps::parallel_for(ptcls, lamb);
  }
  auto x = ptcls->get<0>();
  auto xt = ptcls->get<1>();
  auto pids = ptcls->get<2>();
  o::Write<o::LO> elem_ids, xfaces((size_t)ptcls->capacity(), -1);
  o::Write<o::Real> xpoints(3 * (size_t)ptcls->capacity(), 0);
  StopAtClassInterface wall(requireIntersection);
  const bool found = p::trace_particle_through_mesh(mesh, ptcls, x, xt, pids, elem_ids,
                                                    requireIntersection, xfaces, xpoints, 200, false, wall);
  // the same call with the default functor must equal search_mesh
  o::Write<o::LO> ids_a, ids_b, f_a, f_b;
  o::Write<o::Real> p_a, p_b;
  p::RemoveParticleOnGeometricModelExit<Particle, Seg3d> dflt(mesh, requireIntersection);
  p::trace_particle_through_mesh(mesh, ptcls, x, xt, pids, ids_a, requireIntersection, f_a, p_a, 200, false, dflt);
  p::search_mesh(mesh, ptcls, x, xt, pids, ids_b, requireIntersection, f_b, p_b, 200);
  // statistics on the device (USER lambda with atomics)
  o::Write<unsigned long long> stats(6, 0);  // stopped, exposed hits, left, elem sum, face sum, mismatch
  {
    auto exposed = mesh.side_is_exposed();
    auto tally = PS_LAMBDA(const int&, const int& pid, const int& mask) {
      if (mask > 0) {
        const int f = xfaces[pid], e = elem_ids[pid];
        if (f >= 0 && !exposed[f]) atomicAdd(&stats[0], 1ull);
        if (f >= 0 && exposed[f]) atomicAdd(&stats[1], 1ull);
        if (e < 0) atomicAdd(&stats[2], 1ull);
        atomicAdd(&stats[3], (unsigned long long)(long long)e);
        atomicAdd(&stats[4], (unsigned long long)(long long)f);
        if (ids_a[pid] != ids_b[pid]) atomicAdd(&stats[5], 1ull);
      }
    };
    ps::parallel_for(ptcls, tally, "tally");
  }
  const std::vector<unsigned long long> h = stats.to_host();
  const long stopped = (long)h[0], exposed_hits = (long)h[1], left = (long)h[2], esum = (long)h[3],
             fsum = (long)h[4], mismatch = (long)h[5];
  printf("RESULT found %d stopped %ld exposed_hits %ld left %ld elem_sum %ld face_sum %ld default_mismatch %ld\n",
         (int)found, stopped, exposed_hits, left, esum, fsum, mismatch);
  delete ptcls;
  return 0;
}
