// pseudoPushAndSearch driver on the MI355X-native particle_structs mirror.
//
// Follows test/pseudoPushAndSearch.cpp:87-119,228-338,444-542: particles start at the centroids
// of the elements that own a face on the model face (here: the mesh-container's box side
// y == ymin, the pumipic-data model files are not available), every iteration is
// push (USER lambda) -> legacy 3-D search_mesh (wall hits recorded in xpoints/xface) ->
// updatePtclPositions + rebuild (particles that left the domain are deleted) ->
// tagParentElements, for NUM_ITERATIONS = 30 or until no particle is left.
//
//   usage: pseudoPushAndSearch <mesh.bin> <numPtcls> <push dx> <push dy> <push dz>
#include <cmath>
#include <climits>
#include "../include/pumipic_adjacency.hpp"

#define NUM_ITERATIONS 30

using particle_structs::lid_t;
using particle_structs::MemberTypes;
using particle_structs::SellCSigma;
using pumipic::fp_t;
using pumipic::Vector3d;

typedef MemberTypes<Vector3d, Vector3d, int> Particle;  // position, next position, id
typedef ps::ParticleStructure<Particle> PS;

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

static void push(PS* ptcls, fp_t distance, fp_t dx, fp_t dy, fp_t dz) {
  p::Timer timer;
  auto position_d = ptcls->get<0>();
  auto new_position_d = ptcls->get<1>();
  o::Write<o::Real> ptclUnique_d((size_t)ptcls->capacity(), 0);
  auto lamb = PS_LAMBDA(const int&, const int& pid, const int& mask) {
    if (mask) {
      fp_t dir[3];
      dir[0] = distance * dx;
      dir[1] = distance * dy;
      dir[2] = distance * dz;
      new_position_d(pid, 0) = position_d(pid, 0) + dir[0] + ptclUnique_d[pid];
      new_position_d(pid, 1) = position_d(pid, 1) + dir[1] + ptclUnique_d[pid];
      new_position_d(pid, 2) = position_d(pid, 2) + dir[2] + ptclUnique_d[pid];
    }
  };
  ps::parallel_for(ptcls, lamb);
  p::RecordTime("ps push", timer.seconds());
}

static void updatePtclPositions(PS* ptcls) {
  auto x_ps_d = ptcls->get<0>();
  auto xtgt_ps_d = ptcls->get<1>();
  auto updatePtclPos = PS_LAMBDA(const int&, const int& pid, const int&) {
    for (int i = 0; i < 3; ++i) {
      x_ps_d(pid, i) = xtgt_ps_d(pid, i);
      xtgt_ps_d(pid, i) = 0;
    }
  };
  ps::parallel_for(ptcls, updatePtclPos);
}

static void tagParentElements(PS* ptcls, o::Write<o::LO> has_particles, int loop) {
  auto lamb = PS_LAMBDA(const int& e, const int&, const int& mask) {
    if (mask > 0) has_particles[e] = loop;
  };
  ps::parallel_for(ptcls, lamb);
}

int main(int argc, char** argv) {
  if (argc != 6) {
    fprintf(stderr, "Usage: %s <mesh.bin> <numPtcls> <push dx> <push dy> <push dz>\n", argv[0]);
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
  int numPtcls = atoi(argv[2]);
  fprintf(stderr, "number of elements %d number of particles %d\n", ne, numPtcls);

  // bounding box, model face = the y == ymin side
  double bbmin[3] = {1e300, 1e300, 1e300}, bbmax[3] = {-1e300, -1e300, -1e300};
  for (size_t v = 0; v < coords.size() / 3; ++v)
    for (int i = 0; i < 3; ++i) {
      bbmin[i] = std::min(bbmin[i], coords[3 * v + i]);
      bbmax[i] = std::max(bbmax[i], coords[3 * v + i]);
    }
  // setSourceElements (:228-273): elements with a face on the model face, equal counts, remainder
  // to the last one
  std::vector<lid_t> ppe_h(ne, 0);
  std::vector<int> marked;
  for (int e = 0; e < ne; ++e) {
    int on = 0;
    for (int i = 0; i < 4; ++i) on += std::fabs(coords[3 * (size_t)e2v[4 * (size_t)e + i] + 1] - bbmin[1]) < 1e-12;
    if (on >= 3) marked.push_back(e);
  }
  if (marked.empty()) {
    fprintf(stderr, "no element touches the model face\n");
    return EXIT_FAILURE;
  }
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
  const int sigma = INT_MAX;  // full sorting
  const int V = 1024;
  pumipic::TeamPolicy policy = pumipic::TeamPolicyAuto(10000, 32);
  PS* ptcls = new SellCSigma<Particle>(policy, sigma, V, ne, numPtcls, ptcls_per_elem, element_gids);

  {  // setInitialPtclCoords (:275-298): element centroids = average of the 4 vertices; setPtclIds
    auto cells2nodes = mesh.ask_elem_verts();
    auto nodes2coords = mesh.coords();
    auto x_ps_d = ptcls->get<0>();
    auto pid_d = ptcls->get<2>();
    auto lamb = PS_LAMBDA(const int& e, const int& pid, const int& mask) {
      pid_d(pid) = pid;
      if (mask > 0)
        for (int i = 0; i < 3; i++) {
          double c = nodes2coords[3 * cells2nodes[4 * e] + i];
          c = c + nodes2coords[3 * cells2nodes[4 * e + 1] + i];
          c = c + nodes2coords[3 * cells2nodes[4 * e + 2] + i];
          c = c + nodes2coords[3 * cells2nodes[4 * e + 3] + i];
          x_ps_d(pid, i) = c / 4;
        }
    };
    ps::parallel_for(ptcls, lamb);
  }

  double maxDimLen = 0;
  printf("bbox ");
  for (int i = 0; i < 3; i++) {
    printf("%3d %.3f %.3f ", i, bbmin[i], bbmax[i]);
    maxDimLen = std::max(maxDimLen, bbmax[i] - bbmin[i]);
  }
  printf("\n");
  const fp_t distance = maxDimLen / 20;
  const fp_t dx = atof(argv[3]), dy = atof(argv[4]), dz = atof(argv[5]);
  fprintf(stderr, "push distance %.3f push direction %.3f %.3f %.3f\n", distance, dx, dy, dz);

  o::Write<o::LO> has_particles((size_t)ne, -1);
  tagParentElements(ptcls, has_particles, 0);
  p::Timer fullTimer;
  long wall_hits = 0;
  int iter;
  for (iter = 1; iter <= NUM_ITERATIONS; iter++) {
    if (ptcls->nPtcls() == 0) {
      fprintf(stderr, "No particles remain... exiting push loop\n");
      break;
    }
    fprintf(stderr, "iter %d\n", iter);
    push(ptcls, distance, dx, dy, dz);
    // search(): legacy search_mesh with wall-hit output, then updatePtclPositions + rebuild
    p::Timer st;
    const size_t cap = (size_t)ptcls->capacity();
    o::Write<o::LO> elem_ids;  // empty: allocated and seeded with the parent elements
    o::Write<o::Real> xpoints_d(3 * cap, 0);
    o::Write<o::LO> xface_d(cap, -1);
    auto x = ptcls->get<0>();
    auto xtgt = ptcls->get<1>();
    auto pid = ptcls->get<2>();
    const bool isFound = p::search_mesh(mesh, ptcls, x, xtgt, pid, elem_ids, xpoints_d, xface_d, 100);
    if (!isFound) {
      fprintf(stderr, "search_mesh did not find every particle within the loop limit\n");
      return EXIT_FAILURE;
    }
    {
      std::vector<int> xf = xface_d.to_host();
      for (int f : xf) wall_hits += f >= 0;
    }
    updatePtclPositions(ptcls);
    ptcls->rebuild(elem_ids);
    fprintf(stderr, "search, rebuild, and transfer (seconds) %f\n", st.seconds());
    if (ptcls->nPtcls() == 0) {
      fprintf(stderr, "No particles remain... exiting push loop\n");
      break;
    }
    tagParentElements(ptcls, has_particles, iter);
  }
  p::fence();
  fprintf(stderr, "%d iterations of pseudopush (seconds) %f\n", iter, fullTimer.seconds());
  std::vector<int> hp = has_particles.to_host();
  int touched = 0;
  for (int v : hp) touched += v >= 0;
  printf("RESULT particles %d wall_hits %ld touched_elements %d iterations %d\n", ptcls->nPtcls(), wall_hits,
         touched, iter);
  delete ptcls;
  p::SummarizeTime();
  fprintf(stderr, "done\n");
  return 0;
}
