// pseudoXGCm driver on the MI355X-native particle_structs mirror.
//
// Follows the reference driver's step loop (test/pseudoXGCm.cpp:422-534): build the SCS from a
// Gaussian particles-per-element draw, place particles uniformly in their triangles, set up the
// elliptical push state, then per iteration  push -> search_mesh_2d -> updatePtclPositions +
// migrate/rebuild -> tagParentElements -> gyroScatter x2.  The push and the bookkeeping kernels
// are USER lambdas run through ps::parallel_for exactly as in the reference; search, rebuild
// and scatter go through the C-ABI.
//
//   usage: pseudoXGCm <mesh.bin | mesh.msh> <numPtcls> <max initial model face> <maxIterations>
//                     <degrees per elliptical push> <enable prebarrier>
// <mesh.bin> is the container written by pumi-pic_amd/synth.py:write_mesh_bin (the pumipic-data
// .osh/.ppm meshes of the reference are not available).
#include <cmath>
#include <random>
#include <iostream>
#include "../include/pumipic_adjacency.hpp"
#include "../include/pumipic_gmsh.hpp"

#define ELEMENT_SEED 1024 * 1024
#define PARTICLE_SEED 512 * 512

using particle_structs::lid_t;
using particle_structs::MemberTypes;
using particle_structs::SellCSigma;
using pumipic::fp_t;
using pumipic::Vector3d;

// positions now / after the push, particle id, ellipse semi-axis b, ellipse angle phi
typedef MemberTypes<Vector3d, Vector3d, int, float, float> Particle;
typedef ps::ParticleStructure<Particle> PS;

namespace ellipticalPush {
double h, k, d;
void setup(PS* ptcls, double h_in, double k_in, double d_in) {
  h = h_in;
  k = k_in;
  d = d_in;
  auto x_nm1 = ptcls->get<0>();
  auto ptcl_b = ptcls->get<3>();
  auto ptcl_phi = ptcls->get<4>();
  const double hd = h, kd = k, dd = d;
  auto setMajorAxis = PS_LAMBDA(const int&, const int& pid, const int& mask) {
    if (mask) {
      const double w = x_nm1(pid, 0), z = x_nm1(pid, 1);
      const double phi = atan2(dd * (z - kd), w - hd);
      ptcl_phi(pid) = (float)phi;
      ptcl_b(pid) = (float)((z - kd) / sin(phi));
    }
  };
  ps::parallel_for(ptcls, setMajorAxis);
}
void push(PS* ptcls, p::Mesh& m, double deg) {
  p::Timer timer;
  auto class_ids = m.class_ids();
  auto x_nm0 = ptcls->get<1>();
  auto ptcl_b = ptcls->get<3>();
  auto ptcl_phi = ptcls->get<4>();
  const double hd = h, kd = k, dd = d;
  auto setPosition = PS_LAMBDA(const int& e, const int& pid, const int& mask) {
    if (mask) {
      const double centerFactor = class_ids[e] == 1 ? 0.01 : 1.0;
      const double degP = deg * (centerFactor * 1.0 / class_ids[e]);
      const float phi = ptcl_phi(pid), b = ptcl_b(pid);
      const double rad = phi + degP * M_PI / 180.0;
      x_nm0(pid, 0) = (b * dd) * cos(rad) + hd;
      x_nm0(pid, 1) = b * sin(rad) + kd;
      ptcl_phi(pid) = (float)rad;
    }
  };
  ps::parallel_for(ptcls, setPosition);
  p::RecordTime("elliptical push", timer.seconds());
}
}  // namespace ellipticalPush

static bool readMesh(const char* fn, int& dim, std::vector<double>& coords, std::vector<int>& e2v,
                     std::vector<int>& cls) {
  // test/pseudoXGCm.cpp:306-324: the extension selects the reader ("msh" = Gmsh ASCII; the Omega_h
  // binary ".osh" format belongs to a library that is not in the reference tree)
  const std::string name(fn);
  if (name.size() > 4 && name.substr(name.size() - 4) == ".msh") {
    std::cout << "reading gmsh mesh " << name << "\n";
    pumipic::gmsh::MeshData m;
    std::string err;
    if (!pumipic::gmsh::read(name, m, &err)) {
      fprintf(stderr, "%s\n", err.c_str());
      return false;
    }
    dim = m.dim;
    coords.swap(m.coords);
    e2v.swap(m.elem2verts);
    cls.swap(m.class_id);
    return true;
  }
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

// Gaussian particles-per-element draw over elements with class_id <= mdlFace
static int setSourceElements(const std::vector<int>& cls, std::vector<lid_t>& ppe, int mdlFace,
                             int numPtcls) {
  const int ne = (int)cls.size();
  int numMarked = 0;
  for (int i = 0; i < ne; ++i) numMarked += cls[i] <= mdlFace;
  ppe.assign(ne, 0);
  if (!numMarked) return 0;
  const int nppe = numPtcls / numMarked;
  std::default_random_engine generator(ELEMENT_SEED);
  std::normal_distribution<double> dist(nppe, nppe / 4);
  int total = 0, last = -1;
  for (int i = 0; i < ne; ++i) {
    if (cls[i] <= mdlFace && total < numPtcls) {
      last = i;
      int n = (int)std::round(dist(generator));
      if (n < 0) n = 0;
      total += n;
      if (total > numPtcls) n -= total - numPtcls;
      ppe[i] = n;
    }
  }
  if (total < numPtcls) ppe[last] += numPtcls - total;
  int np = 0;
  for (int v : ppe) np += v;
  return np;
}

static void setInitialPtclCoords(p::Mesh& mesh, PS* ptcls) {
  const int cap = ptcls->capacity();
  std::vector<double> rnd(2 * (size_t)cap);
  std::default_random_engine generator(PARTICLE_SEED);
  std::uniform_real_distribution<double> dist(0.0, 1.0);
  for (int i = 0; i < cap; ++i) {
    double x = dist(generator), y = dist(generator);
    if (x + y > 1) {
      x = 1 - x;
      y = 1 - y;
    }
    rnd[2 * i] = x;
    rnd[2 * i + 1] = y;
  }
  o::Write<double> rand_nums(rnd.size());
  rand_nums.from_host(rnd.data());
  auto cells2nodes = mesh.ask_elem_verts();
  auto nodes2coords = mesh.coords();
  auto x_ps_d = ptcls->get<0>();
  auto pid_d = ptcls->get<2>();
  auto lamb = PS_LAMBDA(const int& e, const int& pid, const int& mask) {
    pid_d(pid) = pid;  // setPtclIds
    if (mask > 0) {
      const int v0 = cells2nodes[3 * e], v1 = cells2nodes[3 * e + 1], v2 = cells2nodes[3 * e + 2];
      const double r1 = rand_nums[2 * pid], r2 = rand_nums[2 * pid + 1];
      for (int i = 0; i < 2; i++)
        x_ps_d(pid, i) = nodes2coords[2 * v0 + i] + r1 * (nodes2coords[2 * v1 + i] - nodes2coords[2 * v0 + i]) +
                         r2 * (nodes2coords[2 * v2 + i] - nodes2coords[2 * v0 + i]);
      x_ps_d(pid, 2) = 0;
    }
  };
  ps::parallel_for(ptcls, lamb);
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
  if (argc != 7) {
    printf("numargs %d expected 7\nUsage: %s <mesh.bin> <numPtcls> <max initial model face> "
           "<maxIterations> <degrees per elliptical push> <enable prebarrier>\n", argc, argv[0]);
    return 1;
  }
  p::pp_check(pp_init(0), "pp_init");
  int dim = 0;
  std::vector<double> coords;
  std::vector<int> e2v, cls;
  if (!readMesh(argv[1], dim, coords, e2v, cls) || dim != 2) {
    fprintf(stderr, "cannot read a 2-D mesh container from %s\n", argv[1]);
    return EXIT_FAILURE;
  }
  p::Mesh mesh(dim, coords, e2v, cls);
  printf("Mesh loaded with <v e f> %d %d %d\n", mesh.nverts(), mesh.nsides(), mesh.nelems());
  const int ne = mesh.nelems();

  // gyro-average ring maps (gyroScatter.hpp:101-166)
  const double rmax = 0.038;
  const int numRings = 3, ptsPerRing = 8;
  const double theta = 0.0;
  o::Write<o::LO> forward_map((size_t)mesh.nverts() * numRings * ptsPerRing * 3);
  o::Write<o::LO> backward_map(forward_map.size());
  p::pp_check(pp_create_gyro_ring_mappings(mesh.handle(), rmax, numRings, ptsPerRing, theta,
                                           forward_map.data(), backward_map.data()),
              "createGyroRingMappings");

  const int numPtcls = atoi(argv[2]);
  const int mdlFace = atoi(argv[3]);
  const int maxIter = atoi(argv[4]);
  const double degPerPush = atof(argv[5]);
  std::vector<lid_t> ppe_h;
  const int actualParticles = setSourceElements(cls, ppe_h, mdlFace, numPtcls);
  fprintf(stderr, "particles created %d\nmax iterations: %d\n", actualParticles, maxIter);
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
  ps::SCS_Input<Particle> scs_input(policy, sigma, V, ne, actualParticles, ptcls_per_elem, element_gids);
  scs_input.padding_strat = ps::PAD_EVENLY;
  scs_input.shuffle_padding = 0.1;
  scs_input.extra_padding = 0;
  scs_input.name = "ps";
  PS* ptcls = new SellCSigma<Particle>(scs_input);
  setInitialPtclCoords(mesh, ptcls);

  const double h = 1.72479370 - .08, k = .020558260, d = 0.6;
  ellipticalPush::setup(ptcls, h, k, d);
  fprintf(stderr, "degrees per elliptical push %f\nellipse center %f %f ellipse ratio %.3f\n",
          degPerPush, h, k, d);

  o::Write<o::LO> has_particles((size_t)ne, -1);
  o::Write<o::Real> fwdTag((size_t)mesh.nverts()), bkwdTag((size_t)mesh.nverts()),
      syncTag(2 * (size_t)mesh.nverts());
  tagParentElements(ptcls, has_particles, 0);

  p::Timer fullTimer;
  int iter;
  for (iter = 1; iter <= maxIter; iter++) {
    if (iter == 1 || iter == maxIter) ptcls->printMetrics();
    const long totNp = ptcls->nPtcls();
    if (totNp == 0) {
      fprintf(stderr, "No particles remain... exiting push loop\n");
      break;
    }
    fprintf(stderr, "iter %d particles %ld\n", iter, totNp);
    ellipticalPush::push(ptcls, mesh, degPerPush);
    // search(): search_mesh_2d from every particle's own element, then rebuild
    o::Write<o::LO> elem_ids((size_t)ptcls->capacity(), -1);
    auto x = ptcls->get<0>();
    auto xtgt = ptcls->get<1>();
    auto pid = ptcls->get<2>();
    const bool isFound = p::search_mesh_2d(mesh, ptcls, x, xtgt, pid, elem_ids, 200);
    if (!isFound) {
      fprintf(stderr, "search_mesh_2d did not find every particle\n");
      return EXIT_FAILURE;
    }
    updatePtclPositions(ptcls);
    p::migrate_lb_ptcls(mesh, ptcls, elem_ids, 1.05);
    if (ptcls->nPtcls() == 0) {
      fprintf(stderr, "No particles remain... exiting push loop\n");
      break;
    }
    tagParentElements(ptcls, has_particles, iter);
    p::Timer st;
    p::pp_check(pp_gyro_scatter(mesh.handle(), ptcls->handle(), forward_map.data(), rmax, numRings,
                                ptsPerRing, fwdTag.data()), "gyroScatter fwd");
    p::pp_check(pp_gyro_scatter(mesh.handle(), ptcls->handle(), backward_map.data(), rmax, numRings,
                                ptsPerRing, bkwdTag.data()), "gyroScatter bkwd");
    p::pp_check(pp_gyro_sync_pack(mesh.nverts(), fwdTag.data(), bkwdTag.data(), syncTag.data()),
                "gyroSync");
    p::RecordTime("gyro scatter", st.seconds());
  }
  const double secs = fullTimer.seconds();
  fprintf(stderr, "%d iterations of pseudopush (seconds) %f\n", iter, secs);
  // summary line for the harness: particle count, scatter mass, last-touched element count
  std::vector<double> w = fwdTag.to_host();
  double mass = 0;
  for (double v : w) mass += v;
  std::vector<int> hp = has_particles.to_host();
  int touched = 0;
  for (int v : hp) touched += v >= 0;
  printf("RESULT particles %d scatter_mass %.17g touched_elements %d\n", ptcls->nPtcls(), mass, touched);
  delete ptcls;
  p::SummarizeTime();
  fprintf(stderr, "done\n");
  return 0;
}
