// pseudoXGCm driver on the MI355X-native particle_structs mirror.
//
// The step loop, the helper functions and their signatures follow test/pseudoXGCm.cpp:102-534 of the
// reference: setSourceElements / setInitialPtclCoords / setPtclIds / updatePtclPositions / rebuild /
// search / tagParentElements are the reference's functions on the mirror types (user lambdas run
// through ps::parallel_for), ellipticalPush.hpp and gyroScatter.hpp are mirrored next to this file.
// What differs from the reference source: the includes, the mesh reader (the pumipic-data .osh / .ppm
// files are not available: a flat binary container or Gmsh .msh, then Mesh::partition instead of a
// .ppm partition file) and MPI, which is replaced by the launcher environment (RANK, WORLD_SIZE,
// LOCAL_RANK, MASTER_ADDR, MASTER_PORT; PP_COMM=rccl|tcp) behind pumipic::comm_world().
//
//   usage: pseudoXGCm <mesh.bin | mesh.msh> <numPtcls> <max initial model face> <maxIterations>
//                     <degrees per elliptical push> <enable prebarrier>
#include <cmath>
#include <iostream>
#include <random>
#include "ellipticalPush.hpp"
#include "gyroScatter.hpp"
#include "../include/pumipic_gmsh.hpp"

#define ELEMENT_SEED 1024 * 1024
#define PARTICLE_SEED 512 * 512

void updatePtclPositions(PS* ptcls) {
  auto x_ps_d = ptcls->get<0>();
  auto xtgt_ps_d = ptcls->get<1>();
  auto updatePtclPos = PS_LAMBDA(const int&, const int& pid, const int&) {
    x_ps_d(pid, 0) = xtgt_ps_d(pid, 0);
    x_ps_d(pid, 1) = xtgt_ps_d(pid, 1);
    x_ps_d(pid, 2) = xtgt_ps_d(pid, 2);
    xtgt_ps_d(pid, 0) = 0;
    xtgt_ps_d(pid, 1) = 0;
    xtgt_ps_d(pid, 2) = 0;
  };
  ps::parallel_for(ptcls, updatePtclPos);
}

void rebuild(p::Mesh& picparts, PS* ptcls, p::Distributor<>& dist, o::LOs elem_ids, const bool output) {
  (void)output;
  updatePtclPositions(ptcls);
  // (the reference's driver builds `dist` and its migrate_lb_ptcls then uses the world form; here the subset is
  // handed on, so that the migration is checked against it)
  pumipic::migrate_lb_ptcls(picparts, ptcls, elem_ids, 1.05, 0.5, &dist);
}

void search(p::Mesh& picparts, PS* ptcls, p::Distributor<>& dist, bool output) {
  o::Mesh* mesh = picparts.mesh();
  Omega_h::LO maxLoops = 200;
  const auto psCapacity = ptcls->capacity();
  o::Write<o::LO> elem_ids((size_t)psCapacity, -1);
  auto x = ptcls->get<0>();
  auto xtgt = ptcls->get<1>();
  auto pid = ptcls->get<2>();
  bool isFound = p::search_mesh_2d(*mesh, ptcls, x, xtgt, pid, elem_ids, maxLoops);
  if (!isFound) {  // assert(isFound) in the reference
    fprintf(stderr, "search_mesh_2d did not find every particle\n");
    exit(EXIT_FAILURE);
  }
  // rebuild the PS to set the new element-to-particle lists
  rebuild(picparts, ptcls, dist, elem_ids, output);
}

void setPtclIds(PS* ptcls, int id_offset) {
  auto pid_d = ptcls->get<2>();
  auto setIDs = PS_LAMBDA(const int&, const int& pid, const int&) { pid_d(pid) = id_offset + pid; };
  ps::parallel_for(ptcls, setIDs);
}

int setSourceElements(p::Mesh& picparts, PS::kkLidView ppe, const int mdlFace, const int numPtclsPerRank) {
  // Deterministically generate random number of particles on each element with classification less
  // than mdlFace (test/pseudoXGCm.cpp:167-222)
  int comm_rank = picparts.rank();
  const auto elm_dim = picparts.dim();
  o::Mesh* mesh = picparts.mesh();
  std::vector<int> face_class_ids = mesh->get_array<o::ClassId>(elm_dim, "class_id").to_host();
  std::vector<int> face_owners = picparts.entOwners(elm_dim).to_host();
  const int ne = mesh->nelems();
  std::vector<int> isFaceOnClass((size_t)ne, 0);
  int numMarked = 0;
  for (int i = 0; i < ne; ++i)
    if (face_class_ids[i] <= mdlFace && face_owners[i] == comm_rank) {
      isFaceOnClass[i] = 1;
      ++numMarked;
    }
  if (!numMarked) return 0;
  int nppe = numPtclsPerRank / numMarked;
  std::vector<int> rand_per_elem((size_t)ne, 0);
  std::default_random_engine generator(ELEMENT_SEED);
  std::normal_distribution<double> dist(nppe, nppe / 4);
  int total = 0, last = -1;
  for (int i = 0; i < ne; ++i) {
    if (isFaceOnClass[i] && total < numPtclsPerRank) {
      last = i;
      rand_per_elem[i] = (int)std::round(dist(generator));
      if (rand_per_elem[i] < 0) rand_per_elem[i] = 0;
      total += rand_per_elem[i];
      if (total > numPtclsPerRank) rand_per_elem[i] -= total - numPtclsPerRank;
    }
  }
  if (total < numPtclsPerRank) rand_per_elem[last] += numPtclsPerRank - total;
  int np = 0;
  for (int v : rand_per_elem) np += v;
  ppe.from_host(rand_per_elem.data());
  return np;
}

void setInitialPtclCoords(p::Mesh& picparts, PS* ptcls, bool output) {
  (void)output;
  // a particle is placed uniformly at random inside its parent triangle (test/pseudoXGCm.cpp:224-264)
  o::Mesh* mesh = picparts.mesh();
  const int cap = ptcls->capacity();
  std::vector<double> rnd(2 * (size_t)std::max(cap, 1));
  std::default_random_engine generator(PARTICLE_SEED);
  std::uniform_real_distribution<double> dist(0.0, 1.0);
  for (int i = 0; i < cap; ++i) {
    double x = dist(generator), y = dist(generator);
    if (x + y > 1) {
      x = 1 - x;
      y = 1 - y;
    }
    rnd[2 * (size_t)i] = x;
    rnd[2 * (size_t)i + 1] = y;
  }
  o::Write<double> rand_nums(rnd.size());
  rand_nums.from_host(rnd.data());
  auto cells2nodes = mesh->ask_elem_verts();
  auto nodes2coords = mesh->coords();
  auto x_ps_d = ptcls->get<0>();
  auto lamb = PS_LAMBDA(const int& e, const int& pid, const int& mask) {
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

void tagParentElements(p::Mesh& picparts, PS* ptcls, int loop) {
  // read from the tag, mark the elements that hold particles, write it back (test/pseudoXGCm.cpp:73-91)
  o::Mesh* mesh = picparts.mesh();
  o::Write<o::LO> ehp_nm1 = mesh->get_array<o::LO>(picparts.dim(), "has_particles");
  auto lamb = PS_LAMBDA(const int& e, const int&, const int& mask) {
    if (mask > 0) ehp_nm1[e] = loop;
  };
  ps::parallel_for(ptcls, lamb);
  mesh->set_tag(o::FACE, "has_particles", o::LOs(ehp_nm1));
}

// PP_DRIVER_DUMP=<prefix>: the structure as raw arrays (capacity entries each, slot order; x: three components
// `stride` apart) for the harness that replays the run on the CPU oracle (tests/test_gpu_driver.py)
template <class T>
static void dumpArray(const std::string& path, const T* dev, size_t n) {
  std::vector<T> h(n);
  if (n) p::pp_check(pp_memcpy_d2h(h.data(), dev, n * sizeof(T)), "dump");
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(h.data(), sizeof(T), n, f) != n) {
    fprintf(stderr, "cannot write %s\n", path.c_str());
    exit(EXIT_FAILURE);
  }
  fclose(f);
}
static void dumpState(const std::string& prefix, const char* tag, PS* ptcls, int rank) {
  const std::string base = prefix + "_r" + std::to_string(rank) + "_" + tag;
  const size_t cap = (size_t)ptcls->capacity();
  pp_ps_layout_t L;
  p::pp_check(pp_ps_layout(ptcls->handle(), &L), "pp_ps_layout");
  const pp_ps_info_t info = ptcls->info();
  auto x = ptcls->get<0>();
  auto id = ptcls->get<2>();
  auto b = ptcls->get<3>();
  auto phi = ptcls->get<4>();
  dumpArray(base + "_mask.u8", L.mask, cap);
  dumpArray(base + "_elem.i32", L.slot_elem, cap);
  dumpArray(base + "_x.f64", x.data(), (size_t)(2 * info.stride) + cap);
  dumpArray(base + "_id.i32", id.data(), cap);
  dumpArray(base + "_b.f32", b.data(), cap);
  dumpArray(base + "_phi.f32", phi.data(), cap);
  FILE* f = fopen((base + "_meta.txt").c_str(), "w");
  fprintf(f, "%zu %lld %d\n", cap, (long long)info.stride, ptcls->nPtcls());
  fclose(f);
}

static bool readMesh(const char* fn, int rank, int& dim, std::vector<double>& coords, std::vector<int>& e2v,
                     std::vector<int>& cls) {
  // test/pseudoXGCm.cpp:306-324: the extension selects the reader ("msh" = Gmsh ASCII; the Omega_h
  // binary ".osh" format belongs to a library that is not in the reference tree)
  const std::string name(fn);
  if (name.size() > 4 && name.substr(name.size() - 4) == ".msh") {
    if (!rank) std::cout << "reading gmsh mesh " << name << "\n";
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

int main(int argc, char** argv) {
  const int numargs = 7;
  if (argc != numargs) {
    printf("numargs %d expected %d\n", argc, numargs);
    auto args = "<mesh> <numPtcls> <max initial model face> <maxIterations> "
                "<degrees per elliptical push> <enable prebarrier>";
    std::cout << "Usage: " << argv[0] << " " << args << "\n";
    exit(1);
  }
  // one process per GPU: the launcher's LOCAL_RANK picks the device (PP_DEVICE overrides, e.g. to run
  // several ranks on one GPU over PP_COMM=tcp)
  const int device = getenv("PP_DEVICE") ? atoi(getenv("PP_DEVICE")) : (getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0);
  p::pp_check(pp_init(device), "pp_init");
  pp_comm* world = pumipic::comm_world();
  const int comm_rank = pp_comm_rank(world), comm_size = pp_comm_size(world);
  if (comm_rank == 0) {
    printf("world ranks %d (%s)\n", comm_size, pp_comm_kind(world));
    printf("particle_structs floating point value size (bits): %zu\n", sizeof(fp_t));
  }
  pumipic::SetTimingVerbosity(0);
  if (comm_rank == comm_size / 2) pumipic::EnableTiming();

  int dim = 0;
  std::vector<double> coords;
  std::vector<int> e2v, cls;
  if (!readMesh(argv[1], comm_rank, dim, coords, e2v, cls) || dim != 2) {
    fprintf(stderr, "cannot read a 2-D mesh container from %s\n", argv[1]);
    return EXIT_FAILURE;
  }
  // The parts.  Default: element blocks with the full mesh buffered and the own block as safe zone (the
  // replica BASELINE's config 5 names).  PP_PARTS=<buffer layers>:<safe layers>: PICparts from a
  // pumipic::Input as the reference builds them (test/pseudoXGCm.cpp:379-384) -- BFS buffer and safe zone,
  // every rank on its part's own mesh, elements travelling as global ids.
  p::Mesh full_mesh(dim, coords, e2v, cls);
  std::unique_ptr<p::Mesh> part_mesh;
  if (const char* spec = getenv("PP_PARTS")) {
    int buffer_layers = 3, safe_layers = 1;
    if (sscanf(spec, "%d:%d", &buffer_layers, &safe_layers) < 1 || buffer_layers < safe_layers) {
      fprintf(stderr, "PP_PARTS=<buffer layers>:<safe layers> with buffer >= safe\n");
      return EXIT_FAILURE;
    }
    const int ne_full = full_mesh.nelems();
    std::vector<int> owner((size_t)ne_full);
    for (int e = 0; e < ne_full; ++e) owner[(size_t)e] = (int)((long long)e * comm_size / (ne_full > 0 ? ne_full : 1));
    p::Input input(full_mesh, p::Input::PARTITION, owner, p::Input::BFS, p::Input::BFS, world);
    input.bufferBFSLayers = buffer_layers;
    input.safeBFSLayers = safe_layers;
    part_mesh.reset(new p::Mesh(input));
    if (!comm_rank) printf("PICparts from an Input: BFS buffer %d layers, safe zone %d layers\n", buffer_layers, safe_layers);
  } else {
    full_mesh.partition(world);  // element blocks, full mesh buffered, safe zone = own block
  }
  p::Mesh& picparts = part_mesh ? *part_mesh : full_mesh;
  o::Mesh* mesh = picparts.mesh();
  if (!comm_rank) printf("Mesh loaded with <v e f> %d %d %d\n", mesh->nverts(), mesh->nsides(), mesh->nelems());
  // the ranks this part exchanges particles with: itself + the buffered ranks (test/pseudoXGCm.cpp:390-396);
  // PP_DIST_SELF_ONLY=1 lists nobody else (a migration to another rank must then be refused: tests)
  std::vector<int> dist_ranks(1, comm_rank);
  if (!getenv("PP_DIST_SELF_ONLY"))
    for (int r : picparts.bufferedRanks(picparts.dim())) dist_ranks.push_back(r);
  p::Distributor<> dist((int)dist_ranks.size(), dist_ranks.data(), world);

  // Build gyro avg mappings
  const auto rmax = 0.038;
  const auto numRings = 3;
  const auto ptsPerRing = 8;
  const auto theta = 0.0;
  setGyroConfig(rmax, numRings, ptsPerRing, theta);
  if (!comm_rank) printGyroConfig();
  Omega_h::LOs forward_map;
  Omega_h::LOs backward_map;
  createGyroRingMappings(mesh, forward_map, backward_map);

  /* Particle data */
  const long int numPtcls = atol(argv[2]);
  const int numPtclsPerRank = (int)(numPtcls / comm_size);
  const bool output = numPtclsPerRank <= 30;
  int64_t totNumReqPtcls = numPtclsPerRank;
  p::pp_check(pp_allreduce_sum_host_i64(world, &totNumReqPtcls, 1), "MPI_Allreduce");
  if (!comm_rank) fprintf(stderr, "particles requested %ld %ld\n", numPtcls, (long)totNumReqPtcls);

  Omega_h::Int ne = mesh->nelems();
  PS::kkLidView ptcls_per_elem("ptcls_per_elem", ne);
  PS::kkGidView element_gids = picparts.globalIds(picparts.dim());
  const int mdlFace = atoi(argv[3]);
  int actualParticles = setSourceElements(picparts, ptcls_per_elem, mdlFace, numPtclsPerRank);
  int64_t totNumPtcls = actualParticles;
  p::pp_check(pp_allreduce_sum_host_i64(world, &totNumPtcls, 1), "MPI_Allreduce");
  if (!comm_rank) fprintf(stderr, "particles created %ld\n", (long)totNumPtcls);
  const auto maxIter = atoi(argv[4]);
  if (!comm_rank) fprintf(stderr, "max iterations: %d\n", maxIter);

  const int sigma = INT_MAX;  // full sorting
  const int V = 1024;
  pumipic::TeamPolicy policy = pumipic::TeamPolicyAuto(10000, 32);
  ps::SCS_Input<Particle> scs_input(policy, sigma, V, ne, actualParticles, ptcls_per_elem, element_gids);
  scs_input.padding_strat = ps::PAD_EVENLY;
  scs_input.shuffle_padding = 0.1;
  scs_input.extra_padding = 0;
  scs_input.name = "ps";
  ps::ParticleStructure<Particle>* ptcls = new SellCSigma<Particle>(scs_input);
  setInitialPtclCoords(picparts, ptcls, output);
  setPtclIds(ptcls, comm_rank * numPtclsPerRank);

  // define parameters controlling particle motion
  const double h = 1.72479370 - .08;
  const auto k = .020558260;
  const auto d = 0.6;
  ellipticalPush::setup(ptcls, h, k, d);
  const char* dump = getenv("PP_DRIVER_DUMP");
  if (dump) dumpState(dump, "initial", ptcls, comm_rank);
  const auto degPerPush = atof(argv[5]);
  if (!comm_rank) fprintf(stderr, "degrees per elliptical push %f\n", degPerPush);
  if (comm_rank == 0) fprintf(stderr, "ellipse center %f %f ellipse ratio %.3f\n", h, k, d);

  o::LOs elmTags((size_t)ne, -1);
  mesh->add_tag(o::FACE, "has_particles", 1, elmTags);
  const auto fwdTagName = "ptclToMeshScatterFwd";
  mesh->add_tag(o::VERT, fwdTagName, 1, o::Reals((size_t)mesh->nverts(), 0.0));
  const auto bkwdTagName = "ptclToMeshScatterBkwd";
  mesh->add_tag(o::VERT, bkwdTagName, 1, o::Reals((size_t)mesh->nverts(), 0.0));
  const auto syncTagName = "ptclToMeshSync";
  mesh->add_tag(o::VERT, syncTagName, 2, o::Reals((size_t)mesh->nverts() * 2, 0.0));
  tagParentElements(picparts, ptcls, 0);

  const auto enable_prebarrier = atoi(argv[6]);
  if (enable_prebarrier) {
    if (!comm_rank) fprintf(stderr, "pre-barrier enabled\n");
    pumipic::enable_prebarrier();
  }
  p::Timer fullTimer;
  int iter;
  int64_t totNp = 0;
  for (iter = 1; iter <= maxIter; iter++) {
    if ((!comm_rank || (comm_rank == comm_size / 2)) && (iter == 1 || iter == maxIter)) ptcls->printMetrics();
    totNp = ptcls->nPtcls();
    p::pp_check(pp_allreduce_sum_host_i64(world, &totNp, 1), "MPI_Allreduce");
    if (totNp == 0) {
      fprintf(stderr, "No particles remain... exiting push loop\n");
      break;
    }
    if (!comm_rank) fprintf(stderr, "iter %d particles %ld\n", iter, (long)totNp);
    static const bool trace = getenv("PP_DRIVER_TRACE") != nullptr;  // per-rank progress lines on stderr
#define PP_TRACE(what) \
  if (trace) fprintf(stderr, "[rank %d] iter %d: %s\n", comm_rank, iter, what)
    PP_TRACE("push");
    ellipticalPush::push(ptcls, *mesh, degPerPush, iter);
    PP_TRACE("barrier");
    p::pp_check(pp_comm_barrier(world), "MPI_Barrier");
    PP_TRACE("search");
    search(picparts, ptcls, dist, output);
    PP_TRACE("searched");
    totNp = ptcls->nPtcls();
    p::pp_check(pp_allreduce_sum_host_i64(world, &totNp, 1), "MPI_Allreduce");
    if (totNp == 0) {
      fprintf(stderr, "No particles remain... exiting push loop\n");
      break;
    }
    tagParentElements(picparts, ptcls, iter);
    PP_TRACE("scatter");
    gyroScatter(mesh, ptcls, forward_map, fwdTagName);
    gyroScatter(mesh, ptcls, backward_map, bkwdTagName);
    PP_TRACE("sync");
    gyroSync(picparts, fwdTagName, bkwdTagName, syncTagName);
    PP_TRACE("synced");
  }
  p::fence();  // (the reference's loop ends in gyroSync's MPI_Allreduce, which waits for the device)
  if (comm_rank == 0) fprintf(stderr, "%d iterations of pseudopush (seconds) %f\n", iter, fullTimer.seconds());

  // summary line for the harness: particle count over all ranks, mass of the SYNCED forward field (the
  // sum over ranks of the ranks' scatter fields), elements this rank ever saw particles in
  std::vector<double> w = mesh->get_array<o::Real>(0, syncTagName).to_host();
  double mass = 0;
  if (part_mesh) {  // every vertex counted once: by its owner; then summed over the ranks
    std::vector<int> vown = picparts.entOwners(0).to_host();
    for (size_t v = 0; v < vown.size(); ++v)
      if (vown[v] == comm_rank) mass += w[2 * v];
    o::Write<o::Real> m1(1, mass);
    p::pp_check(pp_allreduce_sum(world, m1.data(), 1), "MPI_Allreduce");
    mass = m1.to_host()[0];
  } else {
    for (size_t v = 0; v < w.size(); v += 2) mass += w[v];
  }
  if (dump) {
    dumpState(dump, "final", ptcls, comm_rank);
    const std::string base = std::string(dump) + "_r" + std::to_string(comm_rank) + "_final";
    dumpArray(base + "_fwd.f64", mesh->get_array<o::Real>(0, fwdTagName).data(), (size_t)mesh->nverts());
    dumpArray(base + "_bkwd.f64", mesh->get_array<o::Real>(0, bkwdTagName).data(), (size_t)mesh->nverts());
  }
  std::vector<int> hp = mesh->get_array<o::LO>(picparts.dim(), "has_particles").to_host();
  int64_t touched = 0;
  for (int v : hp) touched += v >= 0;
  int64_t np_all = ptcls->nPtcls();
  int64_t np_rank = np_all;
  p::pp_check(pp_allreduce_sum_host_i64(world, &np_all, 1), "MPI_Allreduce");
  printf("RANK %d particles %ld\n", comm_rank, (long)np_rank);
  if (comm_rank == 0)
    printf("RESULT particles %ld scatter_mass %.17g touched_elements %ld\n", (long)np_all, mass, (long)touched);
  delete ptcls;
  pumipic::SummarizeTimeAcrossProcesses();
  if (!comm_rank) fprintf(stderr, "done\n");
  pp_comm_destroy(world);
  return 0;
}
