// The checks of the reference's test/test_comm_array.cpp (minOwnership, sumEntities, fullBufferTest,
// the max reduction, the owned-element sum) and of test/test_lb.cpp's testBalanceArray written against the mirror: PICparts from a pumipic::Input,
// comm arrays from createCommArray, Mesh::reduceCommArray through the owners (pp_picpart_reduce).  Runs as
// one rank or as several rank processes (PP_COMM=tcp, RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
//   comm_array <mesh.bin> <partition file: one owner per element> [buffer layers] [safe layers]
// Entity dimensions: 0..dim as the reference loops them (edges of tets included, pp_mesh_num_edges).
#include <climits>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <vector>

#include "../include/pumipic_adjacency.hpp"

namespace p = pumipic;

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

static bool anySet(o::Write<o::LO> flag) {
  o::HostWrite<o::LO> h(flag);
  return h[0] != 0;
}

// every entity ends with its owner after a MIN over "my rank where I own it, INT_MAX elsewhere"
static bool minOwnership(p::Mesh& picparts, int dim) {
  const int rank = picparts.rank();
  o::Write<o::LO> owner_comm = picparts.createCommArray(dim, 1, INT_MAX);
  o::LOs ent_owners = picparts.entOwners(dim);
  o::parallel_for(picparts.nents(dim), OMEGA_H_LAMBDA(o::LO id) {
    if (ent_owners[id] == rank) owner_comm[id] = rank;
  });
  picparts.reduceCommArray(dim, p::Mesh::MIN_OP, owner_comm);
  o::Write<o::LO> fail(1, 0);
  o::parallel_for(picparts.nents(dim), OMEGA_H_LAMBDA(o::LO id) {
    if (owner_comm[id] != ent_owners[id]) fail[0] = 1;
  });
  return !anySet(fail);
}

// how many parts hold each entity, then 1/that summed over the parts is 1
static bool sumEntities(p::Mesh& picparts, int dim) {
  o::Write<o::LO> sum_comm = picparts.createCommArray(dim, 1, 1);
  picparts.reduceCommArray(dim, p::Mesh::SUM_OP, sum_comm);
  o::Write<o::Real> contribution = picparts.createCommArray(dim, 3, 0.0);
  o::parallel_for(picparts.nents(dim), OMEGA_H_LAMBDA(o::LO id) {
    for (int i = 0; i < 3; ++i) contribution[id * 3 + i] = 1.0 / sum_comm[id];
  });
  picparts.reduceCommArray(dim, p::Mesh::SUM_OP, contribution);
  o::Write<o::LO> fail(1, 0);
  o::parallel_for(picparts.nents(dim), OMEGA_H_LAMBDA(o::LO id) {
    for (int i = 0; i < 3; ++i)
      if (fabs(contribution[id * 3 + i] - 1.0) > .00001) fail[0] = 1;
  });
  return !anySet(fail);
}

// FULL buffer and safe zone: a SUM of ones counts every rank on every entity
static bool fullBufferTest(p::Mesh& mesh, const std::vector<int>& owner, int dim) {
  p::Input input(mesh, p::Input::PARTITION, owner, p::Input::FULL, p::Input::FULL);
  p::Mesh picparts(input);
  o::Write<o::LO> comm_arr = picparts.createCommArray(dim, 1, 1);
  picparts.reduceCommArray(dim, p::Mesh::SUM_OP, comm_arr);
  const int comm_size = picparts.num_ranks();
  o::Write<o::LO> fail(1, 0);
  o::parallel_for(picparts.nents(dim), OMEGA_H_LAMBDA(o::LO id) {
    if (comm_arr[id] != comm_size) fail[0] = 1;
  });
  return !anySet(fail) && picparts.isFullMesh() && picparts.nents(dim) == mesh.nents(dim);
}

// testBalanceArray of the reference's test/test_lb.cpp:77-133: (rank + 1) * 50 particles per element, one
// ParticleBalancer::partition, the particles every rank would hold afterwards -> imbalance <= 1.3
static bool balanceArray(p::Mesh& picparts, p::ParticleBalancer& balancer) {
  const int rank = picparts.rank(), comm_size = picparts.num_ranks();
  const o::LO ne = picparts.nelems();
  const int per_elem = (rank + 1) * 50;
  o::Write<o::LO> ptcls_per_elem((size_t)ne, per_elem);
  auto new_procs = balancer.partition(picparts, ptcls_per_elem, 1.05);
  o::Write<o::LO> send_ptcls((size_t)comm_size, 0);
  o::parallel_for((o::LO)new_procs.size(), OMEGA_H_LAMBDA(o::LO ptcl) { atomicAdd(&send_ptcls[new_procs[ptcl]], 1); });
  o::HostWrite<o::LO> send_host(send_ptcls);
  std::vector<int64_t> per_rank((size_t)comm_size);
  for (int i = 0; i < comm_size; ++i) per_rank[(size_t)i] = send_host[(size_t)i];
  p::pp_check(pp_allreduce_sum_host_i64(picparts.comm(), per_rank.data(), comm_size), "allreduce");
  int64_t total = 0, mx = 0;
  for (int64_t v : per_rank) {
    total += v;
    mx = v > mx ? v : mx;
  }
  const double imb = mx / (total * 1.0 / comm_size);
  if (!rank) fprintf(stderr, "Imbalance after balancing is %f\n", imb);
  return comm_size == 1 || imb <= 1.3;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "Usage: %s <mesh.bin> <partition filename> [buffer layers] [safe layers]\n", argv[0]);
    return EXIT_FAILURE;
  }
  pp_comm* world = p::comm_world();
  const int rank = pp_comm_rank(world);
  int dim;
  std::vector<double> coords;
  std::vector<int> e2v, cls;
  if (!readMesh(argv[1], dim, coords, e2v, cls)) {
    fprintf(stderr, "cannot read mesh %s\n", argv[1]);
    return EXIT_FAILURE;
  }
  p::Mesh mesh(dim, coords, e2v, cls);  // the full mesh, loaded in serial everywhere
  if (rank == 0) printf("Mesh loaded with <v s r> %d %d %d\n", mesh.nverts(), mesh.nsides(), mesh.nelems());
  std::vector<int> owner;
  {
    std::ifstream in_str(argv[2]);
    if (!in_str) {
      if (!rank) fprintf(stderr, "Cannot open file %s\n", argv[2]);
      return EXIT_FAILURE;
    }
    int own;
    while (in_str >> own) owner.push_back(own);
    if ((int)owner.size() != mesh.nelems()) {
      fprintf(stderr, "partition file holds %zu owners for %d elements\n", owner.size(), mesh.nelems());
      return EXIT_FAILURE;
    }
  }
  const int buffer_layers = argc > 3 ? atoi(argv[3]) : 1, safe_layers = argc > 4 ? atoi(argv[4]) : 0;
  int fails = 0;
  // every entity dimension 0..dim, as the reference loops them (test/test_comm_array.cpp:48-66)
  for (int d = 0; d <= dim; ++d)
    if (!fullBufferTest(mesh, owner, d)) {
      printf("fullBufferTest on dimension %d failed on rank %d\n", d, rank);
      ++fails;
    }
  // ---- the parts: core + `buffer_layers` layers of whole parts, safe zone `safe_layers` layers
  p::Mesh picparts(mesh, owner, buffer_layers, safe_layers);  // (test_comm_array.cpp:57: picparts(mesh, owner, 1, 0))
  for (int d = 0; d <= dim; ++d)
    if (!minOwnership(picparts, d)) {
      printf("minOwnership on dimension %d failed on rank %d\n", d, rank);
      ++fails;
    }
  if (!sumEntities(picparts, 0)) {
    printf("sumEntities on dimension 0 failed on rank %d\n", rank);
    ++fails;
  }
  {  // a MAX never lowers a value
    o::Write<o::Real> max_comm = picparts.createCommArray(0, 1, 0.0);
    o::parallel_for(picparts.nents(0), OMEGA_H_LAMBDA(o::LO v) { max_comm[v] = v; });
    picparts.reduceCommArray(0, p::Mesh::MAX_OP, max_comm);
    o::Write<o::LO> fail(1, 0);
    o::parallel_for(picparts.nents(0), OMEGA_H_LAMBDA(o::LO v) {
      if (max_comm[v] < v) fail[0] = 1;
    });
    if (anySet(fail)) {
      fprintf(stderr, "Max reduce failed on %d\n", rank);
      ++fails;
    }
  }
  {  // three values per element, 1 on the owner only: the SUM is 1 everywhere
    o::Write<o::LO> comm_array = picparts.createCommArray(dim, 3, 0);
    o::LOs owners = picparts.entOwners(dim);
    o::parallel_for(picparts.nelems(), OMEGA_H_LAMBDA(o::LO e) {
      for (int i = 0; i < 3; ++i) comm_array[e * 3 + i] = (owners[e] == rank);
    });
    picparts.reduceCommArray(dim, p::Mesh::SUM_OP, comm_array);
    o::HostWrite<o::LO> host_array(comm_array);
    bool success = true;
    for (size_t i = 0; i < host_array.size(); ++i) success = success && host_array[i] == 1;
    if (!success) {
      fprintf(stderr, "Multielement comm operation failed on %d\n", rank);
      ++fails;
    }
  }
  {  // the balancer on parts with the reference test's Input (BFS buffer, FULL safe zone; test_lb.cpp:61-66)
    p::Input lb_input(mesh, p::Input::PARTITION, owner, p::Input::BFS, p::Input::FULL);
    p::Mesh lb_parts(lb_input);
    p::ParticleBalancer balancer(lb_parts);
    if (balancer.getSbarIDs(lb_parts).size() != (size_t)lb_parts.nelems() || !balanceArray(lb_parts, balancer)) {
      fprintf(stderr, "balanceArray failed on %d\n", rank);
      ++fails;
    }
  }
  printf("rank %d: part <v r> %d %d of %d %d, buffers %d, %s\n", rank, picparts.nents(0), picparts.nelems(),
         mesh.nverts(), mesh.nelems(), picparts.numBuffers(dim), fails ? "FAILED" : "all checks passed");
  pp_comm_barrier(world);
  return fails;
}
