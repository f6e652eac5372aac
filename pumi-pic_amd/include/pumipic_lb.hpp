// pumipic_lb.hpp -- src/pumipic_lb.hpp: ParticleBalancer (:33-118) lives in pumipic_adjacency.hpp; here
// printPtclImb (:28, :380-398): rank 0 writes "Ptcl LB <max, min, avg, imb>: ..." to the library's stdout stream.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
#include "pumipic_mpi.hpp"
#include "pumipic_adjacency.hpp"
namespace pumipic {
// One host all-gather of the ranks' particle counts over the library's communicator (every rank calls it, as with the
// reference's three reductions); the average is the integer quotient total / ranks, as the reference prints it.
template <class PS>
void printPtclImb(PS* ptcls, MPI_Comm comm = MPI_COMM_WORLD) {
  const int ranks = pp_comm_size(comm);
  const int mine = ptcls->nPtcls();
  std::vector<int> counts((size_t)ranks, 0);
  pp_check(pp_comm_allgather_host(comm, &mine, counts.data(), sizeof(int)), "printPtclImb");
  if (pp_comm_rank(comm) != 0) return;
  const int most = *std::max_element(counts.begin(), counts.end());
  const int least = *std::min_element(counts.begin(), counts.end());
  const int total = std::accumulate(counts.begin(), counts.end(), 0);
  const float average = (float)(total / ranks);
  printInfo("Ptcl LB <max, min, avg, imb>: %d %d %.3f %.3f\n", most, least, average, (float)most / average);
}
}  // namespace pumipic
