// pumipic_lb.hpp -- src/pumipic_lb.hpp: ParticleBalancer (:33-118) lives in pumipic_adjacency.hpp; here
// printPtclImb (:28, :380-398).
#pragma once
#include "pumipic_mpi.hpp"
#include "pumipic_adjacency.hpp"
namespace pumipic {
// Print particle imbalance statistics: rank 0 writes "<max, min, avg, imb>" of the particle counts to the library's
// stdout stream (printInfo)
template <class PS>
void printPtclImb(PS* ptcls, MPI_Comm comm = MPI_COMM_WORLD) {
  int np = ptcls->nPtcls();
  int min_p = 0, max_p = 0, tot_p = 0;
  MPI_Reduce(&np, &min_p, 1, MPI_INT, MPI_MIN, 0, comm);
  MPI_Reduce(&np, &max_p, 1, MPI_INT, MPI_MAX, 0, comm);
  MPI_Reduce(&np, &tot_p, 1, MPI_INT, MPI_SUM, 0, comm);
  int comm_rank, comm_size;
  MPI_Comm_rank(comm, &comm_rank);
  MPI_Comm_size(comm, &comm_size);
  if (comm_rank == 0) {
    float avg = tot_p / comm_size;  // (integer division, as the reference's)
    float imb = max_p / avg;
    printInfo("Ptcl LB <max, min, avg, imb>: %d %d %.3f %.3f\n", max_p, min_p, avg, imb);
  }
}
}  // namespace pumipic
