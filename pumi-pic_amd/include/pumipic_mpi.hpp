// pumipic_mpi.hpp -- the handful of MPI names the reference's drivers and public signatures spell
// (MPI_Comm in Distributor / pumipic_prebarrier / printPtclImb; MPI_Comm_rank/size, MPI_Allreduce, MPI_Reduce,
// MPI_Barrier on MPI_COMM_WORLD in test/pseudoXGCm.cpp:16-62,434-512, performance_tests/ps_combo160.cpp:17,64-67),
// over the library's communicator (pp_comm: RCCL over xGMI for device data, its TCP side channel for host
// scalars).  There is no MPI underneath: ranks are the processes the launcher started (RANK / WORLD_SIZE /
// MASTER_ADDR / MASTER_PORT), one per GPU.  Host scalars only -- device arrays go through pp_allreduce_sum,
// Mesh::reduceCommArray and ParticleStructure::migrate.
#pragma once
#include <cstring>
#include <vector>
#include "particle_structs.hpp"

typedef pp_comm* MPI_Comm;
#define MPI_COMM_WORLD (::pumipic::comm_world())
#define MPI_SUCCESS 0
enum MPI_Datatype { MPI_CHAR, MPI_INT, MPI_LONG, MPI_LONG_LONG, MPI_FLOAT, MPI_DOUBLE, MPI_UNSIGNED_LONG };
enum MPI_Op { MPI_SUM, MPI_MAX, MPI_MIN };

namespace pumipic {
namespace mpi_detail {
inline size_t size_of(MPI_Datatype t) {
  switch (t) {
    case MPI_CHAR: return 1;
    case MPI_INT: return sizeof(int);
    case MPI_FLOAT: return sizeof(float);
    case MPI_LONG: return sizeof(long);
    case MPI_UNSIGNED_LONG: return sizeof(unsigned long);
    case MPI_LONG_LONG: return sizeof(long long);
    case MPI_DOUBLE: return sizeof(double);
  }
  return 0;
}
template <class T>
void fold(const char* all, int nranks, int count, MPI_Op op, void* out) {
  T* o = (T*)out;
  for (int i = 0; i < count; ++i) {
    T acc;
    memcpy(&acc, all + sizeof(T) * (size_t)i, sizeof(T));
    for (int r = 1; r < nranks; ++r) {
      T v;
      memcpy(&v, all + sizeof(T) * ((size_t)r * count + i), sizeof(T));
      acc = op == MPI_SUM ? (T)(acc + v) : op == MPI_MAX ? (v > acc ? v : acc) : (v < acc ? v : acc);
    }
    o[i] = acc;
  }
}
}  // namespace mpi_detail
}  // namespace pumipic

inline int MPI_Init(int*, char***) {
  (void)::pumipic::comm_world();
  return MPI_SUCCESS;
}
inline int MPI_Finalize() { return MPI_SUCCESS; }
inline int MPI_Comm_rank(MPI_Comm c, int* rank) {
  *rank = pp_comm_rank(c);
  return MPI_SUCCESS;
}
inline int MPI_Comm_size(MPI_Comm c, int* size) {
  *size = pp_comm_size(c);
  return MPI_SUCCESS;
}
inline int MPI_Barrier(MPI_Comm c) {
  ::pumipic::pp_check(pp_comm_barrier(c), "MPI_Barrier");
  return MPI_SUCCESS;
}
// every rank's operands gathered over the host channel and folded in rank order on every rank (deterministic)
inline int MPI_Allreduce(const void* send, void* recv, int count, MPI_Datatype t, MPI_Op op, MPI_Comm c) {
  const size_t bytes = ::pumipic::mpi_detail::size_of(t) * (size_t)count;
  const int n = pp_comm_size(c);
  std::vector<char> all(bytes * (size_t)n);
  ::pumipic::pp_check(pp_comm_allgather_host(c, send, all.data(), (int)bytes), "MPI_Allreduce");
  switch (t) {
    case MPI_CHAR: ::pumipic::mpi_detail::fold<char>(all.data(), n, count, op, recv); break;
    case MPI_INT: ::pumipic::mpi_detail::fold<int>(all.data(), n, count, op, recv); break;
    case MPI_FLOAT: ::pumipic::mpi_detail::fold<float>(all.data(), n, count, op, recv); break;
    case MPI_LONG: ::pumipic::mpi_detail::fold<long>(all.data(), n, count, op, recv); break;
    case MPI_UNSIGNED_LONG: ::pumipic::mpi_detail::fold<unsigned long>(all.data(), n, count, op, recv); break;
    case MPI_LONG_LONG: ::pumipic::mpi_detail::fold<long long>(all.data(), n, count, op, recv); break;
    case MPI_DOUBLE: ::pumipic::mpi_detail::fold<double>(all.data(), n, count, op, recv); break;
  }
  return MPI_SUCCESS;
}
// (the result is defined on the root only; the other ranks' recv buffers are left untouched)
inline int MPI_Reduce(const void* send, void* recv, int count, MPI_Datatype t, MPI_Op op, int root, MPI_Comm c) {
  std::vector<char> tmp(::pumipic::mpi_detail::size_of(t) * (size_t)count);
  MPI_Allreduce(send, tmp.data(), count, t, op, c);
  if (pp_comm_rank(c) == root) memcpy(recv, tmp.data(), tmp.size());
  return MPI_SUCCESS;
}
