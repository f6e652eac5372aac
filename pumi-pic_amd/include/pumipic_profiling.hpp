// pumipic_profiling.hpp -- src/pumipic_profiling.hpp:7-9: the optional barrier in front of a timed operation,
// global-scope spellings of pumipic::enable_prebarrier / pumipic::pumipic_prebarrier (pumipic_adjacency.hpp).
// (MPI_Comm is pp_comm* in pumipic_mpi.hpp; spelled out here so that this header does not depend on that one's place in
// the include order.)
#pragma once
#include "pumipic_adjacency.hpp"
inline void pumipic_enable_prebarrier() { ::pumipic::enable_prebarrier(); }
inline double pumipic_prebarrier(pp_comm* mpi_comm) { return ::pumipic::pumipic_prebarrier(mpi_comm); }
