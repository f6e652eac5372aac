// pumipic_profiling.hpp -- src/pumipic_profiling.hpp:7-9: the optional barrier in front of a timed operation,
// global-scope spellings of pumipic::enable_prebarrier / pumipic::pumipic_prebarrier (pumipic_adjacency.hpp).
#pragma once
#include "pumipic_mpi.hpp"
#include "pumipic_adjacency.hpp"
inline void pumipic_enable_prebarrier() { ::pumipic::enable_prebarrier(); }
inline double pumipic_prebarrier(MPI_Comm mpi_comm) { return ::pumipic::pumipic_prebarrier(mpi_comm); }
