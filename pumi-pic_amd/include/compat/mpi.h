// compat/mpi.h -- NOT an MPI: the reference's tests include <mpi.h> by name (particle_structs/test/destroy_test.cpp:4);
// the handful of MPI names they use are in pumipic_mpi.hpp, over the library's communicator.
#pragma once
#include "../pumipic_mpi.hpp"
