// pp_geom.hpp -- the device geometry of the particle walk lives in the public include directory (the reference's
// device helpers of pumipic_adjacency.hpp / pumipic_utils.hpp are adapters over it: include/pumipic_utils.hpp), so that
// the library's kernels and a user's lambdas evaluate the SAME expressions.
#pragma once
#include "../include/pumipic_geom.hpp"
