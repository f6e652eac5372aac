// pumipic_constants.hpp -- the reference keeps these declarations in a header of their own (src/pumipic_constants.hpp); here they live in
// pumipic_adjacency.hpp / compat/Omega_h_mesh.hpp.
#pragma once
#include "pumipic_adjacency.hpp"
