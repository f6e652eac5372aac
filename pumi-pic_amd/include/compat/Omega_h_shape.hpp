// Omega_h_shape.hpp -- forwards to the one facade header (compat/Omega_h_mesh.hpp): NOT Omega_h.
#pragma once
#include "Omega_h_mesh.hpp"
