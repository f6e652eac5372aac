// compat/Omega_h_library.hpp -- see compat/Omega_h_mesh.hpp (not Omega_h: the names the reference drivers spell).
#pragma once
#include "Omega_h_mesh.hpp"
