// pumipic_mesh.hpp -- src/pumipic_mesh.hpp:20-155: pumipic::Mesh (the PICpart) lives in pumipic_adjacency.hpp next to
// the searches that walk it; pumipic::Library (src/pumipic_library.hpp:8-18) and pumipic::read / write in
// compat/Omega_h_mesh.hpp, with the Omega_h names they take.
#pragma once
#include "pumipic_adjacency.hpp"
#include "pumipic_lb.hpp"
#include "compat/Omega_h_mesh.hpp"
