// pumipic_ptcl_ops.hpp -- src/pumipic_ptcl_ops.hpp:32-85 (setUnsafeProcs, migrate_ptcls, migrate_lb_ptcls) live in
// pumipic_adjacency.hpp; printPtclImb (src/pumipic_lb.hpp:28,380-398) in pumipic_lb.hpp.
#pragma once
#include "pumipic_adjacency.hpp"
#include "pumipic_lb.hpp"
