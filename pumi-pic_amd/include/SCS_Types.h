// SCS_Types.h -- the reference spells this header name (particle_structs/src); everything it declares lives in
// particle_structs.hpp of this library.
#pragma once
#include "particle_structs.hpp"
