// team_policy.hpp -- particle_structs/src/team_policy.hpp:4-11 (TeamPolicyAuto) lives in particle_structs.hpp.
#pragma once
#include "particle_structs.hpp"
