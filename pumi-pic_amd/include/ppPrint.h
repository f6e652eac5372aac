// ppPrint.h -- support/ppPrint.h:20-38 (printInfo / printError / setStdout / setStderr / getStdout / getStderr)
// live in particle_structs.hpp.
#pragma once
#include "particle_structs.hpp"
