// psAssert.h -- the older name of ppAssert.h (particle_structs/test/device_default_test.cpp:5)
#pragma once
#include "ppAssert.h"
