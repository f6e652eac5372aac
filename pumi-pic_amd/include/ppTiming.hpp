// ppTiming.hpp -- support/ppTiming.hpp:34-75 (RecordTime, SummarizeTime[AcrossProcesses], SetTimingVerbosity,
// EnableTiming, enable_prebarrier) live in pumipic_adjacency.hpp.
#pragma once
#include "pumipic_adjacency.hpp"
