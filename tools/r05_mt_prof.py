#!/usr/bin/env python
"""Where the intersection-mode walk's time goes (needs the instrumented build of k_search_mt3 -- a scratch patch, see
DESIGN.md "Round 5: what bounds the Moeller-Trumbore walk"): shader-clock cycles per wave in the refill, the record fetch,
the walk rounds and the start rounds, rounds and active lanes.   PUMIPIC_HIP_LIB=<instrumented .so> python tools/r05_mt_prof.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
import pumipic_amd_loader  # noqa: E402

pp = pumipic_amd_loader.load()
from pumipic_amd import capi  # noqa: E402

capi.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
w = bench.build_workload(pp, capi, "c2mt", n, 0, 1, 0.5)
st = bench.Stepper(pp, capi, w, "c2mt", 0.5)
for _ in range(3):
    st.step()
capi.sync()
out = (C.c_ulonglong * 16)()
lib = C.CDLL(os.environ["PUMIPIC_HIP_LIB"])
assert lib.pp_mt_prof(out) == 0
v = [int(x) for x in out]
visits, refill, fetch, walk, start, nwalk, nstart, awalk, astart, nrefill, total, waves = v[0], *v[2:13]
print("waves %d, visits %d (%.1f per particle)" % (waves, visits, visits / n))
print("cycles per wave: total %.0f = refill %.0f (%.1f%%) + fetch wait %.0f (%.1f%%) + walk rounds %.0f (%.1f%%) + start rounds %.0f (%.1f%%)"
      % (total / waves, refill / waves, 100 * refill / total, fetch / waves, 100 * fetch / total, walk / waves, 100 * walk / total,
         start / waves, 100 * start / total))
print("walk rounds per wave %.0f: %.1f active lanes, %.0f cycles of compute, fetch wait %.0f cycles per round (all rounds)"
      % (nwalk / waves, awalk / max(nwalk, 1), walk / max(nwalk, 1), fetch / max(nwalk + nstart, 1)))
print("start rounds per wave %.0f: %.1f active lanes, %.0f cycles of compute; refills (> 200 cycles) %.0f per wave, %.0f cycles each"
      % (nstart / waves, astart / max(nstart, 1), start / max(nstart, 1), nrefill / waves, refill / max(nrefill, 1)))
