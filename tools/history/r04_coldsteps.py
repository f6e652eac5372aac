#!/usr/bin/env python
"""Round-3 verdict item 3: where do the ~55 ms go that make bench.py's first W + K steps of c3 5x slower than the
steady state?  Builds the c3 workload in a fresh process and prints the host wall time of each of the first steps
(stream synchronised after every step), the structure's capacity, and -- with PP_ALLOC_DEBUG=1 -- every device
re-allocation the library makes.  Usage: PP_ALLOC_DEBUG=1 python tools/r04_coldsteps.py [nsteps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pumipic_amd_loader  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pp = pumipic_amd_loader.load()
from pumipic_amd import capi  # noqa: E402
capi.init(0)
w = bench.build_workload(pp, capi, "c3", 10_000_000, 0, 1, 0.5)
w["origin_trust"] = True
st = bench.Stepper(pp, capi, w, "c3", 0.5)
capi.sync()
print("step  wall_ms  capacity", flush=True)
for i in range(n):
    t0 = time.perf_counter()
    st.step()
    capi.sync()
    dt = (time.perf_counter() - t0) * 1e3
    sys.stderr.flush()
    print("%4d %8.3f %10d" % (i, dt, w["ps"].capacity()), flush=True)

# ---- second part (fresh process: run with argument "bench"): the sequence bench.py's cold pass runs --
# W untimed steps, barrier, K steps with sampled HIP events, barrier -- every piece timed on the host
