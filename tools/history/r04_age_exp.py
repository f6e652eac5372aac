#!/usr/bin/env python
"""Does the c3 step get faster because the POPULATION ages or because the device settles?  100 steps on one
structure (wall clock per 10 steps), then a fresh structure in the same process, 40 steps."""
import os, sys, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import pumipic_amd_loader
pp = pumipic_amd_loader.load()
from pumipic_amd import capi
capi.init(0)
import gc
for label, nsteps in (("first structure", 100), ("fresh structure, same process", 40)):
    w = bench.build_workload(pp, capi, "c3", 10_000_000, 0, 1, 0.5)
    w["origin_trust"] = False
    st = bench.Stepper(pp, capi, w, "c3", 0.5)
    gc.collect(); gc.disable()
    for _ in range(3):
        st.step()
    capi.sync()
    out = []
    for blk in range(nsteps // 10):
        t0 = time.perf_counter()
        for _ in range(10):
            st.step()
        capi.sync()
        out.append(round((time.perf_counter() - t0) * 100, 4))
    gc.enable()
    print(label, "ms per step, in blocks of 10:", out, flush=True)
    del st, w
