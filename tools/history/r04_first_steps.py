#!/usr/bin/env python
"""Why are the first steps after an UNCACHED population build slow?  (c2mt: 17 ms per step over the first 5 timed
steps against 12.1 ms when the population came from the /tmp cache; BENCH_r03's cold pass: 3.49 against 0.67 ms.)
Prints the wall time of each of the first steps (stream synchronised after every step) under one variation:
  python tools/r04_first_steps.py <workload> <variant>
variants: plain | save (population generated and written to the cache) | nosave (population generated, not written to the cache) | sleep (cached + 10 s idle before the
steps) | churn (cached + 6 GB of numpy temporaries allocated, touched and freed before the steps) |
trim (generated + gc.collect() + malloc_trim(0) before the steps)"""
import ctypes
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pumipic_amd_loader  # noqa: E402

wl, variant = sys.argv[1], sys.argv[2]
if variant == "save":  # generated AND written to the cache, as the first bench.py run on a box does
    for f in os.listdir("/tmp"):
        if f.startswith("pp_pop_"):
            os.remove("/tmp/" + f)
if variant in ("nosave", "trim"):
    os.environ["PP_BENCH_CACHE"] = ""
    _savez = np.savez
    np.savez = lambda *a, **k: None  # generated, never written
    for f in os.listdir("/tmp"):
        if f.startswith("pp_pop_"):
            os.rename("/tmp/" + f, "/tmp/hidden_" + f)
pp = pumipic_amd_loader.load()
from pumipic_amd import capi  # noqa: E402
capi.init(0)
t0 = time.perf_counter()
w = bench.build_workload(pp, capi, wl, 10_000_000, 0, 1, 0.5)
w["origin_trust"] = False
st = bench.Stepper(pp, capi, w, wl, 0.5)
print("variant %s: set-up %.1f s" % (variant, time.perf_counter() - t0), flush=True)
if variant == "sleep":
    time.sleep(10)
if variant == "churn":
    for _ in range(3):
        a = np.random.default_rng(0).random(250_000_000)
        b = a * 2
        del a, b
if variant == "trim":
    gc.collect()
    ctypes.CDLL("libc.so.6").malloc_trim(0)
bench.clock_prewarm(capi, 0.3)
out = []
for i in range(14):
    t1 = time.perf_counter()
    st.step()
    capi.sync()
    out.append((time.perf_counter() - t1) * 1e3)
print(" ".join("%.2f" % x for x in out), flush=True)
if variant in ("nosave", "trim"):
    for f in os.listdir("/tmp"):
        if f.startswith("hidden_pp_pop_"):
            os.rename("/tmp/" + f, "/tmp/" + f[7:])
