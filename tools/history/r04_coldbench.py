#!/usr/bin/env python
"""Round-3 verdict item 3, second half: bench.py's cold pass piece by piece in a fresh process -- W untimed steps,
the barrier (capi.sync + torch.cuda.synchronize), K steps with the sampled HIP events, the closing barrier --
each timed on the host, to find which piece holds the one-off ~50 ms.  Usage: python tools/r04_coldbench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pumipic_amd_loader  # noqa: E402
import torch  # noqa: E402

pp = pumipic_amd_loader.load()
from pumipic_amd import capi  # noqa: E402
capi.init(0)
t = time.perf_counter(); torch.cuda.synchronize(); print("first torch.cuda.synchronize %.2f ms" % ((time.perf_counter() - t) * 1e3))
w = bench.build_workload(pp, capi, "c3", 10_000_000, 0, 1, 0.5)
w["origin_trust"] = True
st = bench.Stepper(pp, capi, w, "c3", 0.5)
def T(label, f):
    t0 = time.perf_counter(); f(); print("%-40s %9.3f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True)
for i in range(5):
    T("warm-up step %d (enqueue only)" % i, st.step)
T("capi.sync", capi.sync)
T("torch.cuda.synchronize", torch.cuda.synchronize)
st.ntimed, st.kernel_ms, st.sample_every = 0, [], 2
t_all = time.perf_counter()
for i in range(20):
    T("timed step %d (enqueue only)" % i, lambda: st.step(timed=True))
T("capi.sync", capi.sync)
T("torch.cuda.synchronize", torch.cuda.synchronize)
print("20 steps incl. closing barrier: %.3f ms per step" % ((time.perf_counter() - t_all) * 1e3 / 20))
T("step_avg_ms (event elapsed)", st.step_avg_ms)
st.ntimed, st.kernel_ms = 0, []
t_all = time.perf_counter()
for i in range(20):
    st.step(timed=True)
capi.sync(); torch.cuda.synchronize()
print("second set of 20 steps: %.3f ms per step" % ((time.perf_counter() - t_all) * 1e3 / 20))
