import sys, os, time, json, numpy as np
sys.path.insert(0, os.getcwd())
import bench, pumipic_amd_loader
pp = pumipic_amd_loader.load()
from pumipic_amd import capi
capi.init(0)
class A: pass
a = A(); a.particles=10_000_000; a.deg=0.5; a.remainder="last"; a.sigma=2**31-1
w = bench.build_workload(pp, capi, "c3", a.particles, 0, 1, a.deg)
st = bench.Stepper(pp, capi, w, "c3", a.deg)
for _ in range(30): st.step()
g = bench.also_general_scatter(pp, capi, a, w, st)
m = bench.also_c2mt(pp, capi, a, w, st)
print(os.environ.get("PUMIPIC_HIP_LIB","default")[-22:], "scatter_ms", round(g["ms_per_call"],4), "c2mt_ms", round(m["ms_per_step"],3))
