import sys, os
sys.path.insert(0, os.getcwd())
import bench, pumipic_amd_loader
pp = pumipic_amd_loader.load()
from pumipic_amd import capi
capi.init(0)
class A: particles, deg, remainder, sigma = 10_000_000, 0.5, "last", 2**31-1
w = bench.build_workload(pp, capi, "c3", A.particles, 0, 1, A.deg)
st = bench.Stepper(pp, capi, w, "c3", A.deg)
for _ in range(30): st.step()
r = [bench.also_general_scatter(pp, capi, A, w, st)["ms_per_call"] for _ in range(3)]
print(os.environ.get("PUMIPIC_HIP_LIB","")[-22:], [round(x,4) for x in r])
