#!/usr/bin/env python
"""pp_gyro_scatter_radius on the aged c3 structure, 20 calls (for rocprofv3 --kernel-trace --stats)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pumipic_amd_loader  # noqa: E402

pp = pumipic_amd_loader.load()
from pumipic_amd import capi  # noqa: E402

capi.init(0)


class A:
    particles, deg, remainder, sigma = 10_000_000, 0.5, "last", 2**31 - 1


w = bench.build_workload(pp, capi, "c3", A.particles, 0, 1, A.deg)
st = bench.Stepper(pp, capi, w, "c3", A.deg)
for _ in range(30):
    st.step()
print(bench.also_general_scatter(pp, capi, A, w, st))
