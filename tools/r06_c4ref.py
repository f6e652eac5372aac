#!/usr/bin/env python
"""ps_combo160 in the reference's two-loop shape, for profiling: tools/r05_c4ref.py <ne> <nptcl> [scs|csr] [K]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pumipic_amd_loader  # noqa: E402

pp = pumipic_amd_loader.load()
from pumipic_amd import capi  # noqa: E402

capi.init(0)
ne, n = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "scs"
k = int(sys.argv[4]) if len(sys.argv) > 4 else 20
print(json.dumps(bench.measure_c4ref(pp, capi, ne, n, kind, 1, k)))
