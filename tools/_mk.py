import sys, os
sys.path.insert(0, os.getcwd())
import pumipic_amd_loader
pp = pumipic_amd_loader.load()
s = pp.synth
n = int(sys.argv[1])
c, e, cl = s.kuhn_box(n)
s.write_mesh_bin("gpurun_out/cube%d.osh" % n, 3, c, e, cl)
