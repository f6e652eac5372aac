import sys, os
sys.path.insert(0, os.getcwd())
import pumipic_amd_loader
pp = pumipic_amd_loader.load()
s = pp.synth
c, e, cl = s.kuhn_box(11)
s.write_mesh_bin("gpurun_out/cube.osh", 3, c, e, cl)
