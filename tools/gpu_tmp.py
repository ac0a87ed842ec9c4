import sys, os
import numpy as np
sys.path.insert(0, "nbody3d-webgpu_amd"); sys.path.insert(0, ".")
from nbody3d_amd import Simulation, ic
from oracle import oracle
n = 1000
b, v = ic.uniform_cube(n, seed=31)
ra = oracle.run_f64(b, v, None, 1e-3, 1.0, 1)[2]
for code in (2011, 2641, 4011, 8011, 2644):
    out = {}
    for kind in (400000, 200000):
        with Simulation(n, force_variant=kind + code, jsplit=1) as s:
            s.init(b, v); s.simulate(1, 1e-3, 1.0)
            out[kind] = s.read()[2]; name = s.variant
        err = np.abs(out[kind][:, :3] - ra[:, :3]).max() / np.abs(ra[:, :3]).max()
        print(code, name, "max rel err vs f64 oracle %.3e" % err)
    d = np.abs(out[400000] - out[200000]).max() / np.abs(ra[:, :3]).max()
    print("   fused vs two-kernel: %.3e, identical rows %d / %d" % (d, (out[400000] == out[200000]).all(1).sum(), n))
