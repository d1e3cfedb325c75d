import sys, numpy as np
sys.path.insert(0, '.')
import open_headstage_amd as ohs
from oracle import ohs_oracle as oracle
sys.path.insert(0, 'tests')
import test_gpu_ftz as T
bands = T._decay_bands(ohs)
x = np.zeros((2, 6000), np.float32)
x[0, 0], x[1, 3] = 3e-33, -7e-34
x[0, 3000:3010] = 1e-36
for nb in (1, 2, 3):
    g, o = T._run_pair(ohs, oracle, bands[:nb], x, 1)
    d = np.nonzero(g.view(np.uint32) != o.view(np.uint32))
    print("bands", nb, "differing samples", len(d[0]))
    if len(d[0]):
        c, i = d[0][0], d[1][0]
        print(" first at ch", c, "i", i)
        for k in range(max(0, i - 3), i + 4):
            print("   ", k, repr(g[c, k]), hex(g.view(np.uint32)[c, k]), repr(o[c, k]), hex(o.view(np.uint32)[c, k]))
# elementary check of x86 FTZ semantics through the oracle: exact tiny sums
