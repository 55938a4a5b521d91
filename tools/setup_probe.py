import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from pressurepoissonsolver_amd import capi
for size in (512, 1024, 256):
    mesh = capi.Mesh.uniform(3, int(round(np.log2(size // 32))))
    H = capi.Hierarchy(mesh, 32)
    for k in range(3):
        t0 = time.perf_counter(); g = capi.GMG(H); t1 = time.perf_counter()
        print(size, k, round((t1 - t0) * 1e3, 1), g.setup_ms(), flush=True)
        del g
