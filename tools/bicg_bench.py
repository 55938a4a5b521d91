import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pressurepoissonsolver_amd import capi, problems
if len(sys.argv) > 2 and sys.argv[1] == '--lib':  # another build of the library (same-box comparisons)
    capi.LIB_PATH = os.path.abspath(sys.argv[2])
n=32
mesh = capi.Mesh.uniform(3, 4)
H = capi.Hierarchy(mesh, n)
g = capi.GMG(H)
t = H.tables(0)
f, exact = problems.init_dirichlet(t, n)
for sm in (capi.SMOOTH_RBGS, capi.SMOOTH_PATCH_SOLVE):
    o = g.default_opts(smoother=sm)
    for rep in range(2):  # the first solve pays one-time costs (code load, lazy attributes)
        b = g.new_vector(0, f); x = g.new_vector(0)
        g.sync(); t0=time.time()
        its, rr = g.bicgstab(x, b, o, tol=1e-12)
        g.sync(); dt=time.time()-t0
    err = np.linalg.norm(x.download()-exact)/np.linalg.norm(exact)
    print('smoother',sm,'its',its,'rr %.2e'%rr,'time %.1f ms'%(dt*1e3),'per it %.2f ms'%(dt*1e3/max(its,1)),'err %.3e'%err)
for sm in (capi.SMOOTH_RBGS, capi.SMOOTH_PATCH_SOLVE):
    g.profile(True); g.profile_reset()
    b = g.new_vector(0, f); x = g.new_vector(0)
    g.sync(); t0=time.time()
    its, rr = g.bicgstab(x, b, g.default_opts(smoother=sm), tol=1e-12)
    g.sync(); dt=time.time()-t0
    rows = g.profile_rows()
    print('smoother', sm, 'second run: %.1f ms'%(dt*1e3), 'its', its)
    for k,v in sorted(rows.items(), key=lambda kv:-kv[1]['ms']): print('   ', k, v['calls'], round(v['ms'],2))
    g.profile(False)
