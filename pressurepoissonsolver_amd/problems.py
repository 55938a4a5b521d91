"""Input generators for the drivers' analytic problems (host side, numpy).

Mirrors apps/3d/steady.cpp:221-292 (problem definitions) and apps/shared/Init.cpp:25-245
(cell-centre sampling; physical Dirichlet data folded into f as -2 g/h^2 on boundary cells,
Neumann data as +-g_n/h). Produces vectors in this build's patch order (Hierarchy.tables()).
"""
import numpy as np


def trig_exact(x, y, z):
    """apps/3d/steady.cpp:258-263"""
    x, y, z = x + .3, y + .3, z + .3
    return np.sin(np.pi * x) * np.cos(2.0 / 3 * np.pi * y) * np.sin(5.0 / 6 * np.pi * z)


def trig_rhs(x, y, z):
    """apps/3d/steady.cpp:252-257"""
    return -77.0 / 36 * np.pi ** 2 * trig_exact(x, y, z)


def gauss_exact(x, y, z):
    """apps/3d/steady.cpp:231-233"""
    return np.exp(np.cos(10 * np.pi * x)) - np.exp(np.cos(11 * np.pi * y)) + np.exp(np.cos(12 * np.pi * z))


def gauss_rhs(x, y, z):
    """apps/3d/steady.cpp:234-242"""
    c, e, s = np.cos, np.exp, np.sin
    pi = np.pi
    return -pi * pi * (100 * e(c(10 * pi * x)) * c(10 * pi * x) - 100 * e(c(10 * pi * x)) * s(10 * pi * x) ** 2
                       - 121 * e(c(11 * pi * y)) * c(11 * pi * y) + 121 * e(c(11 * pi * y)) * s(11 * pi * y) ** 2
                       + 144 * e(c(12 * pi * z)) * c(12 * pi * z) - 144 * e(c(12 * pi * z)) * s(12 * pi * z) ** 2)


def trig2d_exact(x, y):
    """apps/2d/steady.cpp:316"""
    return np.sin(np.pi * y) * np.cos(2 * np.pi * x)


def trig2d_rhs(x, y):
    """apps/2d/steady.cpp:314-315"""
    return -5 * np.pi ** 2 * trig2d_exact(x, y)


def trig_normal(x, y, z):
    """d/dx, d/dy, d/dz of trig_exact (apps/3d/steady.cpp:266-284)"""
    x, y, z = x + .3, y + .3, z + .3
    pi = np.pi
    return (pi * np.cos(pi * x) * np.cos(2.0 / 3 * pi * y) * np.sin(5.0 / 6 * pi * z),
            -2.0 / 3 * pi * np.sin(pi * x) * np.sin(2.0 / 3 * pi * y) * np.sin(5.0 / 6 * pi * z),
            5.0 / 6 * pi * np.sin(pi * x) * np.cos(2.0 / 3 * pi * y) * np.cos(5.0 / 6 * pi * z))


def gauss_normal(x, y, z):
    """apps/3d/steady.cpp:243-251"""
    pi = np.pi
    return (-10 * pi * np.sin(10 * pi * x) * np.exp(np.cos(10 * pi * x)) + 0 * y,
            11 * pi * np.sin(11 * pi * y) * np.exp(np.cos(11 * pi * y)) + 0 * x,
            -12 * pi * np.sin(12 * pi * z) * np.exp(np.cos(12 * pi * z)) + 0 * x)


PROBLEMS = {"trig": (trig_rhs, trig_exact), "gauss": (gauss_rhs, gauss_exact)}
NORMALS = {"trig": trig_normal, "gauss": gauss_normal}
PROBLEMS_2D = {"trig": (trig2d_rhs, trig2d_exact)}


def init_dirichlet_2d(tables, n, problem="trig", patches=None):
    """(f, exact) flat vectors for 2D levels, Init::initDirichlet2d (Init.cpp:304-361)."""
    ffun, efun = PROBLEMS_2D[problem] if isinstance(problem, str) else problem
    if patches is None:
        patches = np.arange(len(tables["id"]))
    starts, lengths = tables["starts"], tables["lengths"]
    idx = np.arange(n) + 0.5
    f = np.empty((len(patches), n, n))
    ex = np.empty_like(f)
    for k, p in enumerate(patches):
        h = lengths[p] / n
        Y, X = np.meshgrid(starts[p, 1] + h[1] * idx, starts[p, 0] + h[0] * idx, indexing="ij")
        f[k], ex[k] = ffun(X, Y), efun(X, Y)
        for s in range(4):
            if tables["nbr_kind"][p, s] != 0:
                continue
            ax, up = s // 2, s & 1
            sl = [slice(None)] * 2
            sl[1 - ax] = -1 if up else 0
            xb, yb = X[tuple(sl)].copy(), Y[tuple(sl)].copy()
            if ax == 0:
                xb += (0.5 if up else -0.5) * h[0]
            else:
                yb += (0.5 if up else -0.5) * h[1]
            f[k][tuple(sl)] -= 2 * efun(xb, yb) / h[ax] ** 2
    return f.ravel(), ex.ravel()


def cell_centres(tables, n, patches=None):
    """[P, n, n, n, 3] (z, y, x index order; last axis = x, y, z coordinate). Init.cpp:25-55"""
    starts, lengths = tables["starts"], tables["lengths"]
    if patches is None:
        patches = np.arange(len(tables["id"]))
    h = lengths[patches] / n
    idx = np.arange(n) + 0.5
    out = np.empty((len(patches), n, n, n, 3))
    for k, p in enumerate(patches):
        xs = [starts[p, a] + h[k, a] * idx for a in range(3)]
        Z, Y, X = np.meshgrid(xs[2], xs[1], xs[0], indexing="ij")
        out[k, ..., 0], out[k, ..., 1], out[k, ..., 2] = X, Y, Z
    return out


def init_dirichlet(tables, n, problem="trig", patches=None):
    """(f, exact) flat vectors, Init::initDirichlet (Init.cpp:152-245)."""
    ffun, efun = PROBLEMS[problem] if isinstance(problem, str) else problem
    if patches is None:
        patches = np.arange(len(tables["id"]))
    cc = cell_centres(tables, n, patches)
    f = ffun(cc[..., 0], cc[..., 1], cc[..., 2])
    ex = efun(cc[..., 0], cc[..., 1], cc[..., 2])
    h = tables["lengths"][patches] / n
    for k, p in enumerate(patches):
        for s in range(6):
            if tables["nbr_kind"][p, s] != 0:
                continue
            ax, up = s // 2, s & 1
            sl = [slice(None)] * 3
            sl[2 - ax] = -1 if up else 0
            c = cc[k][tuple(sl)].copy()
            c[..., ax] += (0.5 if up else -0.5) * h[k, ax]  # the boundary face itself
            f[k][tuple(sl)] -= 2 * efun(c[..., 0], c[..., 1], c[..., 2]) / h[k, ax] ** 2
    return f.ravel(), ex.ravel()


def init_neumann(tables, n, problem="trig", patches=None):
    """(f, exact) flat vectors for Neumann physical boundaries, Init::initNeumann (Init.cpp:56-150): the normal
    derivative on the boundary face enters the adjacent cells as +g_n/h on low sides, -g_n/h on high sides."""
    ffun, efun = PROBLEMS[problem]
    nfun = NORMALS[problem]
    if patches is None:
        patches = np.arange(len(tables["id"]))
    cc = cell_centres(tables, n, patches)
    f = ffun(cc[..., 0], cc[..., 1], cc[..., 2])
    ex = efun(cc[..., 0], cc[..., 1], cc[..., 2])
    h = tables["lengths"][patches] / n
    for k, p in enumerate(patches):
        for s in range(6):
            if tables["nbr_kind"][p, s] != 0:
                continue
            ax, up = s // 2, s & 1
            sl = [slice(None)] * 3
            sl[2 - ax] = -1 if up else 0
            c = cc[k][tuple(sl)].copy()
            c[..., ax] += (0.5 if up else -0.5) * h[k, ax]  # the boundary face itself (getXYZ with index -1 / n)
            gn = nfun(c[..., 0], c[..., 1], c[..., 2])[ax]
            f[k][tuple(sl)] += (-1.0 if up else 1.0) * gn / h[k, ax]
    return f.ravel(), ex.ravel()


def splitmix64_uniform(seed, count):
    """i.i.d. U(-1, 1) from splitmix64(seed) — mesh-order independent synthetic RHS."""
    x = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, count + 1, dtype=np.uint64))
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def random_rhs(ids, cells_per_patch, seed=0x5EED):
    """f ~ U(-1,1) keyed by tree node id (seed + id), as BASELINE.md's timing inputs."""
    out = np.empty((len(ids), cells_per_patch))
    for k, i in enumerate(ids):
        out[k] = splitmix64_uniform(seed + int(i), cells_per_patch)
    return out.ravel()
