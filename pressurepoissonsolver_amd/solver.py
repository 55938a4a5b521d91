"""BiCGStab<D>::solve (src/Thunderegg/BiCGStab.h:45-106) for multi-rank runs.

`bicgstab` is the native te_bicgstab: on a sharded hierarchy the scalars of an iteration are summed over the
ranks (Vector.h:294,319 MPI_Allreduce) by the library's own RCCL communicator or by the registered all-reduce
callback (dist.attach / LocalFabric.attach register one), 1 + 2 + 2 doubles per iteration, no Python in the loop.
`bicgstab_host` is the statement-by-statement host mirror over te_vec_* calls (one reduction per scalar): the
independent second statement the tests compare the native solver with.
"""
import math


def bicgstab(gmg, x, b, opts=None, max_it=1000, tol=1e-12, allreduce=None):
    """Returns (iterations, final relative residual); `allreduce(values, op)` replaces the registered callback."""
    if allreduce is not None:
        gmg.set_allreduce(lambda vals, op: allreduce(vals) if op == 0 else allreduce(vals, op))
    return gmg.bicgstab(x, b, opts, max_it=max_it, tol=tol)


def bicgstab_host(gmg, x, b, opts=None, max_it=1000, tol=1e-12, allreduce=None):
    """Returns (iterations, final relative residual). `opts` = GMG cycle options used as the right
    preconditioner Mr (None: unpreconditioned). Statement order follows BiCGStab.h."""
    red = (lambda v: list(v)) if allreduce is None else allreduce
    new = lambda: gmg.new_vector(0)  # noqa: E731
    resid, rhat, p, ap, as_, s = new(), new(), new(), new(), new(), new()
    ms = mp = None
    if opts is not None:
        ms, mp = new(), new()
    gmg.apply(x, resid)
    resid.scaleThenAdd(-1, b)
    r0_norm = math.sqrt(red([resid.twoNormSqLocal()])[0])
    rhat.copy(resid)
    p.copy(resid)
    rho = red([rhat.dot(resid)])[0]
    its, rnorm = 0, r0_norm
    while rnorm / r0_norm > tol and its < max_it:
        if opts is not None:
            gmg.cycle(opts, p, mp)
            gmg.apply(mp, ap)
        else:
            gmg.apply(p, ap)
        alpha = rho / red([rhat.dot(ap)])[0]
        s.copy(resid)
        s.addScaled(-alpha, ap)
        if opts is not None:
            gmg.cycle(opts, s, ms)
            gmg.apply(ms, as_)
        else:
            gmg.apply(s, as_)
        num, den = red([as_.dot(s), as_.dot(as_)])
        omega = num / den
        if opts is not None:
            x.addScaled(alpha, mp, omega, ms)
        else:
            x.addScaled(alpha, p, omega, s)
        resid.addScaled(-alpha, ap, -omega, as_)
        rho_new, rsq = red([resid.dot(rhat), resid.twoNormSqLocal()])
        beta = rho_new * alpha / (rho * omega)
        p.addScaled(-omega, ap)
        p.scaleThenAdd(beta, resid)
        its += 1
        rho = rho_new
        rnorm = math.sqrt(rsq)
    return its, rnorm / r0_norm
