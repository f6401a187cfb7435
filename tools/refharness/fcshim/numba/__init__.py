"""Throw-away stand-in so that the reference's bin/find_cluster.py imports in the build container (numba is absent):
its @jit decorators become no-ops, which changes speed, not results."""


def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
