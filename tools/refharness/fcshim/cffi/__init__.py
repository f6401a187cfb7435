"""Throw-away stand-in so that the reference's bin/find_cluster.py imports in the build container (cffi is absent):
`ffi.verify` fails, which sends the script down its own `except` branch (the mmap constants are only used by the
pypy code path, never by `-a mcl`)."""


class FFI(object):
    NULL = None

    def cdef(self, s):
        pass

    def verify(self, s):
        raise RuntimeError("cffi is not available here")
