"""Container-only: run the REAL reference launcher (bin/find_hit.py) end to end.

The launcher is plain Python 3; the native it shells out to (lib/fsearch-c, an RPython build that
cannot exist here) is replaced by a two-line shell script that runs the converted reference core
under CPython (tools/refharness/refload.py).  Everything lives in a temp copy; nothing of the
reference is written into this repository.

    run(args, max_chr=None) -> None     # args = find_hit.py's own flags

`max_chr` rewrites the launcher's split threshold (find_hit.py:287; the author's own test value is
the commented line 288) in the temp copy so that the >= 4.2e9-byte reference split/merge path
(303-351) can be exercised on small inputs.
"""
import os
import shutil
import stat
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("SWIFTORTHO_REFERENCE", "/root/reference")


def make_tree(max_chr=None):
    root = tempfile.mkdtemp(prefix="ref_find_hit_")
    os.makedirs(os.path.join(root, "bin"))
    os.makedirs(os.path.join(root, "lib"))
    src = open(os.path.join(REFERENCE, "bin", "find_hit.py")).read()
    if max_chr is not None:
        old = "        max_chr = 4200000000\n"
        assert old in src
        src = src.replace(old, "        max_chr = %d\n" % max_chr)
    open(os.path.join(root, "bin", "find_hit.py"), "w").write(src)
    shim = os.path.join(root, "lib", "fsearch-c")
    open(shim, "w").write("#!/bin/sh\nexec %s %s \"$@\"\n" % (sys.executable, os.path.join(HERE, "refload.py")))
    os.chmod(shim, os.stat(shim).st_mode | stat.S_IEXEC | stat.S_IXGRP | stat.S_IXOTH)
    return root


def run(args, max_chr=None, cwd=None):
    root = make_tree(max_chr)
    try:
        env = dict(os.environ, LC_ALL="C")  # `sort -m` of the merge path: bytewise last-resort comparison
        subprocess.run([sys.executable, os.path.join(root, "bin", "find_hit.py")] + list(args), check=True, cwd=cwd, env=env,
                       stdout=subprocess.DEVNULL)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    mc = None
    av = sys.argv[1:]
    if av and av[0].startswith("--max-chr="):
        mc = int(av[0].split("=")[1])
        av = av[1:]
    run(av, mc)
