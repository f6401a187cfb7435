#!/usr/bin/env python3
"""Drop-in for SwiftOrtho's bin/find_hit.py backed by MI355X GPUs (see swiftortho_amd/find_hit.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from swiftortho_amd.find_hit import main  # noqa: E402

if __name__ == '__main__':
    # A command, not a library call: the process ends with the run, so the one-GPU path neither frees its device buffers one by one
    # nor runs the interpreter's teardown (0.1 s of a 0.7 s command) -- output files are closed and flushed by then.
    rc = main(fast_exit=True)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(rc or 0)
