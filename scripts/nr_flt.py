#!/usr/bin/env python3
"""Drop-in for SwiftOrtho's scripts/nr_flt.py (collapse identical sequences): see swiftortho_amd/nr.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftortho_amd.nr import main_nr_flt  # noqa: E402

if __name__ == "__main__":
    sys.exit(main_nr_flt())
