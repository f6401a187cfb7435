#!/usr/bin/env python3
"""Drop-in for SwiftOrtho's bin/find_hit.py backed by MI355X GPUs (see swiftortho_amd/find_hit.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from swiftortho_amd.find_hit import main  # noqa: E402

if __name__ == '__main__':
    sys.exit(main())
