#!/usr/bin/env python3
"""Drop-in for SwiftOrtho's bin/find_orth.py (same flags, same stdout): see swiftortho_amd/find_orth.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftortho_amd.find_orth import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
