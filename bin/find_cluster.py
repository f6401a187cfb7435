#!/usr/bin/env python3
"""Drop-in for SwiftOrtho's bin/find_cluster.py -a mcl (same flags, same stdout): see swiftortho_amd/find_cluster.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftortho_amd.find_cluster import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
