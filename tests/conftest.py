import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    """fsearch-c level goldens (flags of the native)."""
    return sorted(f[:-3] for f in os.listdir(GOLD) if f.endswith(".sc") and not f.startswith(("fh_", "orth_", "nr_", "het_")))


def het_golden_names():
    """length-heterogeneous goldens (tools/refharness/make_het_goldens.py): the REAL reference run over the -l/-u ranges of the
    queries below 4096 residues (it crashes on longer ones, fsearch.py:1362/1396/1487-1490); <name>.sc = the ranges' rows in order"""
    return sorted(f[:-3] for f in os.listdir(GOLD) if f.endswith(".sc") and f.startswith("het_"))


def launcher_golden_names():
    """goldens produced through the reference's bin/find_hit.py (flags of the launcher; block scheme, split + merge)."""
    return sorted(f[:-3] for f in os.listdir(GOLD) if f.endswith(".sc") and f.startswith("fh_"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


def orth_golden_cases():
    """(name, variant) of the find_orth goldens (outputs of the reference's bin/find_orth.py)"""
    import json
    out = []
    for f in sorted(os.listdir(GOLD)):
        if f.startswith("orth_") and f.endswith(".json"):
            for v in json.load(open(os.path.join(GOLD, f)))["variants"]:
                out.append((f[5:-5], v))
    return out
