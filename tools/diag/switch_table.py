#!/usr/bin/env python3
"""The SOHIT_* switches of libsohit.so as a markdown table, generated from swiftortho_amd/csrc/tune.h (the ONE place they are declared):
    python tools/diag/switch_table.py > /tmp/table.md        (tools/diag/README.md holds its output)"""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "swiftortho_amd", "csrc", "tune.h")).read()
kinds = {"B": "0 / 1", "I": "integer", "D": "number", "P": "set = on"}
print("| variable | values | default | effect |\n|---|---|---|---|")
for line in src.splitlines():
    m = re.match(r'\s*X\((\w), (\w+), "(\w+)", ([^,]+), "(.*)"\)\s*\\?$', line)
    if m:
        k, _, env, d, text = m.groups()
        print("| `%s` | %s | %s | %s |" % (env, kinds[k], "-" if k == "P" else d.strip(), text))
    else:
        m = re.match(r'\s*/\* ---- (.*) ---- \*/', line)
        if m:
            print("| **%s** | | | |" % m.group(1))
