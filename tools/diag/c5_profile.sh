#!/bin/bash
# where the two CPU-side CLIs of config 5 spend their time: cProfile of bin/find_orth.py and bin/find_cluster.py on the 100 k-protein set
cd "$GRAFT_REPO_ROOT" || exit 1
T=/tmp/c5p; mkdir -p $T gpurun_out
python3 -c "
import sys; sys.path.insert(0,'.')
from swiftortho_amd import synthprot
open('$T/x.fsa','wb').write(synthprot.synthprot(100000, 300))"
python3 bin/find_hit.py -p blastp -i $T/x.fsa -d $T/x.fsa -o $T/x.sc -e 1e-5 -s 11111011111 -a 1 -j 1 -v 500 > /dev/null 2>&1
python3 -m cProfile -o $T/orth.prof bin/find_orth.py -i $T/x.sc > $T/x.opc
python3 -m cProfile -o $T/clu.prof bin/find_cluster.py -i $T/x.opc -a mcl -I 1.5 > $T/x.grp
python3 - <<PY
import pstats
for n in ("orth", "clu"):
    print("=====", n)
    pstats.Stats("$T/%s.prof" % n).sort_stats("tottime").print_stats(14)
PY
