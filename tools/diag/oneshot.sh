#!/bin/bash
# wall time of ONE `bin/find_hit.py` command on the config-3 workload (process start to output file), with the stage laps
# (SOHIT_TIMING=1).  usage: bash tools/diag/oneshot.sh [proteins] [seed pattern] [runs]
N=${1:-100000}; S=${2:-11111011111}; K=${3:-3}
cd $GRAFT_REPO_ROOT
python3 -c "
import sys; sys.path.insert(0,'.')
from swiftortho_amd import synthprot
open('/tmp/os.fsa','wb').write(synthprot.synthprot($N, 300))"
for k in $(seq 1 $K); do
  T0=$(date +%s%N)
  SOHIT_TIMING=1 python3 bin/find_hit.py -p blastp -i /tmp/os.fsa -d /tmp/os.fsa -o /tmp/os.sc -e 1e-5 -s $S -a 1 -j 1 -v 500 2>&1 | grep find_hit
  echo "wall $(( ($(date +%s%N) - T0) / 1000000 )) ms"
done
wc -l /tmp/os.sc; md5sum /tmp/os.sc
