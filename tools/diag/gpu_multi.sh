set -x
cd $GRAFT_REPO_ROOT
SOHIT_BENCH_BACKEND=gloo SOHIT_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | tail -3 | cut -c1-900
python -c "
from swiftortho_amd import synthprot
open('/tmp/w.fsa','wb').write(synthprot.synthprot(3000,300,77))
"
python bin/find_hit.py -p blastp -i /tmp/w.fsa -d /tmp/w.fsa -o /tmp/w.sc -e 1e-5 -s 111111 -a 1 2>&1 | tail -2
make -C oracle -s >/dev/null 2>&1
./oracle/sohit_cpu -p blastp -i /tmp/w.fsa -d /tmp/w.fsa -o /tmp/w_ora.sc -e 1e-5 -s 111111 -r AST,CFILMVY,DN,EQ,G,H,KR,P,W -M 120000000 -c 50000 -j 1 -v 500 2>/dev/null
cmp /tmp/w.sc /tmp/w_ora.sc && echo "CLI output identical to oracle: $(wc -l < /tmp/w.sc) rows"
python bin/fsearch-c -p blastp -i /tmp/w.fsa -d /tmp/w.fsa -o /tmp/w2.sc -e 1e-5 -s 111111 -M 120000000 -j 1 -l 100 -u 200 && ./oracle/sohit_cpu -p blastp -i /tmp/w.fsa -d /tmp/w.fsa -o /tmp/w2_ora.sc -e 1e-5 -s 111111 -M 120000000 -j 1 -l 100 -u 200 2>/dev/null; cmp /tmp/w2.sc /tmp/w2_ora.sc && echo "fsearch-c stand-in identical on -l/-u block: $(wc -l < /tmp/w2.sc) rows"
SOHIT_BENCH_BACKEND=gloo SOHIT_BENCH_ONE_GPU=1 python bin/find_hit.py -p blastp -i /tmp/w.fsa -d /tmp/w.fsa -o /tmp/w3.sc -e 1e-5 -s 111111 -a 3 2>&1 | tail -1
cmp /tmp/w3.sc /tmp/w_ora.sc && echo "find_hit -a 3 (three ranks, gloo gather, one GPU) identical to oracle: $(wc -l < /tmp/w3.sc) rows"
