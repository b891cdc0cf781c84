# final checks of the build that is committed: smoke, full GPU suite, the driver's single-GPU command, and bench.py on 2 and 8 ranks
# sharing the one GPU (gloo transport, --check compares the gathered frame with the committed hash)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4final}; mkdir -p $OUT
python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench1.json 2> $OUT/bench1.err; tail -c 400 $OUT/bench1.json | head -c 200; echo
for n in 2 8; do timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 10 --warmup 3 --dist-backend gloo --check --no-cpu-baseline > $OUT/bench$n.json 2> $OUT/bench$n.err; python - <<P
import json
try:
    d=json.loads(open('$OUT/bench$n.json').read().strip().splitlines()[-1]); print($n, d['ms_per_step'], d['value'], d.get('check'), d.get('scaling'))
except Exception as e: print($n, 'ERR', e, open('$OUT/bench$n.err').read()[-600:])
P
done
