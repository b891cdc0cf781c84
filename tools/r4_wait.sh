cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4w; mkdir -p $OUT
for rep in 1 2; do for w in spin block; do
  B32_BENCH_WAIT=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $OUT/b_$w$rep.json 2>/dev/null
  python - <<P
import json
d=json.loads(open('$OUT/b_$w$rep.json').read().strip().splitlines()[-1])
print('$w', d['ms_per_step'], d['protocol'].get('host_wait'), d['protocol'].get('ms_per_step_median'))
P
done; done | tee $OUT/wait.txt
