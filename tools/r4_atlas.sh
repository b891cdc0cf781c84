cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "atlas or clut or smoke or frame_parity" 2>&1 | tail -15 | tee $OUT/tests.txt
python tools/exp_variants.py run base atlas 2>&1 | tee $OUT/exp.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r4h/bench.json').read().strip().splitlines()[-1])
print('C3', d['ms_per_step'], {k:(v['ms_per_frame'],v['bit_exact_vs_committed_hash']) for k,v in d['configs'].items()})
P
