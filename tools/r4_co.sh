cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4z; mkdir -p $OUT
python tools/exp_variants.py run base | tee $OUT/exp.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "flight or full_size or frame_parity or dropped or async or band_ranks" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2>/dev/null; python - <<P
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('C3', d['ms_per_step'], d['protocol'].get('ms_per_step_median'), d['protocol'].get('ms_per_step_one_stream'), d['protocol'].get('ms_per_step_safe_mode'), d['protocol'].get('frame_latency_ms'))
print({k:(v['ms_per_frame'],v['bit_exact_vs_committed_hash']) for k,v in d['configs'].items()})
P
