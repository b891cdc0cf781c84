cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4l}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "transparent_faces or atlas or cpp or semantics or flight" 2>&1 | tail -5 | tee $OUT/tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<P
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('C3', d['ms_per_step'], d['protocol'].get('pipelined_frames_in_timed_region'), d['roofline']['valu'])
print({k:(v['ms_per_frame'],v['bit_exact_vs_committed_hash']) for k,v in d['configs'].items()})
P
(timeout 300 python tools/soak.py 150 9101 2>&1 | tail -4; timeout 300 python tools/soak.py 150 9102 2>&1 | tail -4) | tee $OUT/soak.txt
