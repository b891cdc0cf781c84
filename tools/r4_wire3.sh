cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4q}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "f32_semantics or wireframe or editor_modes or wire" 2>&1 | tail -4 | tee $OUT/tests.txt
bash tools/r4_wireprof.sh ${1:-r4q} 2>&1 | grep "wire\|k_cover\|k_setup"
timeout 600 python tools/bench_modes.py 2>&1 | grep "default()\|game()" | tee $OUT/modes.md
