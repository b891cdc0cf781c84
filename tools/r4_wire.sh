cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4m}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "wireframe or editor_modes or wire or batched" 2>&1 | tail -8 | tee $OUT/tests.txt
timeout 600 python tools/bench_modes.py 2>&1 | tail -12 | tee $OUT/modes.md
EXP_ROUTES=1024 timeout 600 python tools/bench_modes.py 2>&1 | grep "default()" | tee $OUT/modes_global_wire.md
