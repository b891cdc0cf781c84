set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4f}; mkdir -p $OUT
shift
python tools/exp_variants.py run base "$@" 2>&1 | tee $OUT/exp.txt
bash tools/pmc_compare.sh base "$@" 2>&1 | grep -v "^+" | tee $OUT/pmc.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "frame_parity or full_size or c5 or C5 or smoke or modes or blend" 2>&1 | tail -4 | tee $OUT/tests.txt
