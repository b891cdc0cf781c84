cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4aa; mkdir -p $OUT
echo "== base"; timeout 600 python tools/bench_modes.py 2>&1 | grep "z-buffer\|game()\|default()" | tee $OUT/modes_base.md
echo "== z5"; B32_LIB=$PWD/bonnie-32_amd/csrc/exp_z5.so timeout 600 python tools/bench_modes.py 2>&1 | grep "z-buffer\|game()\|default()" | tee $OUT/modes_z5.md
