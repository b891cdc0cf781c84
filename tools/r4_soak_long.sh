# long soak on the current build: usage bash tools/r4_soak_long.sh <tag> <seed0> <n_seeds> <seconds each>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4soakL}; mkdir -p $OUT
for i in $(seq 0 $((${3:-3} - 1))); do timeout $((${4:-280} + 120)) python tools/soak.py ${4:-280} $((${2:-9500} + i)) 2>&1 | tail -3; done | tee $OUT/soak.txt
