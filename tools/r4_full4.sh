# full GPU suite + the settings table + kernel stats of game() / default() on the current build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4f4}; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | tee $OUT/tests.txt
timeout 600 python tools/bench_modes.py 2>&1 | tee $OUT/modes.md | grep "default()\|game()"
cd /tmp
for m in game default; do rm -rf /tmp/wp_$m; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/wp_$m -o k -- python $GRAFT_REPO_ROOT/tools/mode_prof.py $m > /tmp/wp_$m.log 2>&1; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/wp_$m -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$OUT/${m}_kstats.md; grep "k_setup\|k_cover" $GRAFT_REPO_ROOT/$OUT/${m}_kstats.md | cut -c1-140; done
