# FETCH_SIZE / WRITE_SIZE (separate passes) of one setting of mode_prof.py, pipeline off.  usage: MODE=game bash tools/r4_modepmc_hbm.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r4hbm}; mkdir -p $OUT
B="python $R/tools/mode_prof.py ${MODE:-game} 64"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/hb_$c; timeout 120 rocprofv3 --kernel-trace --pmc $c -d /tmp/hb_$c -o a -- $B > /tmp/hb_$c.log 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/hb_$c -name "*.db" | head -1) | grep "k_setup\|k_cover\|k_wire\|k_clear"
done | tee $OUT/hbm.txt
rm -rf /tmp/hb_k; timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/hb_k -o k -- $B > /tmp/hb_k.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/hb_k -name "*.db" | head -1) | grep "k_setup\|k_cover\|k_wire" | cut -c1-150 | tee -a $OUT/hbm.txt
