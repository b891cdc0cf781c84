# rocprofv3 evidence for one build: kernel stats + four separate PMC passes of the bench.py C3 workload (run on the GPU box through gpurun).
# usage: bash tools/pmc_passes.sh <tag> [config]   -> gpurun_out/<tag>_kstats.txt, gpurun_out/<tag>_pmc.txt, profiles/pmc_traffic.json refreshed
# (config: C3 default, or C1 / C2 / C5 -- bench.py --config)
TAG=${1:-run}
CFG=${2:-C3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
X=${3:-}      # extra bench.py arguments, e.g. "--routes-off 2048"
B="python $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-pipeline $X"
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmcA -o a -- $B > $O/pmcA.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $O/pmcB -o b -- $B > $O/pmcB.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcC -o c -- $B > $O/pmcC.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcD -o d -- $B > $O/pmcD.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kstats -o k -- python $R/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-configs $X > $O/kstats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kstats1 -o k -- python $R/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-pipeline $X > $O/kstats1.log 2>&1
: > $R/gpurun_out/${TAG}_pmc.txt
for p in A B C D; do f=$(find $O/pmc$p -name "*.db" 2>/dev/null | head -1); if [ -n "$f" ]; then echo "== pass $p" >> $R/gpurun_out/${TAG}_pmc.txt; python $R/tools/rocpd_pmc.py $f > $O/pmc$p.txt; cat $O/pmc$p.txt >> $R/gpurun_out/${TAG}_pmc.txt; fi; done
f=$(find $O/kstats -name "*.db" 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/rocpd_stats.py $f > $R/gpurun_out/${TAG}_kstats.txt
f=$(find $O/kstats1 -name "*.db" 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/rocpd_stats.py $f > $R/gpurun_out/${TAG}_kstats_one_stream.txt
tail -1 $O/kstats.log > $R/gpurun_out/${TAG}_bench_under_rocprof.json
python $R/tools/pmc_traffic.py $O $CFG profiles/${TAG}_pmc.txt > $R/gpurun_out/${TAG}_traffic.json 2>&1 && cp $R/profiles/pmc_traffic.json $R/gpurun_out/${TAG}_pmc_traffic.json
rm -rf $O/pmc?/ $O/kstats/ $O/kstats1/     # the databases are large; the text summaries above are what is kept
ls $R/gpurun_out/ | grep $TAG
