# One parameterised GPU job (replaces the per-experiment scripts): bash tools/gpu_job.sh <tag> <steps...>
#   steps: tests[:<-k expr>]  quick  bench  ab[:args]  modes  configs  console  soak[:n:seed]  pmc[:cfg[:routes_off]]  bands
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-job}; shift
OUT=gpurun_out/$T; mkdir -p $OUT
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  case $name in
    tests)  if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -q -x -m gpu -k "$arg" 2>&1 | tail -15 | tee $OUT/tests.txt; else timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 | tee $OUT/tests.txt; fi ;;
    quick)  timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "frame_parity or full_size or c5 or C5 or smoke or long_lists or needles or band" 2>&1 | tail -8 | tee $OUT/quick.txt ;;
    bench)  python bench.py --gpus 1 --steps 20 --warmup 5 $arg > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json ;;
    ab)     timeout 900 python tools/ab_routes.py $(echo $arg | tr ':' ' ') 2>&1 | grep -v "^$" | tee -a $OUT/ab.txt ;;
    modes)  timeout 900 python tools/bench_modes.py 2>&1 | tail -12 | tee $OUT/modes.md ;;
    configs) timeout 900 python tools/bench_configs.py 2>&1 | tail -7 | tee $OUT/configs.md ;;
    console) python tools/console_frame.py 2>&1 | tee $OUT/console.txt ;;
    soak)   n=$(echo $arg | cut -d: -f1); seed=$(echo $arg | cut -d: -f2); timeout 900 python tools/soak.py ${n:-150} ${seed:-9501} 2>&1 | tail -4 | tee -a $OUT/soak.txt ;;
    pmc)    cfg=$(echo $arg | cut -d: -f1); off=$(echo $arg: | cut -d: -f2); tag=${T}_${cfg:-C3}${off:+_off$off}
            bash tools/pmc_passes.sh $tag ${cfg:-C3} "${off:+--routes-off $off}" > $OUT/passes_$tag.log 2>&1; cat gpurun_out/${tag}_pmc.txt | head -60; head -12 gpurun_out/${tag}_kstats_one_stream.txt ;;
    bands)  (timeout 600 python tools/band_time.py; timeout 600 python tools/weak_time.py) 2>&1 | tee $OUT/bands.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
