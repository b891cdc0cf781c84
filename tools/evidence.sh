# The round's evidence set in one gpurun call (bash tools/evidence.sh <tag>): PMC passes + kernel stats (C3, C5, C2), the driver's bench
# command twice, the script's default, the configuration / settings / band / blend / console tables.  Summaries land in gpurun_out/<tag>_*;
# copy the ones to be judged into profiles/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r06_v7}
bash tools/pmc_passes.sh $T C3 > gpurun_out/${T}_passes.log 2>&1
bash tools/pmc_passes.sh ${T}_C5 C5 >> gpurun_out/${T}_passes.log 2>&1
bash tools/pmc_passes.sh ${T}_C2 C2 >> gpurun_out/${T}_passes.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/${T}_pmc_traffic_all.json
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_cmd.json 2> gpurun_out/${T}_bench_driver_cmd.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_driver_cmd_b.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_default.json 2>/dev/null
timeout 900 python tools/bench_configs.py 2>&1 | tail -7 > gpurun_out/${T}_configs.md
timeout 900 python tools/bench_modes.py 2>&1 | tail -11 > gpurun_out/${T}_modes.md
(timeout 600 python tools/band_time.py; timeout 600 python tools/weak_time.py) > gpurun_out/${T}_bands.txt 2>&1
bash tools/blend_prof.sh > gpurun_out/${T}_blend.txt 2>&1
python tools/console_frame.py > gpurun_out/${T}_console.txt 2>&1
ls gpurun_out | grep $T
# config C4's data path through the product's own boundary with the ranks sharing this one GPU (gloo carries the set-up bytes and the barriers):
# the shared-framebuffer transport (b32_band_export / _import, device-side epoch words), 4 and 8 ranks -- per-rank compute, NOT a scaling curve
for n in 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 20 --warmup 3 \
    --transport shm --dist-backend gloo > gpurun_out/${T}_bench_c4_shm_${n}ranks_one_gpu.json 2> gpurun_out/${T}_bench_c4_shm_${n}ranks_one_gpu.err
done
ls gpurun_out | grep $T
