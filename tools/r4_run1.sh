set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python bench.py > gpurun_out/r4a/bench_base.json 2> gpurun_out/r4a/bench_base.err
tail -c 3000 gpurun_out/r4a/bench_base.json
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "band_ranks" > gpurun_out/r4a/band_ranks.log 2>&1; tail -3 gpurun_out/r4a/band_ranks.log
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 10 --warmup 2 --dist-backend gloo --check --no-cpu-baseline > gpurun_out/r4a/bench_gloo8.json 2> gpurun_out/r4a/bench_gloo8.err; tail -c 1500 gpurun_out/r4a/bench_gloo8.json; tail -5 gpurun_out/r4a/bench_gloo8.err
