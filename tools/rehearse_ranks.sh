# Rehearses the N > 1 path of bench.py on a 1-GPU box: two ranks share cuda:0, gloo backend.
for m in "" "--sync-migration"; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 30 --warmup 5 --backend gloo --share-gpu $m > gpurun_out/bench_2rank.log 2>&1
  echo "rc=$?"
  grep "^{" gpurun_out/bench_2rank.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['migration'], d['ms_per_step'], d['best_fitness_sse'])"
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_1rank_tdr.log 2>&1
echo "rc=$?"
python tools/show_bench.py gpurun_out/bench_1rank_tdr.log
