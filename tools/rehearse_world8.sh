#!/bin/bash
# BASELINE configs[3] and configs[4] AT THEIR STATED SIZE on ONE GPU (eight islands share device 0): the record
# profiles/r04_world8_rehearsal.jsonl.  The library's group host (8 persistent threads, one process) at 8 islands;
# the process host (torch.distributed over gloo) at 4 of the 8 shards - a GPU box admits at most 6 processes on its card.
set -e
out=gpurun_out/r04_world8
mkdir -p $out
for c in 3 4; do
  g=$([ $c = 3 ] && echo 60 || echo 100)
  echo "== group_overhead --config $c --islands 8"
  timeout -k 10 500 python tools/group_overhead.py --config $c --islands 8 --gens $g --quick > $out/overhead_c$c.jsonl
  echo "== bench --gpus 8 --share-gpu --host group --config $c"
  timeout -k 10 300 python bench.py --gpus 8 --share-gpu --host group --config $c --steps 30 --warmup 5 --sustain 0.5 --full-sort-steps 0 > $out/bench_group8_c$c.json 2> $out/bench_group8_c$c.err
  echo "== bench --gpus 4 --shard-of 8 --share-gpu --host process --backend gloo --config $c"
  timeout -k 10 300 python bench.py --gpus 4 --shard-of 8 --share-gpu --host process --backend gloo --config $c --steps 30 --warmup 5 --sustain 0.5 --full-sort-steps 0 > $out/bench_proc4_c$c.json 2> $out/bench_proc4_c$c.err
done
ls -la $out
