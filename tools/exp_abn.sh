#!/bin/bash
# same-box A/B, N rounds, of the default library against variants on the three bench workloads:
#   REPS=3 bash tools/exp_abn.sh variants/x.so [variants/y.so ...]
L=survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so
run() { local lib=$1; shift
  SOTS_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --full-sort-steps 0 --sustain 0.3 "$@" 2>/dev/null > /tmp/b.log
  echo -n "$(basename $lib)  [$*]  "; python3 tools/show_bench.py /tmp/b.log; }
for rep in $(seq 1 ${REPS:-3}); do for lib in $L "$@"; do
  run $lib --config 2; run $lib --config 3 --shard-of 8; run $lib --config 4 --shard-of 8
done; done
