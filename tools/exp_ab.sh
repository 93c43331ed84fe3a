#!/bin/bash
# same-box A/B of the default library against variants: bash tools/exp_ab.sh variants/x.so [variants/y.so ...]
# un-instrumented loop + event pass: configs[2], configs[4]'s shard, configs[3]'s shard, 3-op N = 2048 P = 65536, 2-op P = 16384, 3-op P = 1024
L=survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so
run() { # lib args...
  local lib=$1; shift
  SOTS_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --full-sort-steps 0 --sustain 0.3 "$@" 2>/dev/null > /tmp/b.log
  echo -n "$(basename $lib)  [$*]  "; python3 tools/show_bench.py /tmp/b.log
}
for rep in 1 2; do
  for lib in $L "$@"; do
    run $lib --config 2
    run $lib --config 4 --shard-of 8
    run $lib --config 3 --shard-of 8
    run $lib --synth 3op_series --log2n 11 --parents 16384 --offspring 49152
    run $lib --synth 2op --log2n 10 --parents 4096 --offspring 12288
    run $lib --synth 3op_series --log2n 11 --parents 256 --offspring 768
  done
done
