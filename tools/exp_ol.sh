#!/bin/bash
# A/B of k_synth_ol (operators in the lanes) against the kernels it replaces (variants/no_ol.so: -DSOTS_SYNTH_NO_OL), same box,
# un-instrumented loop + event pass: configs[3]'s shard, configs[2], configs[4]'s shard, 3-op N = 2048 at 65536 and 32768.
L=survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so
run() { # name lib args...
  local name=$1 lib=$2; shift 2
  SOTS_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --full-sort-steps 0 --sustain 0.3 "$@" 2>/dev/null > /tmp/b.log
  echo -n "$name  [$*]  "; python3 tools/show_bench.py /tmp/b.log
}
for rep in 1 2; do
  for v in "ol $L" "no_ol variants/no_ol.so"; do
    set -- $v
    run $1 $2 --config 3 --shard-of 8
    run $1 $2 --config 2
    run $1 $2 --config 4 --shard-of 8
    run $1 $2 --synth 3op_series --log2n 11 --parents 16384 --offspring 49152
    run $1 $2 --synth 4op_series --log2n 12 --parents 16384 --offspring 49152
    run $1 $2 --synth 2op --log2n 10 --parents 8192 --offspring 24576
    run $1 $2 --synth 2op --log2n 10 --parents 4096 --offspring 12288
  done
done
