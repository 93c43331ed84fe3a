#!/bin/bash
# same-box A/B of the selection (configs[2], P = 16384 and P = 4096): bash tools/exp_sel.sh variants/x.so ...
L=survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so
run() { # lib args...
  local lib=$1; shift
  SOTS_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --full-sort-steps 0 --sustain 0.3 "$@" 2>/dev/null > /tmp/b.log
  echo -n "$(basename $lib)  [$*]  "; python3 tools/show_bench.py /tmp/b.log
}
for rep in 1 2 3; do
  for lib in $L "$@"; do
    run $lib --config 2
    run $lib --synth 2op --log2n 10 --parents 4096 --offspring 12288
    run $lib --synth 2op --log2n 10 --parents 1024 --offspring 3072
  done
done
