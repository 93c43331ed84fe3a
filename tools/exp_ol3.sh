L=survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so
run() { local lib=$1; shift; SOTS_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --full-sort-steps 0 --sustain 0.3 "$@" 2>/dev/null > /tmp/b.log; echo -n "$(basename $lib)  [$*]  "; python3 tools/show_bench.py /tmp/b.log; }
for rep in 1 2; do for lib in $L variants/no_ol.so; do
  run $lib --synth 4op_series --log2n 12 --parents 16384 --offspring 49152
  run $lib --synth 4op_series --log2n 12 --parents 32768 --offspring 98304
  run $lib --synth 3op_series --log2n 11 --parents 16384 --offspring 49152
  run $lib --synth 3op_series --log2n 11 --parents 32768 --offspring 98304
  run $lib --synth 4op_series --log2n 12 --parents 20000 --offspring 60032
done; done
