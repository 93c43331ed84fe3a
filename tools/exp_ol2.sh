L=survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so
run() { local lib=$1; shift; SOTS_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --full-sort-steps 0 --sustain 0.3 "$@" 2>/dev/null > /tmp/b.log; echo -n "$(basename $lib)  [$*]  "; python3 tools/show_bench.py /tmp/b.log; }
for rep in 1 2; do for lib in $L variants/olall8.so variants/olall16.so; do
  run $lib --synth 2op --log2n 10 --parents 14336 --offspring 43008
  run $lib --synth 4op_series --log2n 12 --parents 12288 --offspring 36864
  run $lib --synth 3op_series --log2n 11 --parents 12288 --offspring 36864
done; done
