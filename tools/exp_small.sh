for lib in survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so variants/libsots_deep.so; do
  echo "== $lib"
  for cfg in "--parents 16 --offspring 16 --synth 3op_series --log2n 11" "--parents 256 --offspring 768 --synth 3op_series --log2n 11" "--parents 4096 --offspring 12288 --synth 3op_series --log2n 11" "--parents 16 --offspring 16 --synth 4op_series --log2n 12" "--parents 256 --offspring 768 --synth 4op_series --log2n 12" "--parents 4096 --offspring 12288 --synth 4op_series --log2n 12"; do
    SOTS_LIB_PATH=$lib timeout -k 10 300 python bench.py $cfg --steps 400 --warmup 50 --full-sort-steps 0 --no-cpu-baseline --sustain 0 2>/dev/null > /tmp/b.json; echo "$cfg"; python tools/show_bench.py /tmp/b.json
  done
done
SOTS_LIB_PATH=variants/libsots_deep.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
