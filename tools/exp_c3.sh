# usage: bash tools/exp_c3.sh lib1.so ...   (configs[3]'s shard, default build and each variant, twice round-robin)
for rep in 1 2; do
for lib in survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so "$@"; do
  echo -n "$(basename $lib)  "; SOTS_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --config 3 --shard-of 8 --steps 100 --warmup 10 --no-cpu-baseline --full-sort-steps 0 --sustain 0.3 2>/dev/null > /tmp/b.log; python3 tools/show_bench.py /tmp/b.log
done; done
