L=survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so
for rep in 1 2 3; do for lib in $L "$@"; do
SOTS_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --full-sort-steps 0 --sustain 0.3 --config 3 --shard-of 8 2>/dev/null > /tmp/b.log; echo -n "$(basename $lib) "; python3 tools/show_bench.py /tmp/b.log; done; done
