set -e
echo "== parity nowait"; SOTS_LIB_PATH=variants/libsots_nowait.so timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/exp_variants.sh variants/libsots_nowait.so
echo "== P=131072"
for lib in survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/libsots_hip.so variants/libsots_nowait.so; do
SOTS_LIB_PATH=$lib timeout -k 10 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --parents 32768 --offspring 98304 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log
done
