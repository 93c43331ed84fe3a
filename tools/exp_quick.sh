set -e
echo "== parity"; timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; timeout -k 10 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline "$@" 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log; }
run --synth 3op_series --log2n 11
run --synth 4op_series --log2n 12 --parents 8192 --offspring 24576
run --log2n 13 --parents 4096 --offspring 12288
