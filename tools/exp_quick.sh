set -e
echo "== parity"; timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
run() { echo "== $*"; timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log; }
run
run --parents 8192 --offspring 24576
run --synth 3op_series --log2n 11
