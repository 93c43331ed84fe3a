# parity + default bench + two other population sizes (quick A/B after a kernel change)
set -e
echo "== parity"; timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
run() { echo "== $*"; timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log; }
run
run --parents 4096 --offspring 12288
run --parents 32768 --offspring 98304
run --parents 65536 --offspring 196608
