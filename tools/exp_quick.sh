set -e
run() { echo "== $*"; timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log; }
run
run --parents 32768 --offspring 98304
run --parents 65536 --offspring 196608
run --parents 262144 --offspring 786432
