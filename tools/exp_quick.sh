set -e
run() { echo "== $*"; timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>&1 > /tmp/b.log | tail -3; python tools/show_bench.py /tmp/b.log; }
run --parents 262144 --offspring 786432
run --parents 131072 --offspring 393216
run --parents 300000 --offspring 700001
