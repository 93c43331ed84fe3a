set -e
run() { echo "== $*"; timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log; }
for kt in 1 2 4; do
export SOTS_SORT_KT=$kt; echo "#### kt $kt"
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "sort or full_size" 2>&1 | tail -1
run
run --parents 32768 --offspring 98304
run --parents 65536 --offspring 196608
run --parents 262144 --offspring 786432
done
