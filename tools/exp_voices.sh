set -e
echo "== parity"; timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
echo "== parity SOTS_SYNTH_CUT=0"; SOTS_SYNTH_CUT=0 timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
run() { echo "== $*"; timeout -k 10 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline "$@" 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log; }
run
run --parents 4096 --offspring 12288
run --parents 8192 --offspring 24576
run --parents 32768 --offspring 98304
run --synth 3op_series --log2n 11
run --synth triple_parallel
run --synth 4op_series --log2n 12 --parents 8192 --offspring 24576
run --log2n 12 --parents 8192 --offspring 24576
