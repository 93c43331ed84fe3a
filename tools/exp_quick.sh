set -e
for m in 1 0; do
echo "== parity SOTS_FUSE_VARIATION=$m"; SOTS_FUSE_VARIATION=$m timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
done
run() { echo "== $*"; timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log; }
for m in 1 0 1 0; do
export SOTS_FUSE_VARIATION=$m; echo "#### SOTS_FUSE_VARIATION=$m"
run
done
export SOTS_FUSE_VARIATION=1
run --parents 4096 --offspring 12288
run --parents 32768 --offspring 98304
run --synth 4op_series --log2n 12 --parents 8192 --offspring 24576
