set -e
for m in SOTS_DUMMY=1 SOTS_SYNTH_STAGED=1; do
echo "== parity $m"; env $m timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
for P in 16384 65536; do
  par=$((P/4)); off=$((P-par))
  for mode in "SOTS_SYNTH_STAGED=1" "SOTS_SYNTH_DUO=1"; do
    echo "== $mode"; env SOTS_LIB_PATH=variants/libsots_stamp.so $mode timeout -k 10 120 python tools/stamp_probe.py $P 2>/dev/null
    env $mode timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --parents $par --offspring $off 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log
  done
done
