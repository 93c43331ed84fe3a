# in-kernel cycles per sample of the synthesis loop (diagnostic -DSOTS_STAMP build in variants/)
set -e
for P in 16384 65536 131072; do
  for mode in "SOTS_SYNTH_CUT=1" "SOTS_SYNTH_CUT=0"; do
    echo "== P=$P $mode"; env SOTS_LIB_PATH=variants/libsots_stamp.so $mode timeout -k 10 120 python tools/stamp_probe.py $P 2>/dev/null
  done
done
