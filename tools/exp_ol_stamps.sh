# usage: bash tools/exp_ol_stamps.sh v1 v2 ...   (variants/<v>.so, -DSOTS_STAMP builds): in-kernel cycles per sample of the synthesis loop
for v in "$@"; do SOTS_PROBE_WAVES=8 SOTS_LIB_PATH=variants/$v.so timeout -k 5 120 python tools/stamp_probe.py 32768 12 4op_series 2>&1 | tail -3; done
for v in "$@"; do SOTS_PROBE_WAVES=8 SOTS_LIB_PATH=variants/$v.so timeout -k 5 120 python tools/stamp_probe.py 65536 10 2op 2>&1 | tail -3; done
