# the sizes README and DESIGN quote beside the BASELINE configs: small 2-op populations, the reference's 3-op voice, long rows, the other voices
run() { echo "== $*"; timeout -k 10 300 python3 bench.py "$@" --steps 400 --warmup 50 --full-sort-steps 0 --no-cpu-baseline --sustain 0 2>/dev/null > /tmp/b.json; python3 tools/show_bench.py /tmp/b.json; }
for p in "16 48" "256 768" "1024 3072" "4096 12288" "8192 24576"; do set -- $p; run --parents $1 --offspring $2; done
for p in "16 16" "256 768" "4096 12288"; do set -- $p; run --parents $1 --offspring $2 --synth 3op_series --log2n 11; done
for p in "256 768" "4096 12288"; do set -- $p; run --parents $1 --offspring $2 --synth 4op_series --log2n 12; run --parents $1 --offspring $2 --synth triple_parallel --log2n 10; done
run --parents 16384 --offspring 49152 --log2n 11
run --parents 4096 --offspring 12288 --log2n 13
run --parents 65536 --offspring 196608
