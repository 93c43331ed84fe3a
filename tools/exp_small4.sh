run() { echo "== $*"; timeout -k 10 300 python3 bench.py "$@" --steps 400 --warmup 50 --full-sort-steps 0 --no-cpu-baseline --sustain 0 2>/dev/null > /tmp/b.json; python3 tools/show_bench.py /tmp/b.json; }
for p in "16 48" "256 768" "512 1536" "768 2304" "1024 3072"; do set -- $p; run --parents $1 --offspring $2; done
for p in "256 768" "512 1536" "768 2304"; do set -- $p; run --parents $1 --offspring $2 --synth 3op_series --log2n 11; done
for p in "256 768" "512 1536"; do set -- $p; run --parents $1 --offspring $2 --synth 4op_series --log2n 12; done
