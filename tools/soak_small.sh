run() { echo "== $*"; timeout -k 10 300 python3 bench.py "$@" --warmup 50 --full-sort-steps 0 --no-cpu-baseline --sustain 0 2>/dev/null > /tmp/b.json && python3 -c "
import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,1),'us/gen best', d.get('best_fitness_sse'))"; }
run --parents 16 --offspring 16 --synth 3op_series --log2n 11 --steps 20000
run --parents 512 --offspring 1536 --steps 20000
run --parents 1024 --offspring 3072 --steps 10000
run --parents 256 --offspring 768 --synth triple_parallel --log2n 10 --steps 20000
run --parents 512 --offspring 1536 --synth 4op_series --log2n 12 --steps 5000
run --parents 768 --offspring 2304 --synth 3op_series --log2n 11 --steps 10000
run --gpus 2 --share-gpu --host group --parents 512 --offspring 1536 --steps 5000
