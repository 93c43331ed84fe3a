for pad in 0 32 64 128 544; do
  echo "PAD=$pad"
  SOTS_AUDIO_PAD=$pad timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,1),'Mcand/s', {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
"
done
