#!/bin/bash
# clocks and power while the generation loop runs: bash tools/power_probe.sh [bench workload args]
# (the loop is kept running for ~6 s; rocm-smi sampled every 0.5 s beside it)
out=gpurun_out/power_probe.log; : > $out
timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --full-sort-steps 0 --sustain 6 "$@" > /tmp/pp_bench.json 2>/dev/null &
bp=$!
sleep 3
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | tr '\n' ';' >> $out; echo >> $out
  sleep 0.5
done
wait $bp
python3 tools/show_bench.py /tmp/pp_bench.json >> $out
echo "== idle" >> $out
sleep 2
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ';' >> $out; echo >> $out
cat $out
