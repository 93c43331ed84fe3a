#!/bin/bash
# One gpurun call that produces the artefacts profiles/ holds for one workload of the current build:
#   bash tools/final_profile.sh <tag> [bench.py workload args, e.g. --config 3 --shard-of 8]
# writes gpurun_out/<tag>_*; copy what you keep to profiles/.  rocprofv3 gets the program itself after `--`;
# counters are collected in their own passes (tools/pmc_collect.sh).
set -e
tag=${1:-final}; shift || true
out=gpurun_out
mkdir -p $out
echo "== bench $@"; timeout -k 10 600 python bench.py "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err; python tools/show_bench.py $out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== rocprofv3 --kernel-trace --stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustain 0 --full-sort-steps 0 "$@" > $out/${tag}_stats.log 2>&1
cp $(find $out/${tag}_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
python3 tools/show_kernel_stats.py $out/${tag}_kernel_stats.csv > $out/${tag}_kernel_stats.txt; head -8 $out/${tag}_kernel_stats.txt
echo "== PMC passes"
bash tools/pmc_collect.sh $out/${tag}_pmc --sustain 0 --full-sort-steps 0 "$@" > $out/${tag}_pmc.log 2>&1
cp $out/${tag}_pmc/pmc_summary.json $out/${tag}_pmc_summary.json
tail -3 $out/${tag}_pmc.log
