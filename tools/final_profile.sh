#!/bin/bash
# One gpurun call that produces every artefact profiles/ holds for the current build:
#   bash tools/final_profile.sh <tag>       (writes gpurun_out/<tag>_*; copy what you keep to profiles/)
# rocprofv3 gets the program itself after `--`; counters are collected in their own passes.
set -e
tag=${1:-final}
out=gpurun_out
mkdir -p $out
echo "== pytest -m gpu"; timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== bench (default)"; timeout -k 10 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; python tools/show_bench.py $out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== rocprofv3 --kernel-trace --stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/${tag}_stats.log 2>&1
cp $(find $out/${tag}_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
python3 tools/show_kernel_stats.py $out/${tag}_kernel_stats.csv
echo "== PMC passes"
bash tools/pmc_collect.sh $out/${tag}_pmc > $out/${tag}_pmc.log 2>&1
cp $out/${tag}_pmc/pmc_summary.json $out/${tag}_pmc_summary.json
tail -5 $out/${tag}_pmc.log
