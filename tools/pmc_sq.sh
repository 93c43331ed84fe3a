#!/bin/bash
# SQ counters (two passes) of one bench workload for the default library or SOTS_LIB_PATH: bash tools/pmc_sq.sh <outdir> [bench args]
out=${1:-gpurun_out/pmc_sq}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pass$i -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --full-sort-steps 0 --settle-ms 0 "$@" > $out/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $out | grep -A20 "k_synth"
