#!/bin/bash
# usage: tools/pmc_script.sh <tag> <script relative to the repo> [args]  -- as tools/pmc_run.sh (separate rocprofv3 --pmc passes, one
# counter group each, then --kernel-trace --stats) for any script that prints one JSON line; summaries under gpurun_out/<tag>_*.csv
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcs_$i
  rocprofv3 --pmc $grp -d /tmp/pmcs_$i -o run -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/pmcs_$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py counters $(find /tmp/pmcs_* -name "*.db" | sort) > $out/${tag}_pmc_summary.csv
rm -rf /tmp/kts
rocprofv3 --kernel-trace --stats -d /tmp/kts -o run -- python3 $GRAFT_REPO_ROOT/"$@" > $out/${tag}_bench_profiled.json 2> /tmp/kts.err
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py kernels $(find /tmp/kts -name "*.db" | head -1) > $out/${tag}_kernel_stats.csv
tail -1 $out/${tag}_bench_profiled.json | cut -c1-400
head -12 $out/${tag}_kernel_stats.csv
grep "prefilter\|scan_queue\|scanw" $out/${tag}_pmc_summary.csv
