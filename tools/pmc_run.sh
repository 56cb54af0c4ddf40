#!/bin/bash
# usage: tools/pmc_run.sh <tag> <bench args...>   -- separate rocprofv3 --pmc passes (one counter group each, as the
# microarchitecture guide prescribes), summaries under gpurun_out/<tag>_*.csv
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $grp -d /tmp/pmc_$i -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra "$@" > /tmp/pmc_$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py counters $(find /tmp/pmc_* -name "*.db" | sort) > $out/${tag}_pmc_summary.csv
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra "$@" > $out/${tag}_bench_profiled.json 2> /tmp/kt.err
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py kernels $(find /tmp/kt -name "*.db" | head -1) > $out/${tag}_kernel_stats.csv
tail -1 $out/${tag}_bench_profiled.json | cut -c1-300
grep "poa_consensus\|ssw_align" $out/${tag}_pmc_summary.csv
head -8 $out/${tag}_kernel_stats.csv
