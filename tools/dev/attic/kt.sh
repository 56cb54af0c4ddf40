#!/bin/bash
# usage: tools/dev/kt.sh <tag> <python script> [args]  -- rocprofv3 kernel trace of one script, per-kernel table to gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/kt_$tag.log 2>&1
tail -2 /tmp/kt_$tag.log
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py kernels $(find /tmp/kt_$tag -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
head -14 $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
