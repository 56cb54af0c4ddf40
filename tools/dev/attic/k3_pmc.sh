#!/bin/bash
# K3 under chosen PMC counters: tools/dev/k3_pmc.sh "CTR1 CTR2 ..." [more groups...]   (run on the GPU box; one rocprofv3 pass per group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "$@"; do
  i=$((i+1)); rm -rf /tmp/pc_$i
  rocprofv3 --pmc $grp -d /tmp/pc_$i -o run -- python3 $R/tools/ccs_bench.py 100000 > /tmp/pc_$i.log 2>&1
  python3 $R/tools/rocpd_summary.py counters $(find /tmp/pc_$i -name '*.db') | grep poa
done
