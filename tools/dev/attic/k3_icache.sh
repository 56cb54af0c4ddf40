#!/bin/bash
# instruction-cache counters of K3 on the C3 step (tools/dev/k3_icache.sh): separate --pmc passes
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf /tmp/ic_$i
  rocprofv3 --pmc $grp --kernel-include-regex "poa_consensus_kernel" -d /tmp/ic_$i -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --steps 3 --warmup 1 > /tmp/ic_$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py counters $(find /tmp/ic_* -name "*.db" | sort) | grep "poa_consensus_kernel"
