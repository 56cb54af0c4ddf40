#!/bin/bash
# what K3's waves wait for (round 5): instruction-cache, fetch and issue counters of poa_consensus_kernel on the C3 batch, separate --pmc passes.
# usage (GPU box): bash tools/dev/k3_issue_counters.sh > gpurun_out/k3_issue_counters.txt
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_REQ" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH" "SQC_TC_STALL SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1)); rm -rf /tmp/ki_$i
  rocprofv3 --pmc $grp --kernel-include-regex "poa_consensus_kernel" -d /tmp/ki_$i -o run -- python3 $GRAFT_REPO_ROOT/tools/ccs_bench.py 100000 > /tmp/ki_$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py counters $(find /tmp/ki_* -name "*.db" | sort) | grep "poa_consensus_kernel"
