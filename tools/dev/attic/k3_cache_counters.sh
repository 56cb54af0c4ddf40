#!/bin/bash
# K3's L1 / L2 behaviour (round 5): vector-L1 reads, what goes on to L2 and how long that takes, L2 hits and misses; separate --pmc passes.
# usage (GPU box): [CLH_LIB=.../libclh_x.so] bash tools/dev/k3_cache_counters.sh
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TOTAL_READ TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES" "TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_TOTAL_WRITE" "TCC_HIT TCC_MISS TCC_REQ" "TCP_TOTAL_WRITEBACK_INVALIDATES TCP_PENDING_STALL_CYCLES TCP_TCP_LATENCY" "TCC_EA0_RDREQ TCC_READ TCC_STREAMING_REQ"; do
  i=$((i+1)); rm -rf /tmp/kc_$i
  rocprofv3 --pmc $grp --kernel-include-regex "poa_consensus_kernel" -d /tmp/kc_$i -o run -- python3 $GRAFT_REPO_ROOT/tools/ccs_bench.py 100000 > /tmp/kc_$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py counters $(find /tmp/kc_* -name "*.db" | sort) | grep "poa_consensus_kernel"
