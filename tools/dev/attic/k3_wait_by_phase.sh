#!/bin/bash
# wave-time accounting of K3 per phase: phase-doubling builds (tools/dev/variants.sh dp2/bt2/sort2/hb2) under --pmc; differences to base
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base dp2 bt2 sort2 hb2; do
  rm -rf /tmp/pw_$v
  lib=$R/ciri_long_amd/libclh_$v.so; [ $v = base ] && lib=$R/ciri_long_amd/libclh.so
  CLH_LIB=$lib rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d /tmp/pw_$v -o run -- python3 $R/tools/ccs_bench.py 100000 > /tmp/pw_$v.log 2>&1
  echo "$v $(python3 $R/tools/rocpd_summary.py counters $(find /tmp/pw_$v -name '*.db') | grep poa | sed 's/.*CcsParams),//' | tr '\n' ' ')"
done
