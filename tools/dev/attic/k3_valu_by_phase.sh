#!/bin/bash
# VALU instructions of K3 per phase: the kernel with one phase run twice (tools/dev/variants.sh dp2/bt2/sort2/hb2) under
# rocprofv3 --pmc SQ_INSTS_VALU; the difference to the base build is that phase's count.  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base dp2 bt2 sort2 hb2; do
  rm -rf /tmp/pv_$v
  lib=$R/ciri_long_amd/libclh_$v.so; [ $v = base ] && lib=$R/ciri_long_amd/libclh.so
  CLH_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/pv_$v -o run -- python3 $R/tools/ccs_bench.py 100000 > /tmp/pv_$v.log 2>&1
  echo "$v $(python3 $R/tools/rocpd_summary.py counters $(find /tmp/pv_$v -name '*.db') | grep poa | tr '\n' ' ')  $(grep K3 /tmp/pv_$v.log)"
done
