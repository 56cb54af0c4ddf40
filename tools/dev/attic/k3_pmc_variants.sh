#!/bin/bash
# chosen PMC counters for K3 builds: tools/dev/k3_pmc_variants.sh "CTR1 CTR2 .." name1 name2 ...  (base = libclh.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ctr=$1; shift
for v in "$@"; do
  rm -rf /tmp/pw_$v
  lib=$R/ciri_long_amd/libclh_$v.so; [ $v = base ] && lib=$R/ciri_long_amd/libclh.so
  CLH_LIB=$lib rocprofv3 --pmc $ctr -d /tmp/pw_$v -o run -- python3 $R/tools/ccs_bench.py 100000 > /tmp/pw_$v.log 2>&1
  echo "$v $(python3 $R/tools/rocpd_summary.py counters $(find /tmp/pw_$v -name '*.db') | grep poa | sed 's/.*CcsParams),//' | tr '\n' ' ')"
done
