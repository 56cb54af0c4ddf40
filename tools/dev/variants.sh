#!/bin/bash
# build libclh_<name>.so variants of K3 with extra -D flags:  tools/dev/variants.sh name1 "flags1" name2 "flags2" ...
cd "$(dirname "$0")/../../ciri_long_amd/csrc"
OBJS="clh_api.o ssw_traceback.o edit_distance.o genome.o splice_scan.o fastx_ccs.o ssw_wavefront.p0.o ssw_wavefront.p1.o ssw_wavefront.p2.o ssw_wavefront.p3.o"
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $f -c ccs_poa.hip -o /tmp/ccs_poa.$n.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libclh_$n.so /tmp/ccs_poa.$n.o $OBJS -lz -lpthread ) &
done
wait
ls ../libclh_*.so
