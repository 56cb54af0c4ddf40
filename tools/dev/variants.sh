#!/bin/bash
# build libclh_<name>.so variants of one kernel file with extra -D flags:
#   [FILE=ccs_poa] tools/dev/variants.sh name1 "flags1" name2 "flags2" ...
cd "$(dirname "$0")/../../ciri_long_amd/csrc"
FILE=${FILE:-ccs_poa}
ALL="clh_api ssw_prefilter ssw_scan ssw_scan_wide ssw_lanes ssw_traceback ssw_traceback_rows ccs_poa edit_distance genome splice_scan fastx_ccs"
OBJS="ssw_wavefront.p0.o ssw_wavefront.p1.o ssw_wavefront.p2.o ssw_wavefront.p3.o"
for f in $ALL; do [ $f != $FILE ] && OBJS="$OBJS $f.o"; done
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $f -c $FILE.hip -o /tmp/$FILE.$n.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libclh_$n.so /tmp/$FILE.$n.o $OBJS -lz -lpthread ) &
done
wait
ls ../libclh_*.so
