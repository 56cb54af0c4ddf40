#!/bin/bash
# libclh_dbg.so with only the RV=1 and row-strip classes of K1 (fast to build); extra -D flags may be passed
set -e
cd "$(dirname "$0")/../ciri_long_amd/csrc"
for f in clh_api ssw_wavefront ssw_traceback ccs_poa edit_distance genome fastx_ccs; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCLH_STRIPS_BUILD "$@" -c $f.hip -o /tmp/$f.dbg.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libclh_dbg.so /tmp/clh_api.dbg.o /tmp/ssw_wavefront.dbg.o /tmp/ssw_traceback.dbg.o /tmp/ccs_poa.dbg.o /tmp/edit_distance.dbg.o /tmp/genome.dbg.o /tmp/fastx_ccs.dbg.o -lz -lpthread
