#!/bin/bash
# builds libclh_dbg.so with the K3 phase clocks compiled in, then prints the breakdown
set -e
cd "$(dirname "$0")/../ciri_long_amd/csrc"
for f in clh_api ssw_prefilter ssw_scan ssw_scan_wide ssw_lanes ssw_wavefront ssw_traceback ssw_traceback_rows ccs_poa edit_distance genome splice_scan fastx_ccs; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCLH_DEBUG_POA -DCLH_PROBE_BUILD -c $f.hip -o /tmp/$f.dbg.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libclh_dbg.so /tmp/clh_api.dbg.o /tmp/ssw_prefilter.dbg.o /tmp/ssw_scan.dbg.o /tmp/ssw_scan_wide.dbg.o /tmp/ssw_lanes.dbg.o /tmp/ssw_wavefront.dbg.o /tmp/ssw_traceback.dbg.o /tmp/ssw_traceback_rows.dbg.o /tmp/ccs_poa.dbg.o /tmp/edit_distance.dbg.o /tmp/genome.dbg.o /tmp/splice_scan.dbg.o /tmp/fastx_ccs.dbg.o -lz -lpthread
cd ../..
[ -n "$BUILD_ONLY" ] && exit 0
python tools/k3_phases.py "$@"
rm -f ciri_long_amd/libclh_dbg.so
