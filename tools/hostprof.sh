#!/bin/bash
# Host cycle profile of the native region loop on the GPU box: builds libplat_caller.so with -DPLAT_HOSTPROF (per-thread cycle counters of
# named scopes, caller_common.hpp), runs the WGS share once with PLAT_CALLER_TRACE=1, prints the last timed run's scopes, rebuilds the
# product library.   gpurun --timeout 600 -- 'bash tools/hostprof.sh [workers]'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/platypus_amd
cp libplat_caller.so /tmp/libplat_caller_keep.so
g++ -O3 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -fvisibility=hidden -DPLAT_HOSTPROF csrc/host/region_caller.cpp -o libplat_caller.so -L. -lplat_mi355x -Wl,-rpath,'$ORIGIN' || exit 1
cd $R
PLAT_CALLER_TRACE=1 PLAT_CALLER_WORKERS=${1:-16} python bench.py --config 4 --steps 3 --no-cpu-baseline > gpurun_out/hostprof.json 2> gpurun_out/hostprof.err
cp /tmp/libplat_caller_keep.so platypus_amd/libplat_caller.so
grep -a "prof\]\|per region" gpurun_out/hostprof.err | tail -${2:-40}
