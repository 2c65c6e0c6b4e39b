#!/bin/bash
# A/B builds of ONE source with extra -D flags: tools/ab_build.sh <source.hip> <tag> [-DFLAG ...] -> build/ab/libcmr_<tag>.so
# (development only; tools/*_bench.py --lib build/ab/libcmr_<tag>.so)
set -e
src=$1; tag=$2; shift 2
mkdir -p build/ab
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Icmr_agent_amd/csrc "$@" -c $src -o build/ab/${base}_${tag}.o
objs=$(ls build/*.o | grep -v "/${base}.o" | grep -v "/ab_")          # (the A/B library's own objects build/ab_*.o define the same symbols)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libcmr_${tag}.so $objs build/ab/${base}_${tag}.o
echo build/ab/libcmr_${tag}.so
