#!/bin/bash
# A/B builds of one source file: tools/ab_build.sh <name> <file.hip> [-DMACRO ...] -> csrc/libqldpc_hip_<name>.so (the timers objects of the
# other sources + this file compiled with the timers flag and the given macros).  Load it with --build libqldpc_hip_<name>.so in tools/kbench*.py.
set -euo pipefail
cd "$(dirname "$0")/../qldpc-branched-off_amd/csrc"
name=$1; src=$2; shift 2
make -s timers >/dev/null
mkdir -p build/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-result -DQLDPC_OSD_TIMERS "$@" -c "$src" -o "build/ab/${name}_${src%.hip}.o"
objs=$(ls build/timers/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o "libqldpc_hip_${name}.so" $objs "build/ab/${name}_${src%.hip}.o" -ldl
echo "libqldpc_hip_${name}.so"
