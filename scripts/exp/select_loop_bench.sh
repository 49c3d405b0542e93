#!/bin/bash
# builds (if hipcc is there and the binary is missing or stale) and runs the host-only selection-loop timing
set -e
cd "$(dirname "$0")/../.."
BIN=scripts/exp/_select_loop_bench
if [ ! -x $BIN ] || [ m-loam_amd/csrc/select.hip -nt $BIN ] || [ m-loam_amd/csrc/alive_pool.hpp -nt $BIN ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Im-loam_amd/csrc scripts/exp/select_loop_bench.hip -o $BIN -Wl,--unresolved-symbols=ignore-all 2>/dev/null
fi
$BIN
