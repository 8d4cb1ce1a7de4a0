#!/bin/bash
# build the libraries, regenerate + compile the run-time kernels of the C3 plan offline (tools/rtc_check.cpp) and
# summarise their resource usage and the memory-instruction skeleton of one of them:  tools/rtc_isa.sh [k_compact]
set -e
cd "$(dirname "$0")/.."
make -s -j16 -C aresdb_amd/csrc 2>&1 | grep -E "error" -A6 | head -30 || true
mkdir -p tools/bin /tmp/rtc
[ tools/bin/rtc_check -nt tools/rtc_check.cpp -a tools/bin/rtc_check -nt aresdb_amd/lib/libalgorithm.so ] || hipcc --offload-arch=gfx950 -O1 -std=c++17 -Iinclude -Iaresdb_amd/csrc/algo -o tools/bin/rtc_check tools/rtc_check.cpp -Laresdb_amd/lib -lalgorithm -lhiprtc -Wl,-rpath,$PWD/aresdb_amd/lib
tools/bin/rtc_check /tmp/rtc/k > /tmp/rtc/log 2>&1 || true
echo "kernels compiled: $(grep -c "compile rc 0" /tmp/rtc/log)"; grep -B2 -A12 "error:" /tmp/rtc/log | head -40
for f in k k_compact k_merge k_cmerge k_table; do echo -n "$f: "; /opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/rtc/$f.co | grep -E "vgpr_count|sgpr_spill|private_seg|group_seg" | tr -s ' \n' ' '; echo; done
K=${1:-k_compact}
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 /tmp/rtc/$K.co > /tmp/rtc/$K.s 2>/dev/null
grep -n "global_load_dwordx4\|s_barrier\|s_waitcnt vmcnt\|global_store" /tmp/rtc/$K.s | awk -F'//' '{print $1}' | awk '{print $1, $2, $3, $4}' | tr '\n' ';' | cut -c1-3000; echo
