#!/bin/bash
# Register / scratch / LDS usage of every kernel in the given translation units (default: all), from the compiler's
# resource-usage remarks.  Usage: tools/kernel_resources.sh [tu_k2x tu_k2s_ks13 ...]
cd "$(dirname "$0")/../microstructure_fingerprinting_amd/csrc" || exit 1
TUS=${@:-mfx_api tu_k2 tu_k2s_ks4 tu_k2s_ks8 tu_k2s_ks13 tu_k2s_ks16 tu_k2x}
for t in $TUS; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage \
    -c $t.hip -o /tmp/kres_$t.o 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill):\s+(\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur: print(cur)
        cur = {"kernel": v}
    else:
        cur[k.split(" [")[0]] = v
if cur: print(cur)
' | while read -r l; do echo "$l" | sed "s/^{//;s/}$//" | c++filt 2>/dev/null; done
done
