#!/bin/bash
# Register / scratch / LDS usage per kernel of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage).
#   tools/kres.sh gemm_split.hip [-DGS_PHASE=0 ...]
cd "$(dirname "$0")/../rq-vae-recommender_amd/csrc"
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -fno-slp-vectorize -munsafe-fp-atomics -I../../include -I. -Rpass-analysis=kernel-resource-usage "$@" \
  -c $SRC -o /tmp/kres_$$.o 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur = {"fn": t.split(":", 1)[1].strip()[:70]}
    else:
        k, _, v = t.partition(":")
        if k.strip() in ("VGPRs", "AGPRs", "VGPR Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs"):
            cur[k.strip().split(" [")[0]] = v.strip()
if cur: print(cur)
'
rm -f /tmp/kres_$$.o
