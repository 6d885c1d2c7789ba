#!/bin/bash
# Developer A/B builds: recompile ONE kernel file with extra -D flags and link it with the product objects into
# tools/_ab/librqhip_<name>.so (git-ignored); a tool loads it with rqhip._lib.load(<path>) before its first op
# (tools/gemm_probe.py <path>, GS_LIB=<path> tools/pmc_gemm.py).
#   tools/ab_build.sh gsp32 gemm_split.hip -DGS_PROBE=32      # gemm_split without its A loads (bit mask in the file)
#   tools/ab_build.sh probe rq_backward.hip -DRQ_BWD_PROBE     # phase-skipping switches, $RQ_BWD_PROBE bit mask:
#        1 = no table read/add/write, 2 = no accumulation scan, 4 = no staging / barriers / accumulation,
#        8 = no partial-table flush (results are NOT correct with any bit set)
#   tools/ab_build.sh sb8 rq_forward.hip -DRQ_STAGE_BATCH=8     # codebook staging: 8 loads in flight per thread
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
mkdir -p tools/_ab
C=rq-vae-recommender_amd/csrc
make -s -C $C
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -munsafe-fp-atomics -Iinclude -I$C"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$SRC -o tools/_ab/${SRC%.hip}_$NAME.o
OBJS=$(ls $C/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/librqhip_$NAME.so tools/_ab/${SRC%.hip}_$NAME.o $OBJS
echo built tools/_ab/librqhip_$NAME.so
