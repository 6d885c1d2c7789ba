#!/bin/bash
# Developer A/B: tools/_ab/librqhip_head.so = the library of the last COMMIT (every csrc/*.hip / *.h that differs from HEAD is taken from HEAD),
# to alternate against the working tree's in-tree build (tools/ab_lib.sh tools/_ab/librqhip_head.so).
set -e
cd "$(dirname "$0")/.."
C=rq-vae-recommender_amd/csrc
make -s -C $C
T=tools/_ab/head_src; rm -rf $T; mkdir -p $T
git archive HEAD $C include | tar -x -C $T
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -munsafe-fp-atomics -I$T/include -I$T/$C"
OBJS=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if git diff --quiet HEAD -- $f $C/*.h include/rqhip.h; then OBJS="$OBJS $C/$b.o"; else
    /opt/rocm/bin/hipcc $FLAGS -c $T/$C/$b.hip -o tools/_ab/head_$b.o 2>&1 | grep -v hip-link || true
    OBJS="$OBJS tools/_ab/head_$b.o"; echo "rebuilt $b from HEAD"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/librqhip_head.so $OBJS
echo built tools/_ab/librqhip_head.so
