#!/bin/bash
# same box, alternating arms: the C2 step with the 128 x 512 tile on / off (bench.py --no-wide-tiles)
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$REPO"
for i in 1 2 3; do
  for arm in "" "--no-wide-tiles"; do
    python bench.py --steps 100 --no-cpu-baseline --no-parity --no-small-batch --no-strict --min-seconds 0 $arm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${arm:-wide}'.ljust(16), d['value'], d['ms_per_step'], d['roofline']['family_ms_per_step'], [ (k['kernel'],k['mean_us']) for k in d['roofline_kernels']['kernels'][:7]])"
  done
done
