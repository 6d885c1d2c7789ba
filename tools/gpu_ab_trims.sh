#!/bin/bash
# same box, alternating arms: the C2 step with / without round 6's overhead trims and with / without the seam path
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$REPO"
for i in 1 2 3; do
  for arm in "" "--no-trims" "--no-seam" "--no-seam --no-trims"; do
    python bench.py --steps 100 --no-cpu-baseline --no-parity --no-small-batch --no-strict --min-seconds 0 $arm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${arm:-round 6}'.ljust(22), d['value'], d['ms_per_step'], d['roofline']['family_ms_per_step'])"
  done
done
