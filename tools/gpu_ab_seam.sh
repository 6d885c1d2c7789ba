#!/bin/bash
# same box, alternating arms: the C2 step and the small-batch steps with the seam kernel on / off
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$REPO"
for i in 1 2 3; do
  for arm in "" "--no-seam"; do
    python bench.py --steps 100 --no-cpu-baseline --no-parity --no-small-batch --no-strict --min-seconds 0 $arm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${arm:-seam}'.ljust(10), d['value'], d['ms_per_step'], d['roofline']['family_ms_per_step'], d['breakdown_ms'])"
  done
done
for i in 1 2; do
  for arm in "" "--no-seam"; do
    echo "small batch ${arm:-seam}"; python tools/bench_small_batch.py --json $arm 2>/dev/null | tail -1 | cut -c1-600
  done
done
