#!/bin/bash
# usage: ab_lib.sh <other.so> [reps]  -- alternates bench runs of the in-tree library and another build
L=$1; N=${2:-2}
for i in $(seq $N); do
for lib in "" "--lib $L"; do
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-parity --no-strict --no-small-batch $lib 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench [$lib]', d['value'], d['ms_per_step'], d['roofline']['family_ms_per_step'], d['roofline']['frac'])"
done; done
