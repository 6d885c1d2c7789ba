#!/bin/bash
# usage: ab_flag.sh "<bench flags of arm B>" [reps]  -- alternates bench runs without / with the flags (same library)
F=$1; N=${2:-2}
for i in $(seq $N); do
for fl in "" "$F"; do
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-parity --no-strict --no-small-batch $fl 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench [$fl]', d['value'], d['ms_per_step'], d['roofline']['family_ms_per_step'], d['roofline']['frac'])"
done; done
