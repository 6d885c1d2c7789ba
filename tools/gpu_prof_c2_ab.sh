#!/bin/bash
# rocprofv3 kernel stats of the C2 step, seam on and off (same box)
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out/prof_c2ab_r06"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for arm in seam noseam; do
  flag=""; [ $arm = noseam ] && flag="--no-seam"
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$arm" -o c2 -- python "$REPO/bench.py" --steps 40 --warmup 3 --no-cpu-baseline --no-parity --no-small-batch --no-strict --min-seconds 0 $flag > "$OUT/$arm.json" 2> "$OUT/$arm.err"
  f=$(find "$OUT/$arm" -name "*kernel_stats.csv" | head -1)
  echo "== $arm"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = ("kmeans", "copyBuffer")
for r in rows[:45]:
    if any(k in r['Name'] for k in skip): continue
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
  find "$OUT/$arm" -name "*kernel_trace.csv" -delete
done
