#!/bin/bash
# rocprofv3 kernel stats of the batch-640 / batch-64 steps (tools/bench_small_batch.py), seam on and off
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out/prof_small_r06"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for arm in seam noseam; do
  flag=""; [ $arm = noseam ] && flag="--no-seam"
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$arm" -o small -- python "$REPO/tools/bench_small_batch.py" --json $flag > "$OUT/$arm.json" 2> "$OUT/$arm.err"
  f=$(find "$OUT/$arm" -name "*kernel_stats.csv" | head -1)
  echo "== $arm"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
  find "$OUT/$arm" -name "*kernel_trace.csv" -delete
done
