#!/bin/bash
# rocprofv3 kernel stats of the batch-640 / batch-64 steps (tools/bench_small_batch.py): the layers' own kernels (csrc/mlp_small.hip) on and off
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out/prof_small_r06"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for arm in ${ARMS:-small nosmall}; do
  flag=""; [ $arm = nosmall ] && flag="--no-small"
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$arm" -o small -- python "$REPO/tools/bench_small_batch.py" ${SHAPE:-} $flag > "$OUT/$arm.out" 2> "$OUT/$arm.err"
  f=$(find "$OUT/$arm" -name "*kernel_stats.csv" | head -1)
  echo "== $arm"; grep -E "^(eager|graph)" "$OUT/$arm.out"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 10 + 200 + 1 + 3 + 1 + 10 + 200      # eager warm-up + timed + profiled + graph warm-up + capture + replays of tools/bench_small_batch.py
print(f"{'kernel':100s} {'calls':>6s} {'avg us':>8s} {'us/step':>8s}")
tot = 0.0
for r in rows[:40]:
    per = float(r['TotalDurationNs']) / 1e3 / steps
    tot += per
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} {per:8.1f}")
print(f"sum of the rows above: {tot:.1f} us per step (approximate: {steps} steps assumed)")
PY
  find "$OUT/$arm" -name "*kernel_trace.csv" -delete
done
