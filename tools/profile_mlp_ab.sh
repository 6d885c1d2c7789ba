#!/bin/bash
# rocprofv3 A/B of the C2 training step, same box, same process setup: MLP GEMMs + weight gradients as two fp16 pieces under
# exact power-of-two scales (`split`, the product), as three bf16 pieces (`split6`, round 3) and as library fp32 GEMMs +
# fp32-MFMA weight gradients (`library`, round 2).
#   gpurun --timeout 900 -- 'bash tools/profile_mlp_ab.sh'   ->  gpurun_out/mlp_ab/{split,split6,library}.{json,stats.csv}
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/mlp_ab"
mkdir -p "$OUT"
sha256sum "$REPO/rq-vae-recommender_amd/csrc/librqhip.so" > "$OUT/librqhip.sha256"
cd /tmp && export TMPDIR=/tmp
for arm in split split6 library split split6 library; do     # un-profiled wall clock first, alternating (clock state drifts)
    timeout -k 5 200 python "$REPO/bench.py" --mlp $arm --steps 100 --warmup 5 --no-cpu-baseline --no-parity --min-seconds 0 \
        >> "$OUT/$arm.json" 2>> "$OUT/$arm.err"
done
for arm in split split6 library; do
    timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$arm" -o ab -- \
        python "$REPO/bench.py" --mlp $arm --steps 20 --warmup 3 --no-cpu-baseline --no-parity --min-seconds 0 \
        > "$OUT/${arm}_under_rocprof.json" 2> "$OUT/prof_$arm.err"
    cp $(find "$OUT/prof_$arm" -name "*kernel_stats.csv" | head -1) "$OUT/$arm.stats.csv"
    rm -rf "$OUT/prof_$arm"
done
python - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
lines = [f"# C2 step, MLP arithmetic A/B on one MI355X box (tools/profile_mlp_ab.sh; librqhip.so sha256 {open(out + '/librqhip.sha256').read()[:16]})"]
for arm in ("split", "split6", "library"):
    runs = [json.loads(l) for l in open(f"{out}/{arm}.json") if l.startswith("{")]
    lines.append(f"{arm:8s} un-profiled, 100 steps each: " + ", ".join(f"{r['ms_per_step']:.3f} ms/step ({r['value'] / 1e6:.2f} M items/s, loss {r['final_loss']:.6f})" for r in runs))
for arm in ("split", "split6", "library"):
    rows = sorted(csv.DictReader(open(f"{out}/{arm}.stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))
    lines.append(f"\n== --mlp {arm}: rocprofv3 --kernel-trace --stats, 23 steps; top kernels ==")
    lines.append(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'total_ms':>9s} {'pct':>6s}")
    for r in rows[:16]:
        lines.append(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:9.1f} {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['Percentage']):6.2f}")
open(f"{out}/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
