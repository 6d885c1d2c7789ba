#!/bin/bash
# rocprofv3 evidence for the MFMA-vs-LDS/VALU clause of BASELINE.json's north_star: per-kernel times of
# the three forms of the forward scan at the C2 shape (100 000 x 3 x 256 x 32, STE), one process, --kernel-trace --stats.
#   gpurun --timeout 300 -- 'bash tools/profile_scan_ab.sh'     ->  gpurun_out/scan_ab/ ; summary copied to profiles/ by hand
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/scan_ab"; mkdir -p "$OUT"
sha256sum "$REPO/rq-vae-recommender_amd/csrc/librqhip.so" > "$OUT/librqhip.sha256"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o ab -- \
    python "$REPO/tools/ab_scan.py" c2 c4_micro d64 > "$OUT/ab_under_rocprof.jsonl" 2> "$OUT/stats.err"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
python - "$f" "$OUT" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rq_forward" in r["Name"] or "rq_csq" in r["Name"]]
with open(sys.argv[2] + "/scan_ab_kernel_stats.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python tools/ab_scan.py c2 c4_micro d64   (MI355X; librqhip.so sha256 "
            + open(sys.argv[2] + "/librqhip.sha256").read().split()[0][:16] + ")\n")
    f.write("# template arguments: <KSTEPS, MODE, FULLD, threads, MARGIN, FILT, RESIDENT>; FILT = filtered bf16-split scan,\n"
            "# otherwise the all-fp32 MFMA scan; rq_forward_valu_kernel = LDS / VALU scan (no matrix instruction)\n")
    f.write(f"{'kernel':88s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}\n")
    for r in rows:
        f.write(f"{r['Name'][:88]:88s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} {float(r['MinNs'])/1e3:9.1f} "
                f"{float(r['MaxNs'])/1e3:9.1f}\n")
print(open(sys.argv[2] + "/scan_ab_kernel_stats.txt").read())
PY
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -type f -size +4M -delete
cat "$OUT/ab_under_rocprof.jsonl"
