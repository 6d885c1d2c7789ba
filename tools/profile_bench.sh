#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'TAG=r03 [CFG=c4] bash tools/profile_bench.sh'
# Pass 1: --kernel-trace --stats of the bench command (no CPU baseline / parity / secondary legs).  Passes 2, 3: one PMC counter each
# (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only -- never combined with sys/hip traces.  Outputs under
# gpurun_out/prof_$TAG/; tools/summarize_profile.py turns them into the files committed under profiles/.
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
TAG="${TAG:-r06}"
CFG="${CFG:-c2}"   # c2 | c4
OUT="$REPO/gpurun_out/prof_${TAG}$([ $CFG = c2 ] || echo _$CFG)"
mkdir -p "$OUT"
sha256sum "$REPO/rq-vae-recommender_amd/csrc/librqhip.so" > "$OUT/librqhip.sha256"
cd /tmp && export TMPDIR=/tmp
# (a short line first: tools/summarize_profile.py wants one; the DEFAULT line is taken at the end, after the counter passes have been
# summarised on this box, so that its `roofline.traffic` finds the PMC file of the very library it loaded)
timeout -k 5 200 python "$REPO/bench.py" --config $CFG --no-cpu-baseline --no-parity --no-small-batch --no-strict > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
    python "$REPO/bench.py" --config $CFG --steps $([ $CFG = c2 ] && echo 20 || echo 3) --warmup 3 --no-cpu-baseline --no-parity --no-small-batch --no-strict --min-seconds 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- \
        python "$REPO/bench.py" --config $CFG --steps $([ $CFG = c2 ] && echo 5 || echo 1) --warmup 2 --no-cpu-baseline --no-parity --no-small-batch --no-strict --min-seconds 0 \
            --pmc-window "$OUT/window_tags_$c.json" > /dev/null 2> "$OUT/pmc_$c.err"
done
# keep what summarize_profile.py reads (gpurun copies back at most 64 MiB): the stats table, and the counter rows
# of the hand-written kernels; drop the per-dispatch traces
for f in $(find "$OUT" -name "*counter_collection.csv"); do
    { head -1 "$f"; grep "rqhip::" "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"
done
# per-dispatch durations of the forward kernel (the bench line's HIP-event mean is checked against these), then drop the trace
for f in $(find "$OUT/stats" -name "*kernel_trace.csv"); do
    python - "$f" "$OUT/rq_forward_dispatches.json" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rq_forward_kernel" in r["Kernel_Name"] or "rq_seam_kernel<1>" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [{"kernel": r["Kernel_Name"][:80], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
      "start_us": int(r["Start_Timestamp"]) / 1e3, "grid": int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)} for r in rows]
json.dump(d, open(sys.argv[2], "w"))
PY
done
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" ! -name "*.sha256" -delete
find "$OUT" -name "*.csv" -size +8M -delete
( cd "$REPO" && python tools/summarize_profile.py $TAG $CFG > "$OUT/summarize_on_box.log" 2>&1 )
timeout -k 5 400 python "$REPO/bench.py" --config $CFG > "$OUT/bench_n1.json.new" 2> "$OUT/bench_n1.err"   # the default line: parity gate, CPU baseline, secondaries
[ -s "$OUT/bench_n1.json.new" ] && mv "$OUT/bench_n1.json.new" "$OUT/bench_n1.json"
du -sh "$OUT"; tail -3 "$OUT/stats.err"
find "$OUT" -name "*.csv" | sed "s#$REPO/##"
tail -c 600 "$OUT/bench_n1.json"
