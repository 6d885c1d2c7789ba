#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh on the GPU box) into the committed evidence:
   profiles/<tag>_bench_kernel_stats.csv, <tag>_bench_kernel_stats_summary.txt, <tag>_bench_n1.json,
   <tag>_pmc_traffic_c2.json (the file bench.py reads `roofline.traffic` from)
usage: python tools/summarize_profile.py [tag, default r03] [c2 | c4]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
CFG = sys.argv[2] if len(sys.argv) > 2 else "c2"
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{TAG}" + ("" if CFG == "c2" else f"_{CFG}"))
PRE = TAG if CFG == "c2" else f"{TAG}_{CFG}"     # file-name prefix under profiles/
DST = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True))
    if not hits:
        raise SystemExit(f"missing {pattern} under {SRC}")
    return hits[-1]


sha_file = os.path.join(SRC, "librqhip.sha256")
lib_sha = open(sha_file).read().split()[0] if os.path.exists(sha_file) else None
bench = json.loads(open(os.path.join(SRC, "bench_n1.json")).read().strip().splitlines()[-1])
with open(os.path.join(DST, f"{PRE}_bench_n1.json"), "w") as f:
    f.write(json.dumps(bench) + "\n")

stats = one("stats/**/*kernel_stats.csv")
shutil.copyfile(stats, os.path.join(DST, f"{PRE}_bench_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(DST, f"{PRE}_bench_kernel_stats_summary.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats of `python bench.py --config {CFG} --steps {20 if CFG == 'c2' else 3} --warmup 3 --no-cpu-baseline --no-parity` (MI355X, {TAG}; librqhip.so sha256 {str(lib_sha)[:16]})\n")
    f.write(f"# bench line of the same build: profiles/{PRE}_bench_n1.json ({bench['value'] / 1e6:.2f} M items/s, "
            f"{bench['ms_per_step']:.2f} ms/step; roofline launch mean {bench['roofline_rq']['launch_ms_mean'] * 1e3:.1f} us by HIP events)\n")
    f.write(f"# top kernels by total time; names shortened; full CSV: {PRE}_bench_kernel_stats.csv\n\n")
    f.write(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}\n")
    for r in rows[:40]:
        f.write(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} "
                f"{float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f}\n")
    ours = [r for r in rows if "rqhip::" in r["Name"]]
    f.write("\n# hand-written kernels (librqhip.so)\n")
    for r in ours:
        f.write(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} "
                f"{float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f}\n")

cfg_name = bench["config"].get("name", "c2")
pmc = {"librqhip_sha256": lib_sha,   # of the library the counters were collected on (bench.py checks it before using them)
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 5 "
                  "--warmup 2 --no-cpu-baseline --no-parity   (tools/profile_bench.sh)",
       "note": "KB per launch; max = the B=100000 launches (mean includes the 20000-row k-means warm-up launch). gfx950 "
               "correction from MI355X_MICROARCH.md (HBM section): FETCH_SIZE under-reports coalesced streaming reads by "
               "2x -> doubled by the consumer (bench.py); WRITE_SIZE uncorrected.",
       "kernels": {}}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    path = one(f"pmc_{counter}/**/*counter_collection.csv")
    per = {}
    for r in csv.DictReader(open(path)):
        if "rqhip::" not in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        per.setdefault(r["Kernel_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])  # one row per XCD/dimension: sum
    for name, d in per.items():
        vals = list(d.values())
        k = pmc["kernels"].setdefault(name, {})
        k[f"{counter}_KB_max"] = max(vals)
        k[f"{counter}_KB_mean"] = sum(vals) / len(vals)
        k["launches_fetch" if counter == "FETCH_SIZE" else "launches_write"] = len(vals)
# ---- HBM bytes per launch SHAPE: the dispatches between the two marker launches of bench.py --pmc-window, matched in order to the tags ----
MAIN = ("gemm_f16_kernel", "gemm_split_kernel", "wgrad_split_kernel", "wgrad_split_jobs_kernel", "wgrad_split_jobs_mixed_kernel", "wgrad_kernel<",
        "wgrad_jobs_kernel")
shape_bytes, shape_note = {}, []
for counter, weight in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):     # gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)
    tags_file = os.path.join(SRC, f"window_tags_{counter}.json")
    if not os.path.exists(tags_file):
        shape_note.append(f"{counter}: no window_tags file")
        continue
    tags = json.load(open(tags_file))["tags"]
    disp = {}
    for r in csv.DictReader(open(one(f"pmc_{counter}/**/*counter_collection.csv"))):
        if "rqhip::" not in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r.get("Grid_Size", 0) or 0), "kb": 0.0})
        d["kb"] += float(r["Counter_Value"])
    order = [disp[k] for k in sorted(disp)]
    marks = [i for i, d in enumerate(order) if "maxima_kernel" in d["name"] and d["grid"] == 256]
    if len(marks) < 2:
        shape_note.append(f"{counter}: markers not found ({len(marks)})")
        continue
    win = order[marks[-2] + 1:marks[-1]]
    seq, cur = [], None
    for d in win:
        if any(m in d["name"] for m in MAIN):
            cur = {"kb": d["kb"], "name": d["name"]}
            seq.append(cur)
        elif ("wgrad_reduce_kernel" in d["name"] or "wgrad_reduce_jobs_kernel" in d["name"]) and cur is not None:
            cur["kb"] += d["kb"]               # the partial-block reduction belongs to the weight gradient before it
    if len(seq) != len(tags):
        shape_note.append(f"{counter}: {len(seq)} matrix dispatches in the window vs {len(tags)} tags")
        continue
    per = {}
    for d, (kind, fl, by) in zip(seq, tags):
        per.setdefault(f"{kind}:{int(fl)}:{int(by)}", []).append(d["kb"] * 1024.0)
    for k, v in per.items():
        shape_bytes[k] = shape_bytes.get(k, 0.0) + weight * sum(v) / len(v)
    shape_note.append(f"{counter}: {len(seq)} dispatches matched")
ok_shapes = all("matched" in n for n in shape_note) and len(shape_note) == 2
with open(os.path.join(DST, f"{TAG}_pmc_kernels_{cfg_name}.json"), "w") as f:
    json.dump({"librqhip_sha256": lib_sha,
               "what": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes) of every matrix-kernel launch shape of one "
                       "training step; key = 'kind:algorithmic flops:algorithmic bytes' (the tags of rqhip_profile_*); a weight gradient includes "
                       "its partial-block reduction launches; bench.py --pmc-window brackets the step, tools/summarize_profile.py matches in order",
               "matching": shape_note, "kernels": shape_bytes if ok_shapes else {}}, f, indent=1)
    f.write("\n")
print("per-shape traffic:", shape_note)
for k, v in sorted(shape_bytes.items(), key=lambda kv: -kv[1]):
    kind, fl, by = k.split(":")
    print(f"  {kind:12s} {float(fl) / 1e9:8.2f} GFLOP  algorithmic {float(by) / 1e6:8.1f} MB  PMC {v / 1e6:8.1f} MB  ratio {v / float(by):.3f}")

# the training step's quantisation launch: the fused seam kernel where the step takes it (rq_seam_kernel<1> / <2>: the modes with levels
# behind a GEMM; <0> also carries the bare-GEMM launches of the backward), else the forward kernel with the most launches
_have = lambda k: "FETCH_SIZE_KB_max" in pmc["kernels"][k] and "WRITE_SIZE_KB_max" in pmc["kernels"][k]   # noqa: E731
fwk = [k for k in pmc["kernels"] if ("rq_seam_kernel<1>" in k or "rq_seam_kernel<2>" in k) and _have(k)] if bench.get("roofline_rq", {}).get("fused_seam") else []
fwk = fwk or [k for k in pmc["kernels"] if "rq_forward_kernel" in k and _have(k)]
if fwk:
    v = pmc["kernels"][fwk[0]]
    rows = bench["config"]["micro_batch_rows"]
    corrected = (2 * v["FETCH_SIZE_KB_max"] + v["WRITE_SIZE_KB_max"]) * 1024.0
    algorithmic = bench["roofline_rq"]["hbm_view"]["algorithmic_bytes_per_row"] * rows
    pmc["rq_forward_kernel"] = {"kernel": fwk[0][:60], "rows_per_launch": rows, "hbm_bytes_per_launch_corrected": corrected,
                                "algorithmic_bytes_per_launch": algorithmic, "ratio": corrected / algorithmic}
with open(os.path.join(DST, f"{TAG}_pmc_traffic_{cfg_name}.json"), "w") as f:
    json.dump(pmc, f, indent=1)
    f.write("\n")
# per-dispatch durations of the forward kernel: launches INSIDE training steps (milliseconds apart) vs back-to-back ones
disp = os.path.join(SRC, "rq_forward_dispatches.json")
if os.path.exists(disp):
    import statistics
    alld = json.load(open(disp))
    names = sorted({r["kernel"] for r in alld}, key=lambda n: -sum(r["kernel"] == n for r in alld))
    d = [r for r in alld if r["kernel"] == names[0]]          # the training step's forward kernel (most launches)
    in_step, back_to_back = [], []
    for prev, cur in zip([None] + d[:-1], d):
        gap = cur["start_us"] - prev["start_us"] if prev and "start_us" in cur else 1e9
        (back_to_back if gap < 1000.0 else in_step).append(cur["us"])
    out = {"kernel": names[0], "source": "rocprofv3 --kernel-trace of the bench run",
           "in_step": {"launches": len(in_step), "median_us": statistics.median(in_step) if in_step else None,
                       "min_us": min(in_step) if in_step else None, "max_us": max(in_step) if in_step else None,
                       "note": "first launch of a burst: preceded by other kernels of the training step (or by the warm-up)"},
           "back_to_back": {"launches": len(back_to_back),
                            "median_us": statistics.median(back_to_back) if back_to_back else None},
           "hip_event_mean_us_on_the_bench_line_of_the_same_run": None}
    try:
        br = json.loads(open(os.path.join(SRC, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
        out["hip_event_mean_us_on_the_bench_line_of_the_same_run"] = br["roofline_rq"]["launch_ms_mean"] * 1e3
    except Exception:  # noqa: BLE001
        pass
    with open(os.path.join(DST, f"{PRE}_rq_forward_dispatches.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("forward kernel dispatches:", json.dumps(out))
fw = [k for k in pmc["kernels"] if "rq_forward_kernel" in k or "rq_seam_kernel" in k]
for k in fw:
    v = pmc["kernels"][k]
    print(k, "-> traffic per launch (2*FETCH+WRITE) =", (2 * v["FETCH_SIZE_KB_max"] + v["WRITE_SIZE_KB_max"]) * 1024 / 1e6, "MB")
print(f"wrote profiles/{PRE}_*")
