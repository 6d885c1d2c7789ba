#!/bin/bash
# One gpurun call of a round-4 development cycle: the GEMM / weight-gradient / module tests, two short bench runs (product and an
# A/B arm), and the kernel durations of one profiled bench run.   usage: bash tools/gpu_r4.sh <tag> [pytest args...]
TAG=${1:-dev}; shift
O=gpurun_out/$TAG; mkdir -p $O; R=$GRAFT_REPO_ROOT
TESTS=${@:-tests/test_gpu_gemm_split.py tests/test_gpu_wgrad.py tests/test_gpu_modules.py}
timeout 900 python -m pytest $TESTS -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for arm in split split6; do
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-parity --min-seconds 0 --mlp $arm 2>$O/bench_$arm.err | tail -1 > $O/bench_$arm.json
python - $O/bench_$arm.json $arm <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["launch_ms_mean"], d["breakdown_ms"])
    rk = d["roofline_kernels"]
    print("  matrix kernels:", rk["matrix_kernels"], "recorded us/step", rk["recorded_us_per_step"])
    for e in rk["kernels"]:
        print("  ", e["kernel"], e["calls_per_step"], "x", e["mean_us"], "us  GF", e["algorithmic_gflop"], "MB", e["algorithmic_mb"], "TF", e["achieved_tflops"],
              "issued frac", e.get("frac_vs_issued_dtype_peak"), "hbm frac", e["frac_vs_hbm_peak"])
except Exception as ex:
    print(sys.argv[2], "bench failed:", ex)
    print(open(sys.argv[1].replace(".json", ".err")).read()[-3000:])
PY
done 2>&1 | tee $O/bench_step.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_step -o p -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --min-seconds 0 > /dev/null 2>&1
python - $R/$O/prof_step <<'PY' | tee $R/$O/step_kernels.log
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        print(r["Name"][:90], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "total_ms", round(float(r["TotalDurationNs"]) / 1e6, 2))
PY
find $R/$O/prof_step -name "*trace.csv" -delete
