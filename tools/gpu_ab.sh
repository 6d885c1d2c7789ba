#!/bin/bash
# One gpurun call of a development round: the tests of the kernel that changed, then kernel durations (rocprofv3
# --kernel-trace --stats) of the product build vs tools/_ab builds.   gpurun --timeout 600 -- 'bash tools/gpu_ab.sh'
O=gpurun_out/ab; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; grep -E "passed|failed|error" $O/pytest_all.log | tail -3
for lib in "" tools/_ab/librqhip_bwdold.so; do
  timeout 120 python tools/bench_kernels.py bwd --reps 50 ${lib:+--lib $lib} 2>&1 | grep -E "bwd|library"
done | tee $O/bench_bwd.log
cd /tmp && export TMPDIR=/tmp
for lib in product tools/_ab/librqhip_bwdold.so; do
  for shape in 100000,32,256,3 1048576,32,256,3 125000,32,1024,4; do
    d=$R/$O/prof_$(basename $lib .so)_${shape%%,*}
    timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/tools/bench_kernels.py bwd --one $shape --reps 30 $([ $lib = product ] || echo --lib $R/$lib) > /dev/null 2>&1
    python - $d "$lib $shape" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rq_backward" in r["Name"] or "cbgrad" in r["Name"]:
            print(sys.argv[2], "|", r["Name"][:60], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "min_us", round(float(r["MinNs"]) / 1e3, 2))
PY
    find $d -name "*trace.csv" -delete
  done
done | tee $R/$O/kernel_durations.log
