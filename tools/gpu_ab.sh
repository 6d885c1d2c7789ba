#!/bin/bash
# One gpurun call of a development round: tests of gemm_split, the step, and the kernel durations of one profiled bench run.
O=gpurun_out/ab; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_modules.py -x -q > $O/pytest_gemm.log 2>&1; grep -E "passed|failed|error" $O/pytest_gemm.log | tail -3
for i in 1 2; do
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity --min-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['launch_ms_mean'], d['breakdown_ms'])"
done | tee $O/bench_step.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_step -o p -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --min-seconds 0 > /dev/null 2>&1
python - $R/$O/prof_step <<'PY' | tee $R/$O/step_kernels.log
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:64], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
find $R/$O/prof_step -name "*trace.csv" -delete
