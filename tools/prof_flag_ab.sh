#!/bin/bash
# usage: prof_flag_ab.sh "<bench flags of arm B>"  -- rocprofv3 kernel stats of the default step and of the step with the flags, top kernels side by side
F=$1; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for arm in A B; do
  fl=""; [ $arm = B ] && fl="$F"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_$arm -o t -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-small-batch --no-strict --min-seconds 0 $fl > /dev/null 2>&1
done
python - $R/gpurun_out/pf_A $R/gpurun_out/pf_B <<'PY'
import csv, glob, sys
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load(sys.argv[1]), load(sys.argv[2])
names = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0, 0))[2] + b.get(n, (0, 0, 0))[2]))
print(f"{'kernel':64} {'calls':>5} {'A avg_us':>9} {'B avg_us':>9} {'A tot_ms':>9} {'B tot_ms':>9}")
for n in names[:30]:
    x, y = a.get(n, (0, 0, 0)), b.get(n, (0, 0, 0))
    print(f"{n.replace('void ', '').replace('rqhip::', '')[:64]:64} {x[0]:5d} {x[1]:9.1f} {y[1]:9.1f} {x[2]:9.2f} {y[2]:9.2f}")
PY
rm -rf $R/gpurun_out/pf_A $R/gpurun_out/pf_B
