#!/bin/bash
# SQ counters of rq_forward_kernel at the C2 shape (separate --pmc passes, kernel-trace only).
#   gpurun --timeout 300 -- 'bash tools/pmc_forward.sh'
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/pmc_fwd"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o pmc -- python "$REPO/tools/pmc_forward.py" > /dev/null 2> "$OUT/p$i.err"
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "rq_forward_kernel" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in acc.items():
    v = sorted(d.values()); print(f"{c:28s} median per launch {v[len(v)//2]:.4g}  (n={len(v)})")
PY
done
