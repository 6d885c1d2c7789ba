#!/bin/bash
# SQ counters of wgrad_split_kernel at the 512 x 768 layer (separate --pmc passes, kernel-trace only).
#   gpurun --timeout 400 -- 'bash tools/pmc_wgrad.sh'
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/pmc_wgrad"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o pmc -- python "$REPO/tools/pmc_wgrad.py" > /dev/null 2> "$OUT/p$i.err"
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { tail -3 "$OUT/p$i.err"; continue; }
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "wgrad_split" not in r["Kernel_Name"]: continue
    kind = "masked" if "true>" in r["Kernel_Name"].replace(" ", "") or ", true" in r["Kernel_Name"] else "plain"
    acc[(kind, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
for (kind, c), d in sorted(acc.items()):
    v = sorted(d.values()); print(f"{kind:7s} {c:28s} median per launch {v[len(v)//2]:.4g}  (n={len(v)})")
PY
done
find "$OUT" -name "*.csv" -size +2M -delete
