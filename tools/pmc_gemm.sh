#!/bin/bash
# SQ counters of gemm_split_kernel at the two 78.6 GFLOP layers (separate --pmc passes, kernel-trace only).
#   gpurun --timeout 400 -- 'bash tools/pmc_gemm.sh'
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/pmc_gemm"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o pmc -- python "$REPO/tools/pmc_gemm.py" > /dev/null 2> "$OUT/p$i.err"
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { tail -3 "$OUT/p$i.err"; continue; }
  t=$(find "$OUT/p$i" -name "*kernel_trace.csv" | head -1)
  python - "$f" "$t" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_split_kernel" not in r["Kernel_Name"]: continue
    kind = "relu 768->512" if "<true>" in r["Kernel_Name"] else "plain 512->768"
    acc[(kind, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
for (kind, c), d in sorted(acc.items()):
    v = sorted(d.values()); print(f"{kind:15s} {c:28s} median per launch {v[len(v)//2]:.5g}  (n={len(v)})")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if "gemm_split_kernel" in r["Kernel_Name"]:
        dur["relu 768->512" if "<true>" in r["Kernel_Name"] else "plain 512->768"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    v.sort(); print(f"{k:15s} duration under this pass: median {v[len(v)//2]:.1f} us")
PY
done
find "$OUT" -name "*.csv" -size +2M -delete
