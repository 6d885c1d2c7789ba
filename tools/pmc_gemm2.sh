#!/bin/bash
# clock and memory-path counters of gemm_split_kernel: product build vs the no-global-loads probe build
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/pmc_gemm2"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "Name:[[:space:]]*[A-Za-z0-9_]*\|^[[:space:]]*[A-Z][A-Z0-9_]*[a-z_]*" | head -0
for lib in "" tools/_ab/librqhip_gsp16.so; do
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TA_BUSY_avr TD_BUSY_avr TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  GS_LIB=$lib timeout -k 5 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o pmc -- python "$REPO/tools/pmc_gemm.py" > /dev/null 2> "$OUT/p$i.err"
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "set $i failed:"; grep -i "error\|invalid\|not" "$OUT/p$i.err" | head -3; continue; }
  t=$(find "$OUT/p$i" -name "*kernel_trace.csv" | head -1)
  python - "$f" "$t" "${lib:-product}" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_split_kernel" not in r["Kernel_Name"]: continue
    kind = "relu 768->512" if "<true>" in r["Kernel_Name"] else "plain 512->768"
    acc[(kind, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if "gemm_split_kernel" in r["Kernel_Name"]:
        dur["relu 768->512" if "<true>" in r["Kernel_Name"] else "plain 512->768"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    v.sort(); print(f"[{sys.argv[3][-12:]}] {k:15s} duration under this pass: median {v[len(v)//2]:.1f} us")
for (kind, c), d in sorted(acc.items()):
    v = sorted(d.values()); print(f"[{sys.argv[3][-12:]}] {kind:15s} {c:30s} {v[len(v)//2]:.5g}")
PY
  rm -rf "$OUT/p$i"
done
done
