#!/bin/bash
# Round 6 (end): SQ / TCP / TCC counters of the product GEMM (gemm_f16_kernel<1, 256, 2> 768 -> 512, static schedule, straight-line epilogue) and of the batched weight gradient (wgrad_split_jobs_kernel: dW[512, 768] + dW[256, 512], 8 tiles x 32 ranges); separate --pmc
# passes with --kernel-trace only.   gpurun --timeout 900 -- 'bash tools/pmc_matrix_r06.sh > gpurun_out/r05_pmc_matrix.txt 2>&1'
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/pmc_matrix_r06"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TA_BUSY_avr TD_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o pmc -- python "$REPO/tools/pmc_matrix_r06.py" > /dev/null 2> "$OUT/p$i.err"
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "pass $i ($set) failed:"; grep -i "error\|invalid\|not " "$OUT/p$i.err" | head -3; continue; }
  t=$(find "$OUT/p$i" -name "*kernel_trace.csv" | head -1)
  python - "$f" "$t" <<'PY'
import csv, sys, collections
def kind(n):
    return "gemm_f16<1> 768->512" if "gemm_f16_kernel<1," in n else "wgrad_jobs dW[512,768]+[256,512]" if "wgrad_split_jobs_kernel" in n else None
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = kind(r["Kernel_Name"])
    if k: acc[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    k = kind(r["Kernel_Name"])
    if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(dur.items()):
    v.sort(); print(f"{k:34s} duration under this pass: median {v[len(v)//2]:.1f} us (n={len(v)})")
for (k, c), d in sorted(acc.items()):
    v = sorted(d.values()); print(f"{k:34s} {c:32s} median per launch {v[len(v)//2]:.5g}")
PY
  rm -rf "$OUT/p$i"
done
