#!/bin/bash
# One gpurun call of a development round: the tests of the kernels that changed, then timings.
#   gpurun --timeout 600 -- 'bash tools/gpu_ab.sh'
O=gpurun_out/ab; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_modules.py tests/test_gpu_train.py -x -q > $O/pytest_gemm.log 2>&1; grep -E "passed|failed|error" $O/pytest_gemm.log | tail -3
timeout 200 python tools/bench_gemm_split.py 100000 2>&1 | grep -v Warn | tee $O/bench_gemm.log
for i in 1 2; do
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity --min-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['launch_ms_mean'], d['breakdown_ms'])"
done | tee $O/bench_step.log
