#!/bin/bash
# usage (GPU box): tools/ab_slice96.sh -- the 96-channel slices of FTC_OP_MBHEAD (stage 6 at batch 8: 256 instead of 192 workgroups) against 128 everywhere
OUT=gpurun_out/r5j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_ops.py -q -x -k "mbconv_slice or train_mode_forward" 2>&1 | tail -8
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-fp32 --no-sustained --no-seam2 --no-configs"
$B --dump-ops $OUT/ops_96.json > $OUT/b96.json 2> $OUT/b96.err
FTC_MBSLICE_96=0 $B --dump-ops $OUT/ops_128.json > $OUT/b128.json 2> $OUT/b128.err
$B > $OUT/b96b.json 2>/dev/null
python - <<'P'
import json
for f in ["b96","b128","b96b"]:
    try:
        j=json.loads(open(f"gpurun_out/r5j/{f}.json").read().strip().splitlines()[-1]); print(f, j["value"], j["single_stream"], j.get("forward_ms_sum_of_kernels"))
    except Exception as e: print(f,"ERR",e)
for f in ["96","128"]:
    o=json.load(open(f"gpurun_out/r5j/ops_{f}.json"))
    for k,v in o["by_kernel"].items():
        if "mbconv" in k or "se_gate" in k or "64x64,bk=64,nbuf=2" in k: print("   ",f,k, round(v["ms"],3), v["launches"], round(1000*v["ms"]/v["launches"],1))
P
timeout 600 python -m pytest tests/test_gpu_detector.py -q -x -k "bf16 or fp16" 2>&1 | tail -3
