#!/bin/bash
# usage (GPU box): tools/lanes_tuning.sh -- end-to-end two-lane / one-stream images/s of the bf16 batch-8 bench with alternative kernel choices for the
# MBConv project GEMMs (FTC_TUNING_OVERRIDE files, tools/ovr_old/*.txt or OVR_DIR): the tuning table picks the fastest kernel IN ISOLATION; under two lanes what
# counts is CU-time and L2->LDS bytes, so a bigger tile on fewer CUs may win although it loses alone.
OUT=gpurun_out/ovr; mkdir -p $OUT
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-fp32 --no-sustained --no-seam2 --no-configs"
$B --dump-ops $OUT/ops_base.json > $OUT/base.json 2> $OUT/base.err
for f in ${OVR_DIR:-tools/ovr_old}/*.txt; do n=$(basename $f .txt); FTC_TUNING_OVERRIDE=$f $B --dump-ops $OUT/ops_$n.json > $OUT/$n.json 2> $OUT/$n.err; done
$B > $OUT/base2.json 2> $OUT/base2.err
python - <<'P'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/ovr/*.json")):
    if "ops_" in f: continue
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f"{os.path.basename(f):34s} lanes {j['value']:7.2f}  single {j['single_stream']['value']:7.2f}  sum-of-kernels {j.get('forward_ms_sum_of_kernels')}")
    except Exception as e: print(f, "ERR", e)
P
