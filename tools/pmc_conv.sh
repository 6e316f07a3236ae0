#!/bin/bash
# usage: tools/pmc_conv.sh "<layer substring>" [mode]   -- two PMC passes over one conv_bench layer
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L="$1"; MODE="${2:-bf16}"; AUX="${3:-0}"; TAG=$(echo "$L" | tr ' >-' '___')
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc -o ${TAG}_p1 -- python tools/conv_bench.py --mode $MODE --reps 3 --nbuf $AUX --only "$L" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pmc -o ${TAG}_p2 -- python tools/conv_bench.py --mode $MODE --reps 3 --nbuf $AUX --only "$L" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_LEVEL_WAVES --output-format csv -d gpurun_out/pmc -o ${TAG}_p3 -- python tools/conv_bench.py --mode $MODE --reps 3 --nbuf $AUX --only "$L" > /dev/null 2>&1
python - <<PY
import csv,collections,glob
agg=collections.OrderedDict()
dur=[]
for f in sorted(glob.glob('gpurun_out/pmc/${TAG}_p*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'conv' in r['Kernel_Name']:
            agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for f in sorted(glob.glob('gpurun_out/pmc/${TAG}_p1_kernel_trace.csv')):
    for r in csv.DictReader(open(f)):
        if 'conv' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print("$L", "$MODE", "dur_us", sum(dur)/max(1,len(dur)))
for k,v in agg.items(): print(f"  {k:28s} {sum(v)/len(v):.4g}")
PY
