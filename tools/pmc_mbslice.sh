#!/bin/bash
# usage (GPU box): tools/pmc_mbslice.sh [B]   -- three PMC passes over tools/mbslice_bench.py (FTC_OP_MBHEAD alone), summary on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="${1:-8}"
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc -o mbs_p1 -- python tools/mbslice_bench.py $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pmc -o mbs_p2 -- python tools/mbslice_bench.py $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_LEVEL_WAVES --output-format csv -d gpurun_out/pmc -o mbs_p3 -- python tools/mbslice_bench.py $B > /dev/null 2>&1
python - <<PY
import csv,collections,glob
agg=collections.OrderedDict()
dur=[]
for f in sorted(glob.glob('gpurun_out/pmc/**/mbs_p*_counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'mbconv_slice' in r['Kernel_Name']:
            agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for f in sorted(glob.glob('gpurun_out/pmc/**/mbs_p1_kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'mbconv_slice' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print("mbconv_slice B=$B dur_us", sum(dur)/max(1,len(dur)), "launches", len(dur))
for k,v in agg.items(): print(f"  {k:28s} {sum(v)/len(v):.5g}")
PY
