#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/profile_round.sh <tag>     e.g. r02a
# One kernel-trace pass (durations, --stats) and five PMC passes (counters only with --kernel-trace, as gpurun requires) over the
# SAME command: the timed bf16 batch-8 bench without its CPU / fp32 legs.  Raw output under gpurun_out/prof_<tag>/; the summaries
# that get committed are made afterwards by profiles/pmc_kernels.py and profiles/summarize_rocpd.py.
set -u
TAG="${1:-r02}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
# --lanes 1: the profiled passes run the steps on ONE stream, so that a kernel's duration is its own (two lanes co-schedule kernels of two
# batches) and the launch order maps onto the plan's ops (profiles/pmc_kernels.py)
CMD="python bench.py --steps 3 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32 --no-profile --no-sustained --no-seam2 --no-configs"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-sustained --no-seam2 --no-configs --dump-ops "$OUT/ops.json" > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
rocprofv3 --kernel-trace --stats -d "$OUT" -o trace -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT" -o write -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o sq1 -- $CMD > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o sq2 -- $CMD > "$OUT/sq2.log" 2>&1
# round 3: the train step (BASELINE configs[4]) and the fp16x3 parity mode, kernel durations only
rocprofv3 --kernel-trace --stats -d "$OUT" -o train -- python bench.py --train --steps 3 --warmup 1 --no-profile --no-cpu-baseline > "$OUT/train.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT" -o x3 -- python bench.py --precision fp16x3 --steps 3 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32 --no-profile --no-sustained --no-seam2 --no-configs > "$OUT/x3.log" 2>&1
# round 6: the PMC passes for the contract-grade mode too (fp16x3: north_star_value.roofline.traffic), in their own directory -- pmc_kernels.py maps
# dispatches to the plan of ONE precision
X3="$OUT/x3pmc"
mkdir -p "$X3"
CMDX3="python bench.py --precision fp16x3 --steps 3 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32 --no-profile --no-sustained --no-seam2 --no-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$X3" -o fetch -- $CMDX3 > "$X3/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$X3" -o write -- $CMDX3 > "$X3/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$X3" -o sq1 -- $CMDX3 > "$X3/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$X3" -o sq2 -- $CMDX3 > "$X3/sq2.log" 2>&1
ls -la "$OUT" | head -40
# summaries are made ON the box (gpurun copies back at most 64 MiB of gpurun_out/): the raw counter CSVs stay behind
mkdir -p "$ROOT/gpurun_out/summ_$TAG"
mv "$X3" "$ROOT/gpurun_out/x3pmc_$TAG"                        # (out of $OUT: pmc_kernels.py globs one level of sub-directories)
python profiles/pmc_kernels.py "$OUT" --tag "$TAG" > "$ROOT/gpurun_out/summ_$TAG/pmc_kernels.log" 2>&1
python profiles/pmc_kernels.py "$ROOT/gpurun_out/x3pmc_$TAG" --tag "$TAG" --precision fp16x3 > "$ROOT/gpurun_out/summ_$TAG/pmc_kernels_fp16x3.log" 2>&1
rm -rf "$ROOT/gpurun_out/x3pmc_$TAG"
python profiles/summarize_rocpd.py "$OUT/trace_results.db" "profiles/${TAG}_bf16_b8_kernel_stats.txt"
python profiles/summarize_rocpd.py "$OUT/train_results.db" "profiles/${TAG}_train_bf16_b8_kernel_stats.txt"
python profiles/summarize_rocpd.py "$OUT/x3_results.db" "profiles/${TAG}_fp16x3_b8_kernel_stats.txt"
cp profiles/${TAG}_* "$ROOT/gpurun_out/summ_$TAG/"
cp "$OUT/bench_unprofiled.json" "$ROOT/gpurun_out/summ_$TAG/${TAG}_bf16_b8_bench.json"
cp "$OUT/ops.json" "$ROOT/gpurun_out/summ_$TAG/ops.json"
grep '^{"metric"' "$OUT/train.log" | tail -1 > "$ROOT/gpurun_out/summ_$TAG/train_bench_line.txt"
grep '^{"metric"' "$OUT/x3.log" | tail -1 > "$ROOT/gpurun_out/summ_$TAG/x3_bench_line.txt"
rm -rf "$OUT"
