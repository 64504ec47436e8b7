#!/usr/bin/env bash
# Collect the measurement artifacts of one round on the GPU box (run through gpurun from the repository root):
#   bash profiles/collect_round.sh r02
# Writes raw rocprofv3 output under gpurun_out/<round>/raw (scratch) and the summaries that get committed under
# gpurun_out/<round>/out; copy the latter to profiles/<round>/ afterwards.  PMC passes are separate runs with
# --kernel-trace only (FETCH_SIZE and WRITE_SIZE cannot share a pass).
set -o pipefail
R=${1:-r04}
OUT=gpurun_out/$R/out
RAW=gpurun_out/$R/raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEP="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica --no-extra-modes"
# PART=a: items 1-4, PART=b: items 5-7 (a gpurun call is limited to 20 minutes); default both
PART=${PART:-ab}
if [[ $PART == *a* ]]; then
# 1. the default bench line under the profiler: per-kernel statistics of the whole command
rocprofv3 --kernel-trace --stats -d $RAW/default --output-format csv -- python3 bench.py > $OUT/bench_default_output.json 2> $RAW/default.err
cp $(ls $RAW/default/*/*kernel_stats.csv | head -1) $OUT/bench_default_kernel_stats.csv
python3 profiles/summarize.py $RAW/default 1 -1 > $OUT/bench_default_kernel_stats.txt
# 2. un-profiled bench line
python3 bench.py > $OUT/bench_unprofiled_output.json 2> $RAW/unprofiled.err
# 3. one ADMM iteration, kernel by kernel, under the profiler (the shipped schedule; round 4 has the same timeline WITHOUT a
#    profiler: step_timeline_unprofiled.txt / step_completions_unprofiled.txt below)
rocprofv3 --kernel-trace -d $RAW/step --output-format csv -- python3 bench.py $STEP > $RAW/step.json 2> $RAW/step.err
python3 profiles/step_trace.py $RAW/step > $OUT/step_timeline.txt
# 3b. the same iteration through lshm_trace_* (no profiler): start + stop events per launch, then stop events only
python3 profiles/step_trace_unprofiled.py > $OUT/step_timeline_unprofiled.txt 2> $RAW/trace.err
TRACE_ENDS_ONLY=1 python3 profiles/step_trace_unprofiled.py > $OUT/step_completions_unprofiled.txt 2>> $RAW/trace.err
python3 profiles/phase_times_probe.py > $OUT/phase_times.txt 2> $RAW/phase.err
python3 profiles/deep2d_probe.py > $OUT/deep2d_probe.txt 2> $RAW/deep2d.err
python3 profiles/chain1d_full_probe.py > $OUT/chain1d_full_probe.txt 2> $RAW/chain1d_full.err
python3 profiles/resid_conv0_probe.py > $OUT/resid_conv0_probe.txt 2> $RAW/resid_conv0.err
python3 profiles/conv0_bwd_tile_probe.py > $OUT/conv0_bwd_tile_probe.txt 2> $RAW/conv0_bwd_tile.err
# 4. HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes, keyed by the profiled command
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $RAW/pmc_step_$C --output-format csv -- python3 bench.py $STEP > /dev/null 2> $RAW/pmc_step_$C.err
  rocprofv3 --pmc $C --kernel-trace -d $RAW/pmc_khm10_$C --output-format csv -- python3 bench.py --only-khm --K 10 > /dev/null 2> $RAW/pmc_khm10_$C.err
  rocprofv3 --pmc $C --kernel-trace -d $RAW/pmc_khm64_$C --output-format csv -- python3 bench.py --only-khm --K 64 > /dev/null 2> $RAW/pmc_khm64_$C.err
  rocprofv3 --pmc $C --kernel-trace -d $RAW/pmc_bf16_$C --output-format csv -- python3 bench.py $STEP --bf16 > /dev/null 2> $RAW/pmc_bf16_$C.err
done
python3 profiles/pmc_traffic.py $OUT/hbm_traffic.json step:$RAW/pmc_step_FETCH_SIZE:$RAW/pmc_step_WRITE_SIZE \
  khm_N1048576_K10:$RAW/pmc_khm10_FETCH_SIZE:$RAW/pmc_khm10_WRITE_SIZE khm_N1048576_K64:$RAW/pmc_khm64_FETCH_SIZE:$RAW/pmc_khm64_WRITE_SIZE \
  step_bf16:$RAW/pmc_bf16_FETCH_SIZE:$RAW/pmc_bf16_WRITE_SIZE
fi
if [[ $PART == *b* ]]; then
# 5. matrix-pipe utilisation per kernel of the step
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace -d $RAW/pmc_mfma --output-format csv -- python3 bench.py $STEP > /dev/null 2> $RAW/pmc_mfma.err
rocprofv3 --kernel-trace --stats -d $RAW/stepstats --output-format csv -- python3 bench.py $STEP > /dev/null 2> $RAW/stepstats.err
python3 profiles/mfma_util.py $RAW/pmc_mfma $(ls $RAW/stepstats/*/*kernel_stats.csv | head -1) > $OUT/mfma_utilisation.txt
# 5b. the same for BASELINE configs[2] (bf16 operands + storage)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace -d $RAW/pmc_mfma_bf16 --output-format csv -- python3 bench.py $STEP --bf16 > /dev/null 2> $RAW/pmc_mfma_bf16.err
rocprofv3 --kernel-trace --stats -d $RAW/stepstats_bf16 --output-format csv -- python3 bench.py $STEP --bf16 > /dev/null 2> $RAW/stepstats_bf16.err
python3 profiles/mfma_util.py $RAW/pmc_mfma_bf16 $(ls $RAW/stepstats_bf16/*/*kernel_stats.csv | head -1) > $OUT/bf16_mfma_utilisation.txt
# 6. BASELINE configs[2]: bf16 bench line + kernel statistics + timeline
rocprofv3 --kernel-trace --stats -d $RAW/bf16 --output-format csv -- python3 bench.py --bf16 --no-cpu-baseline > $OUT/bf16_bench_output.json 2> $RAW/bf16.err
cp $(ls $RAW/bf16/*/*kernel_stats.csv | head -1) $OUT/bf16_kernel_stats.csv
python3 profiles/summarize.py $RAW/bf16 1 -1 > $OUT/bf16_kernel_stats.txt
rocprofv3 --kernel-trace -d $RAW/bf16step --output-format csv -- python3 bench.py $STEP --bf16 > /dev/null 2> $RAW/bf16step.err
python3 profiles/step_trace.py $RAW/bf16step > $OUT/bf16_step_timeline.txt
# 7. K = 64 (config 5's cluster count): bench line and timeline
python3 bench.py --K 64 --no-cpu-baseline --no-rica > $OUT/k64_bench_output.json 2> $RAW/k64.err
rocprofv3 --kernel-trace -d $RAW/k64step --output-format csv -- python3 bench.py $STEP --K 64 > /dev/null 2> $RAW/k64step.err
python3 profiles/step_trace.py $RAW/k64step > $OUT/step_timeline_k64.txt
fi
rm -rf $RAW
ls -la $OUT
