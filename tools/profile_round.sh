#!/bin/bash
# rocprofv3 passes behind profiles/: kernel-trace stats of the bench command, then one --pmc pass per counter group on the
# full-size network kernels (tools/prof_kernels.py).  Run on the GPU box: bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
REPO=$PWD
rm -rf "$OUT"; mkdir -p "$OUT/pmc"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --quick > "$OUT/bench.json" 2> "$OUT/bench.err"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM"; do
  d="$OUT/pmc/$(echo $grp | tr ' ' '_' | cut -c1-48)"
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$d" -o p -- python $REPO/tools/prof_kernels.py > "$d.log" 2>&1 || echo "pass $grp failed" >> "$OUT/pmc_errors.log"
done
python $REPO/tools/summarize_rocprof.py stats "$OUT/stats" "$OUT/kernel_stats.md" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu --quick"
python $REPO/tools/summarize_rocprof.py trace "$OUT/stats" "$OUT/kernel_trace_by_size.md" "rocprofv3 --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu --quick"
python $REPO/tools/summarize_rocprof.py pmc "$OUT/pmc" "$OUT/pmc.json"
# keep the merge small
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete
