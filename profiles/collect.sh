#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers.  Run on the GPU box from the repository root:
#   bash profiles/collect.sh r01
# Passes 1-2: kernel trace + stats of the headline bench command (1M-body direct sum) and of the Barnes-Hut
# workload.  Later passes: HBM traffic and SQ counters of both, each counter group in a run of its own (no
# tracing domains together with --pmc).  Outputs land in gpurun_out/prof_<tag>/ ; summarise.py condenses them into
# profiles/<tag>_*.txt and profiles/traffic.json, which are the files that get committed.
set -u
# A second argument picks a part (a GPU call here is limited to 20 minutes): traces | pmc | all (default).
TAG=${1:-r03}
PART=${2:-all}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { echo "== $*" >&2; timeout -s KILL "$@"; }
if [ "$PART" = all ] || [ "$PART" = traces ]; then
run 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_direct" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/bench_trace_direct.json" 2> "$OUT/bench_trace_direct.err" || exit 1
run 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_tree" -- python3 bench.py --workload tree --steps 20 --warmup 2 --no-cpu-baseline --no-extra > "$OUT/bench_trace_tree.json" 2> "$OUT/bench_trace_tree.err" || exit 1
run 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_dist" -- python3 profiles/dist_step.py > "$OUT/dist_step.log" 2> "$OUT/dist_step.err" || exit 1
run 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_walk16" -- python3 profiles/walk_16M.py > "$OUT/walk_16M.log" 2> "$OUT/walk_16M.err" || exit 1
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
run 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_direct" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc1.err" || exit 1
run 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_direct" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc2.err" || exit 1
run 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_tree" -- python3 bench.py --workload tree --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc3.err" || exit 1
run 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_tree" -- python3 bench.py --workload tree --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc4.err" || exit 1
run 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq_direct" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc5.err" || exit 1
run 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq_tree" -- python3 bench.py --workload tree --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc6.err" || exit 1
run 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_lds_tree" -- python3 bench.py --workload tree --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc7.err" || exit 1
run 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_grbm_direct" -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc8.err" || exit 1
run 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq_walk16" -- python3 profiles/walk_16M.py > /dev/null 2> "$OUT/pmc9.err" || exit 1
fi
find "$OUT" -name "*.csv" | head -40 >&2
python3 profiles/summarise.py "$TAG"
