#!/bin/bash
# Collect the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r03 [commit]
# rocprofv3 runs follow MI355X_MICROARCH.md: kernel-trace/stats and each PMC set in their own runs, csv output.
set -o pipefail
R=${1:-r03}
COMMIT=${2:-unknown}
F="corr|warp|featnorm|census|photo|splat|smooth|down4|up4|level|pair"
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err || exit 1
python3 tools/kbench.py --iters 50 --filter "$F" > $O/kbench.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bench -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_line_rocprof.json 2> $O/stats_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_kbench -o k -- python3 tools/kbench.py --iters 20 --filter "$F" > $O/stats_kbench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 tools/kbench.py --iters 3 --filter "$F" --manifest $O/manifest_fetch.json > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 tools/kbench.py --iters 3 --filter "$F" --manifest $O/manifest_write.json > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_valu -o v -- python3 tools/kbench.py --iters 3 --filter "$F" --manifest $O/manifest_valu.json > $O/pmc_valu.log 2>&1 || exit 1
python3 tools/pmc_calls.py traffic $O/manifest_fetch.json $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $O/pmc_traffic.json $COMMIT || exit 1
python3 tools/pmc_calls.py valu $O/manifest_valu.json $O/pmc_valu/v_counter_collection.csv $O/pmc_valu.json $COMMIT || exit 1
python3 tools/level_bench.py > $O/level_bench.log 2>&1 || exit 1
python3 tools/kbench_cold.py > $O/kbench_cold.log 2>&1 || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value tools/ubench/div_exact.hip -o /tmp/div_exact 2>/dev/null && timeout -k 10 300 /tmp/div_exact 16384 > $O/div_exact.log 2>&1 || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/atomic_shape.hip -o /tmp/atomic_shape 2>/dev/null && timeout -k 10 120 /tmp/atomic_shape > $O/atomic_shape.log 2>&1 || exit 1
for w in "pwclite+unflow_loss 384 640 8" "pwclite_uflow+uflow_loss 448 1024 4" "pwcflow+uflow_loss 256 448 8" "pwclite3+mv_loss 384 640 8"; do
  set -- $w
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $1 --size $2 $3 --batch $4 2>/dev/null | tail -1 >> $O/bench_other_configs.jsonl || exit 1
done
python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2.err || exit 1
ARFLOW_FORCE_COLLECTIVES=1 ARFLOW_GLOBAL_LOSS_NORM=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/bench_rccl.err || exit 1
ls $O
