#!/bin/bash
# usage: tools/prof.sh <tag> <workload> [bench args...]  — kernel trace + PMC passes around bench.py, condensed into
# gpurun_out/prof_<tag>.txt, gpurun_out/kernel_stats_<tag>.csv, gpurun_out/traffic_<tag>.json (bench line under the
# profiler: gpurun_out/prof_<tag>_bench.json).  Each counter group is its own run (TCC slot limits, MI355X_MICROARCH.md).
# The un-profiled bench line (with its cpu_baseline) is taken first, in the same session on the same board:
# gpurun_out/prof_<tag>_bench_unprofiled.json.  Kernel statistics leave the warm-up dispatches out (tools/kernel_times.py).
TAG=$1; WL=$2; shift 2
P=/tmp/prof_$TAG; mkdir -p $P gpurun_out
STEPS=3; WARM=1; KT_WARM=3
python bench.py --workload $WL "$@" > gpurun_out/prof_${TAG}_bench_unprofiled.json 2> $P/bench.err || { tail -5 $P/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- python bench.py --workload $WL --steps 20 --warmup $KT_WARM --no-cpu "$@" > gpurun_out/prof_${TAG}_bench.json 2> $P/kt.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $P/pmc1 -- python bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu "$@" > /dev/null 2> $P/pmc1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $P/pmc2 -- python bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu "$@" > /dev/null 2> $P/pmc2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d $P/pmc3 -- python bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu "$@" > /dev/null 2> $P/pmc3.err
python tools/prof_summary.py $P/kt $P/pmc1 $P/pmc2 $P/pmc3 | grep -A10 "kernel_stats\|kernel: void pdog::dog\|kernel: pdog::dog" | grep -v "at::native" | cut -c1-200 > gpurun_out/prof_${TAG}.txt
python tools/kernel_times.py $P/kt $KT_WARM | grep "^Name\|pdog::" > gpurun_out/kernel_stats_${TAG}.csv
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc4 -- python bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu "$@" > /dev/null 2> $P/pmc4.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc5 -- python bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu "$@" > /dev/null 2> $P/pmc5.err
python tools/prof_summary.py $P/pmc4 $P/pmc5 | grep -A3 "kernel: void pdog::dog\|kernel: pdog::dog" | cut -c1-200 >> gpurun_out/prof_${TAG}.txt
VAR=$(python -c "import json;print(json.load(open('gpurun_out/prof_${TAG}_bench.json'))['config']['kernel_for_this_batch'])")
BATCH=$(python -c "import json;print(json.load(open('gpurun_out/prof_${TAG}_bench.json'))['config']['batch_per_gpu'])")
python tools/traffic_from_pmc.py $WL $(python -c "import json;print(json.load(open('gpurun_out/prof_${TAG}_bench.json'))['config']['variant'])") $BATCH $((STEPS+WARM)) $P/pmc4 $P/pmc5 > gpurun_out/traffic_${TAG}.json
