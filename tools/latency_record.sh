#!/bin/bash
# usage: tools/latency_record.sh > gpurun_out/<round>_latency.txt  — the single-clip paths on one board, one session:
# host functor and device chain (tools/latency.py), chains over other geometries (tools/chain_latency.py), hard frames
# (every window flagged: tools/hard_chain.py), the cfg1 clip through bench.py --chain, and the register / spill / scratch
# figures of the latency kernels (tools/kernel_regs.sh).
python tools/latency.py 2>&1 | grep -v amdgpu.ids
for g in "1080 1920 25 128" "1080 1920 25 256" "2160 3840 25 512" "1080 1920 50 256" "1080 1920 12 45"; do python tools/chain_latency.py $g 2>&1 | tail -1; done
for a in "45 noise" "45 faint" "256 noise" "256 faint"; do python tools/hard_chain.py $a 2>&1 | grep -v amdgpu.ids; done
python bench.py --workload cfg1 --chain --steps 20 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench --workload cfg1 --chain: %.2f us per frame, %.0f frames/s; CPU oracle on the same chain: %.0f frames/s (%s threads)' % (r['us_per_frame'], r['value'], r['cpu_baseline']['value'], r['cpu_baseline']['cores']))"
tools/kernel_regs.sh pawsometracker.jl_amd/csrc/_obj/pawsome_dog.o "dog_fused_kernel|dog_tiled_kernel"; tools/kernel_regs.sh pawsometracker.jl_amd/csrc/_obj/lat_inst_65.o "."
