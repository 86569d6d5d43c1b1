#!/bin/bash
# round 5: randomised parity sweeps over the final build (kernels + whole steps, both precision modes)
mkdir -p gpurun_out
( timeout 900 python tools/gpu_fuzz.py 11; timeout 900 python tools/gpu_fuzz.py 12 ) 2>&1 | grep -v "amdgpu.ids" | tail -30 > gpurun_out/r05r_gpu_fuzz.txt
( timeout 1200 python tools/gpu_engine_fuzz.py 21 14; timeout 900 python tools/gpu_engine_fuzz.py 22 10 ) 2>&1 | grep -v "amdgpu.ids" | tail -40 > gpurun_out/r05r_engine_fuzz.txt
tail -8 gpurun_out/r05r_gpu_fuzz.txt; tail -30 gpurun_out/r05r_engine_fuzz.txt
