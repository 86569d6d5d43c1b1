#!/bin/bash
# round 5: crop adjoint with pipelined entry loads against the previous build, same box, alternating
mkdir -p gpurun_out
{
for rep in 1 2; do
  echo "== base (rep $rep)"; timeout 300 python tools/exp/ab_lib.py tools/exp/libbase_sampler.so tools/sampler_bench.py 2>&1 | grep -v amdgpu.ids
  echo "== new (rep $rep)"; timeout 300 python tools/sampler_bench.py 2>&1 | grep -v amdgpu.ids
done
echo "== C4 size, base"; timeout 300 python tools/exp/ab_lib.py tools/exp/libbase_sampler.so tools/sampler_bench.py 95 2160 3840 2>&1 | grep -v amdgpu.ids
echo "== C4 size, new"; timeout 300 python tools/sampler_bench.py 95 2160 3840 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05s_sampler_ab.txt 2>&1
cat gpurun_out/r05s_sampler_ab.txt
