#!/bin/bash
# round 5: crop adjoint with pipelined entry loads against the previous build, same box, alternating
# (record of a rejected experiment: tools/exp/libbase_sampler.so was the library built from the committed sampler.hip, the in-tree library the
# variant -- neither the variant source nor that .so is kept; profiles/r05_sampler_pipelined_ab.txt has the numbers and sampler.hip a comment)
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
