#!/bin/bash
# round 5: split-precision QKV with half-length k-loops on the V column tiles (one launch): parity subset + C2 bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity_configs.py tests/test_gpu_kernels.py -m gpu -x -q -s -k "stress or hilo or split or precise or wave_specialised or ws or headline or c2" 2>&1 | grep -v "amdgpu.ids" | grep "stress\|PRECISE\|passed\|failed\|Error\|error\|split" | tail -20
for i in 1 2; do timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['steps_per_s'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"; done
timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-legs --f16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16', d['value'], d['roofline']['frac'])"
