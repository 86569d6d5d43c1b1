#!/bin/bash
# round 5: one-rank communicator self-check test + library calibration of the GEMM
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_comm.py -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > gpurun_out/r05q_comm_test.txt
timeout 600 python tools/exp/gemm_vs_library.py > gpurun_out/r05q_gemm_vs_library.txt 2>&1
cat gpurun_out/r05q_comm_test.txt gpurun_out/r05q_gemm_vs_library.txt
