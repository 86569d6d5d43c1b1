#!/bin/bash
# bench.py's N > 1 code path end to end on a 1-GPU box: two ranks on cuda:0, reduction through gloo (RCCL refuses two ranks on one
# device), then the single-rank line for comparison.  The direct-RCCL communicator itself is covered at world size 1 by tests/test_gpu_comm.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
APH_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/mr_gloo2.json 2> gpurun_out/mr_gloo2.err
echo "rc $?"; cut -c1-400 gpurun_out/mr_gloo2.json; tail -3 gpurun_out/mr_gloo2.err
