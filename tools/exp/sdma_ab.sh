cd $GRAFT_REPO_ROOT
echo "env: HSA_ENABLE_SDMA=$HSA_ENABLE_SDMA HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY"; env | grep -i "sdma\|^HSA\|^HIP\|^ROC\|^GPU_" | head
for v in 1 0; do echo "HSA_ENABLE_SDMA=$v"; HSA_ENABLE_SDMA=$v python tools/exp/save_ab.py 400 x 2>&1 | tail -1; done
