# drop-in alias: the reference's depth/depth.py surface on the HIP kernels
from aphantasia_amd.depthwarp import InferDepthAny, depthwarp, grid_warp, resize  # noqa: F401
