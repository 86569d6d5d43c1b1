from aphantasia_amd.transforms import *  # noqa: F401,F403
from aphantasia_amd.transforms import normalize, transforms_fast  # noqa: F401
