from aphantasia_amd.utils import *  # noqa: F401,F403
from aphantasia_amd.utils import (slice_imgs, sim_func, pad_up_to, tile_pad, basename, img_list, img_read, file_list,  # noqa: F401
                                  txt_clean, checkout, old_torch)
