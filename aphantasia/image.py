from aphantasia_amd.image import *  # noqa: F401,F403
from aphantasia_amd.image import to_valid_rgb, fft_image, dwt_image, pixel_image, rfft2d_freqs, resume_fft  # noqa: F401
