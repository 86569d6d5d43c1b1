"""Drop-in alias: `from aphantasia.image import fft_image, to_valid_rgb, dwt_image` etc. resolve to the
MI355X implementation in aphantasia_amd (same public names as the reference's `aphantasia` package for the
hot-path modules image / utils / transforms)."""
