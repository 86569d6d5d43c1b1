# drop-in alias of the reference's `depth` package (illustrip.py:30 `from depth import depth`)
