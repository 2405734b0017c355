# K assembly, where the time goes (round 4): the kernel as it is; its arithmetic alone; its stores alone.  The two diagnostic
# builds are NOT in the tree -- they were one-line switches in kmatrix_body (arithmetic loop skipped / stores skipped unless a
# value that cannot occur turns up).  Measured at n = 16 384, D = 10, full symmetric, rotating outputs (profiles/r04_kmatrix_hbm_roofline.txt):
#   SExp   both 490-530 us   arithmetic alone 283 us   stores alone 400 us   (a write-only fill_ of the same bytes: 318 us)
#   Matern both 545-581 us   arithmetic alone 399 us   stores alone 400 us
# i.e. the phases overlap only partly, and the tile-shaped stores reach 0.67 of 8 TB/s where fill_ reaches 0.85.
timeout 300 python tools/gpu_kmatrix_roofline.py
