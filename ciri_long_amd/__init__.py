"""ciri_long_amd (repository directory also reachable as ciri-long_amd/): MI355X (gfx950) implementation of CIRI-long's per-read hot path
(find_ccs consensus -> find_bsj clip re-alignment by Smith-Waterman), behind CIRI-long's own Python interfaces.

Modules mirror the reference tree:
    ssw_wrap   <- libs/striped_smith_waterman/ssw_wrap.py   (Aligner / PyAlignRes over libclh.so)
    hip        -- ctypes binding of the batched C ABI (include/ciri_long_hip.h)

The arithmetic runs in hand-written HIP kernels (csrc/); there is no CPU fallback: importing works anywhere, the
first compute call raises if libclh.so or a GPU is missing.
"""
__version__ = '0.1.0'
