"""tc_gan_amd -- MI355X-native hot path of ahmadianlab/tc-gan.

Host code stays in Python and mirrors the reference's operator surface for the
SSN fixed-point / BPTT-cWGAN path; all arithmetic runs in hand-written HIP
kernels (gfx950) reached through the ctypes C ABI in ``tc_gan_amd.clib``
(``include/ssnode_mi355x.h``).  There is no CPU fallback: importing
``tc_gan_amd.clib`` raises ``OSError`` when ``ext/libssnode.so`` is missing, and
every compute call raises ``GPUUnavailableError`` without a HIP device.
"""

__version__ = '0.1.0'
