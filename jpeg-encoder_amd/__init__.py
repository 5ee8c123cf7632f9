"""Python view of libjpegenc_amd.so (the C-ABI in include/jpegenc_amd.h).

The product is the shared library and the C++ host classes; this package only
binds the C-ABI with ctypes so that tests and the benchmark can drive it and use
torch for device memory.  There is no fallback: if the library has not been
built (make -C jpeg-encoder_amd/csrc) importing `binding` raises.
"""
from .binding import Context, JeaError, LIB_PATH, load_library  # noqa: F401
