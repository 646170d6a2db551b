"""simdjson_amd -- MI355X-native stage-1 (structural indexing), minify and validate_utf8 for simdjson.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C-ABI of include/sjgpu.h, the
simdjson::implementation plug-in shim), the ctypes mirror of the C-ABI (capi), synthetic corpora and
the in-tree build.  The CPU checkers live under oracle/ and are never imported from this package.
"""
__version__ = "0.1.0"
