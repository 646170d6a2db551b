"""Filesystem layout shared by the build, the ctypes bindings, the tests and bench.py."""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")            # in-tree build outputs (git-ignored *.so, travel via gpurun)
INCLUDE_DIR = os.path.join(REPO_ROOT, "include")
ORACLE_DIR = os.path.join(REPO_ROOT, "oracle")
ORACLE_OUT = os.path.join(ORACLE_DIR, "_ref")
REFERENCE_DIR = os.environ.get("SIMDJSON_REFERENCE", "/root/reference")

LIB_SJGPU = os.environ.get("SJGPU_LIB") or os.path.join(LIB_DIR, "libsjgpu.so")  # HIP kernels + C-ABI (the product)
LIB_CORPUS = os.path.join(LIB_DIR, "libsjcorpus.so")      # synthetic corpora (host tooling)
LIB_PLUGIN = os.path.join(LIB_DIR, "libsimdjson_mi355x.so")  # simdjson::implementation shim (needs reference headers to build)
LIB_ORACLE = os.path.join(ORACLE_OUT, "libsjoracle.so")   # test infrastructure
LIB_REF = os.path.join(ORACLE_OUT, "libsjref.so")         # test infrastructure (real reference)
