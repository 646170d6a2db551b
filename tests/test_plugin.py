"""The drop-in boundary: simdjson's own front-ends (dom::parser, ondemand::parser, parse_many, minify,
validate_utf8) running on the UNMODIFIED reference library with the mi355x implementation activated
(simdjson_amd/csrc/plugin/, tests/plugin/plugin_test.cpp).  The binary is built in the build container
(it needs the reference's headers and library object) and travels with the repo."""
import os
import subprocess

import pytest

from simdjson_amd import _paths, build

BIN = os.path.join(_paths.LIB_DIR, "plugin_test")


def _binary():
    build.build_corpus()
    build.build_oracle()
    build.build_sjgpu()
    build.build_plugin()
    return build.build_plugin_test()


def test_plugin_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = _binary()
    if exe is None:
        pytest.skip("reference headers absent and no prebuilt plugin_test")
    out = subprocess.run([exe, "--expect-no-gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_plugin_dropin_on_gpu():
    exe = _binary()
    assert exe is not None and os.path.exists(exe), "plugin_test was not built (needs the build container)"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "plugin test OK" in out.stdout


REFERENCE_SUITE = sorted(build.REFERENCE_TESTS)
# ~5.5 minutes on an MI355X box: tens of thousands of tiny stage-1 calls, each a PCIe round trip.  It passes
# (round 1, twice); run it with SJGPU_SLOW_TESTS=1.  The other four take ~45 s together.
SLOW = {"ref_dom_document_stream_tests"}


@pytest.mark.gpu
@pytest.mark.parametrize("name", REFERENCE_SUITE)
def test_reference_own_test_programs_on_mi355x(name):
    """The reference's OWN test programs (compiled in place from /root/reference/tests, untouched) run with the
    mi355x backend activated before main(): document_stream (dom + ondemand, threaded stage-1 worker, RFC 7464
    and comma-delimited matrices, truncation constants), the seeded stream fuzz corpus, and unicode_tests."""
    if name in SLOW and os.environ.get("SJGPU_SLOW_TESTS", "0") != "1":
        pytest.skip("slow (set SJGPU_SLOW_TESTS=1)")
    _binary()
    built = {os.path.basename(p): p for p in build.build_reference_tests()}
    assert name in built, f"{name} was not built (needs the build container)"
    out = subprocess.run([built[name]], capture_output=True, text=True, timeout=1200)
    tail = (out.stdout[-2500:] + out.stderr[-1500:])
    assert out.returncode == 0, tail
    assert "[active implementation: mi355x]" in out.stderr, tail
