"""The drop-in boundary: simdjson's own front-ends (dom::parser, ondemand::parser, parse_many, minify,
validate_utf8) running on the UNMODIFIED reference library with the mi355x implementation activated
(simdjson_amd/csrc/plugin/, tests/plugin/plugin_test.cpp).  The binaries are built in the build container
(they need the reference's headers and library object) into build/tests/ and travel with the repo; a binary
that was built from other sources than the ones in the tree is refused (build.binary_is_current)."""
import os
import subprocess

import pytest

from simdjson_amd import _paths, build

EXAMPLES = os.path.join(_paths.REPO_ROOT, "tests", "golden", "jsonexamples")


def _binary():
    build.build_corpus()
    build.build_oracle()
    build.build_sjgpu()
    build.build_plugin()
    exe = build.build_plugin_test()
    if exe is not None:
        assert build.binary_is_current("plugin_test"), "build/tests/plugin_test is stale (sources changed since it was built)"
    return exe


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
    out = subprocess.run([exe, "--jsonexamples", EXAMPLES], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "plugin test OK" in out.stdout
    assert "configs[0] twitter.json: n = 55263, fnv = 9964107509431273939" in out.stdout, out.stdout[-3000:]


REFERENCE_SUITE = sorted(build.REFERENCE_TESTS)


@pytest.mark.gpu
@pytest.mark.parametrize("name", REFERENCE_SUITE)
def test_reference_own_test_programs_on_mi355x(name):
    """The reference's OWN test programs (compiled in place from /root/reference/tests, untouched) run with the
    mi355x backend activated before main(): document_stream (dom + ondemand, threaded stage-1 worker, RFC 7464
    and comma-delimited matrices, truncation constants), the seeded stream fuzz corpus, and unicode_tests."""
    _binary()
    built = {os.path.basename(p): p for p in build.build_reference_tests()}
    assert name in built, f"{name} was not built (needs the build container)"
    assert build.binary_is_current(name), f"build/tests/{name} is stale (sources changed since it was built)"
    out = subprocess.run([built[name]], capture_output=True, text=True, timeout=1200)
    tail = (out.stdout[-2500:] + out.stderr[-1500:])
    assert out.returncode == 0, tail
    assert "[active implementation: mi355x]" in out.stderr, tail
