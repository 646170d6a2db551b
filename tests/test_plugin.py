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
