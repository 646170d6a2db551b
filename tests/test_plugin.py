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
    out = subprocess.run([built[name]], capture_output=True, text=True, timeout=1200, cwd=_paths.REPO_ROOT)  # the data directory is compiled in relative to the checkout
    tail = (out.stdout[-2500:] + out.stderr[-1500:])
    assert out.returncode == 0, tail
    assert "[active implementation: mi355x]" in out.stderr, tail


# ---- in-tree registration: the backend selected BY NAME, the way the reference selects its own kernels ----------
INTREE_SUITE = sorted(build.INTREE_TESTS)


def _intree(name):
    _binary()
    built = {os.path.basename(p): p for p in build.build_intree()}
    assert name in built, f"{name} was not built (needs the build container)"
    assert build.binary_is_current(name), f"build/tests/{name} is stale (sources changed since it was built)"
    return built[name]


def test_intree_registration_keeps_the_cpu_kernels_and_lists_mi355x():
    """The patched library (simdjson_mi355x.patch applied to a scratch copy, -DSIMDJSON_IMPLEMENTATION_MI355X=1) still passes
    the reference's basictests on its CPU kernel, and knows the name `mi355x`: without a GPU the programs run and every
    parser creation answers UNSUPPORTED_ARCHITECTURE -- no abort, no CPU stand-in."""
    import torch
    exe = _intree("intree_basictests")
    out = subprocess.run([exe, "-a", "haswell"], capture_output=True, text=True, timeout=600, cwd=_paths.REPO_ROOT)
    assert out.returncode == 0 and "Basic tests are ok." in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    if not torch.cuda.is_available():
        out = subprocess.run([exe, "-a", "mi355x"], capture_output=True, text=True, timeout=600, cwd=_paths.REPO_ROOT)
        assert "Unsupported architecture value" not in out.stderr
        assert "Running tests against this implementation: mi355x" in out.stdout
        assert out.returncode != 0 and "UNSUPPORTED_ARCHITECTURE" in (out.stdout + out.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("name", INTREE_SUITE)
@pytest.mark.parametrize("how", ["-a mi355x", "SIMDJSON_FORCE_IMPLEMENTATION"])
def test_intree_reference_tests_select_mi355x_by_name(name, how):
    """tests/dom/basictests.cpp and errortests.cpp of the reference, untouched, against the patched library:
    `-a mi355x` (tests/dom/basictests.cpp:2461-2470) and SIMDJSON_FORCE_IMPLEMENTATION=mi355x (src/implementation.cpp:296-312)."""
    exe = _intree(name)
    env = dict(os.environ)
    args = [exe]
    if how == "-a mi355x":
        args += ["-a", "mi355x"]
    else:
        env["SIMDJSON_FORCE_IMPLEMENTATION"] = "mi355x"
    out = subprocess.run(args, capture_output=True, text=True, timeout=900, env=env, cwd=_paths.REPO_ROOT)
    tail = out.stdout[-2500:] + out.stderr[-1500:]
    assert out.returncode == 0, tail
    if name == "intree_basictests":
        assert "Running tests against this implementation: mi355x" in out.stdout, tail
        assert "Basic tests are ok." in out.stdout, tail


@pytest.mark.gpu
def test_sanitizers_over_the_host_shim():
    """AddressSanitizer and ThreadSanitizer builds of libsjgpu's host code + the plug-in + plugin_test (scripts/sanitize.sh):
    dom::parser, ondemand, threaded parse_many (the stage-1 worker thread), minify / validate_utf8, all seven stage1 modes
    on the GPU, and not one report.  The binaries are built in the build container and travel."""
    exe = {s: os.path.join(_paths.REPO_ROOT, "build", "san", f"plugin_test_{s}") for s in ("address", "thread")}
    if not all(os.path.exists(p) for p in exe.values()):
        pytest.skip("build/san/* not built (bash scripts/sanitize.sh build, needs the reference)")
    base = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0",
                TSAN_OPTIONS="report_signal_unsafe=0:halt_on_error=0:suppressions=" + os.path.join(_paths.REPO_ROOT, "scripts", "tsan.supp"))
    routes = [{}, {"SJGPU_STREAM_FROM_MB": "1", "SJGPU_STREAM_CHUNK_MB": "1"},  # the overlapped path: copy threads, 1 MiB ranges
              {"SJGPU_DEVICES": "0,0", "SJGPU_MGPU_FROM_MB": "1"}]              # two shard threads per document (the one device twice)
    for san, path in exe.items():
        for route in routes:
            out = subprocess.run([path, "--jsonexamples", EXAMPLES], capture_output=True, text=True, timeout=900, env=dict(base, **route))
            text = out.stdout + out.stderr
            assert "plugin test OK" in out.stdout, (san, route, text[-3000:])
            assert "ERROR: AddressSanitizer" not in text and "WARNING: ThreadSanitizer" not in text, (san, route, text[-4000:])
            # (one exit code is forgiven: ROCm's AddressSanitizer runtime can die in a CHECK of its OWN device allocator -- sanitizer_allocator_device.h,
            # "dev_runtime_unloaded_" -- when libamdhip64's exit handlers free memory after that allocator has been unloaded: after main has returned, after the
            # verdict above, no report about this repository's code; seen in round 5 on the overlapped route, gpurun_out/r5b_asan.log)
            teardown = san == "address" and "sanitizer_allocator_device.h" in text and "__cxa_finalize" in text
            assert out.returncode == 0 or teardown, (san, route, text[-2000:])
