"""CPU-side checks of the product boundary: the C-ABI library loads, exports every symbol
include/rt_mi355x.h declares, and fails loudly (no fallback) when no HIP device exists.
No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", txt)) - {"rt_callback_fn"})


def test_library_loads_and_exports_every_declared_symbol():
    from raytracertest_amd import api
    lib = api.load_library()
    decl = declared_symbols()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), "missing export " + name
    assert sorted(api.ABI_SYMBOLS) == decl, "api.ABI_SYMBOLS out of sync with the header"
    assert b"gfx950" in lib.rt_version()


def test_gfx950_code_object_is_embedded():
    from raytracertest_amd import api
    blob = open(api.library_path(), "rb").read()
    assert b"gfx950" in blob and b"trace_kernel" in blob


def test_no_cpu_fallback_without_device():
    import raytracertest_amd as R
    if R.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(R.RtError, match="no HIP device|no CPU fallback"):
        R.RayTracer((8, 8), seed=1)


def test_options_struct_layout_matches_header():
    from raytracertest_amd import api
    assert ctypes.sizeof(api.Options) == 56                  # (48 up to round 2: `transport` was appended; struct_size versions it)
    assert api.Options.seed.offset == 24 and api.Options.flags.offset == 32 and api.Options.transport.offset == 48


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under raytracertest_amd/ or include/ may
    import, include or link it."""
    bad = []
    for base in ("raytracertest_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"oracle", txt, re.I):
                        bad.append(os.path.join(dirpath, f))
    assert bad == []


def test_header_is_plain_c99(tmp_path):
    """The boundary must be bindable from C: include/rt_mi355x.h compiles as pedantic C99."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "rt_mi355x.h"\nint main(void) { rt_options o; o.struct_size = sizeof o; '
                   'return rt_device_count() < 0 ? (int)o.struct_size : 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                    "-c", str(src), "-o", str(tmp_path / "abi.o")], check=True)
