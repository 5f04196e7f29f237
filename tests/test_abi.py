"""The C-ABI library loads on a machine without a GPU, exports every symbol include/nbody.h declares,
and fails loudly (no CPU fallback) when asked to compute.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "nbody.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*\s*|int\s+|void\s+)(\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_and_binding_agree(nb):
    assert declared_symbols() == sorted(nb._lib.SYMBOLS)


def test_library_exports_every_declared_symbol(nb):
    lib = ctypes.CDLL(nb._lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_no_torch_types_and_no_oracle_in_product(nb):
    """The boundary is plain C; the product never touches oracle/."""
    hdr = open(os.path.join(ROOT, "include", "nbody.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)     # comments may mention torch; declarations may not
    assert "torch" not in code and "hip" not in code.lower() and "#include <stddef.h>" in code
    assert re.findall(r"#include\s*[<\"]([^>\"]+)", code) == ["stddef.h"]
    pkg = os.path.join(ROOT, "mini-nbody_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".c", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "nbody_ref" not in text.replace("oracle/nbody_ref.c", ""), f


def test_fails_loudly_without_gpu(nb):
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(nb.NBodyError) as e:
        nb.NBody(1024)
    assert e.value.code == nb._lib.ERR_NO_DEVICE
    lib = nb._lib.load()
    assert lib.nbody_step(0.01, 1) == nb._lib.ERR_NOT_INIT
    assert lib.bodyForce(None, None, 0.01, 4) == nb._lib.ERR_NOT_INIT
    assert b"no CPU path" in lib.nbody_error_string(nb._lib.ERR_NO_DEVICE)


def test_option_validation_needs_no_gpu(nb):
    lib = nb._lib.load()
    assert lib.nbody_set_option(nb.OPT_IBLOCK, 3) == nb._lib.ERR_ARG
    assert lib.nbody_set_option(nb.OPT_VARIANT, 9) == nb._lib.ERR_ARG
    assert lib.nbody_set_option(999, 0) == nb._lib.ERR_ARG
    assert lib.nbody_set_option(nb.OPT_IBLOCK, 0) == 0
    assert lib.nbody_set_option(nb.OPT_WSPLIT, 2) == nb._lib.ERR_ARG
    assert lib.nbody_set_option(nb.OPT_WSPLIT, -1) == 0
    # loop forms of the diagnostic build (experiment encodings, timing-only forms with wrong results) are refused
    for phase in list(range(2, 19)):
        assert lib.nbody_set_option(nb.OPT_ISA_PHASE, phase) == nb._lib.ERR_UNSUPPORTED, phase
    assert lib.nbody_set_option(nb.OPT_ISA_PHASE, 19) == nb._lib.ERR_ARG
    assert lib.nbody_set_option(nb.OPT_ISA_PHASE, 1) == 0
