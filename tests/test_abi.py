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
    pkg = os.path.join(ROOT, "mini_nbody_amd")
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
    for phase in list(range(2, 21)):
        assert lib.nbody_set_option(nb.OPT_ISA_PHASE, phase) == nb._lib.ERR_UNSUPPORTED, phase
    assert lib.nbody_set_option(nb.OPT_ISA_PHASE, 21) == nb._lib.ERR_ARG
    assert lib.nbody_set_option(nb.OPT_ISA_PHASE, 1) == 0


def test_enum_values_of_the_header_and_the_binding_agree(nb):
    """the Python mirror restates include/nbody.h's enumerators by value: a drift would silently set the wrong option"""
    src = open(os.path.join(ROOT, "include", "nbody.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    vals = {k: int(v) for k, v in re.findall(r"\b(NBODY_[A-Z0-9_]+)\s*=\s*(-?\d+)", src)}
    vals.update({k: int(v) for k, v in re.findall(r"#define\s+(NBODY_[A-Z0-9_]+)\s+(-?\d+)", src)})
    L = nb._lib
    pairs = {"NBODY_OPT_VARIANT": L.OPT_VARIANT, "NBODY_OPT_IBLOCK": L.OPT_IBLOCK, "NBODY_OPT_JSUB": L.OPT_JSUB, "NBODY_OPT_JSLICES": L.OPT_JSLICES,
             "NBODY_OPT_ARITH": L.OPT_ARITH, "NBODY_OPT_SUM_ORDER": L.OPT_SUM_ORDER, "NBODY_OPT_TIMING": L.OPT_TIMING, "NBODY_OPT_COMM": L.OPT_COMM,
             "NBODY_OPT_OVERLAP": L.OPT_OVERLAP, "NBODY_OPT_ISA_PHASE": L.OPT_ISA_PHASE, "NBODY_OPT_WAVES_PER_SIMD": L.OPT_WAVES_PER_SIMD,
             "NBODY_OPT_GRAPH": L.OPT_GRAPH, "NBODY_OPT_SUM_BLOCK": L.OPT_SUM_BLOCK, "NBODY_OPT_FUSE_COMBINE": L.OPT_FUSE_COMBINE,
             "NBODY_OPT_ISA_LONG_BUFFERS": L.OPT_ISA_LONG_BUFFERS, "NBODY_OPT_XCD_MAP": L.OPT_XCD_MAP, "NBODY_OPT_WSPLIT": L.OPT_WSPLIT,
             "NBODY_VARIANT_AUTO": L.VARIANT_AUTO, "NBODY_VARIANT_SMEM": L.VARIANT_SMEM, "NBODY_VARIANT_LDS": L.VARIANT_LDS,
             "NBODY_VARIANT_READLANE": L.VARIANT_READLANE, "NBODY_VARIANT_ISA": L.VARIANT_ISA,
             "NBODY_ARITH_FMA3": L.ARITH_FMA3, "NBODY_ARITH_REFERENCE": L.ARITH_REFERENCE, "NBODY_ARITH_STRICT": L.ARITH_STRICT,
             "NBODY_ARITH_REFERENCE_STRICT": L.ARITH_REFERENCE_STRICT, "NBODY_SUM_SEQ": L.SUM_SEQ, "NBODY_SUM_FPGA16": L.SUM_FPGA16,
             "NBODY_SUM_BLOCKED": L.SUM_BLOCKED, "NBODY_COMM_RING": L.COMM_RING, "NBODY_COMM_ALLGATHER": L.COMM_ALLGATHER,
             "NBODY_COMM_AUTO": L.COMM_AUTO, "NBODY_COMM_DIRECT": L.COMM_DIRECT,
             "NBODY_ERR_NOT_INIT": L.ERR_NOT_INIT, "NBODY_ERR_ARG": L.ERR_ARG, "NBODY_ERR_NO_DEVICE": L.ERR_NO_DEVICE,
             "NBODY_ERR_RCCL_LOAD": L.ERR_RCCL_LOAD, "NBODY_ERR_STATE": L.ERR_STATE, "NBODY_ERR_UNSUPPORTED": L.ERR_UNSUPPORTED}
    missing = [k for k in pairs if k not in vals]
    assert not missing, missing
    assert {k: vals[k] for k in pairs} == pairs
