"""mini_nbody_amd — MI355X-native all-pairs N-body force path.

The package holds only what the path needs: csrc/ (HIP kernels + the C-ABI of
include/nbody.h), host/ (the C host program) and this thin Python mirror of the
interface.  (`mini-nbody_amd/` at the repository root is a one-line alias
for the directory's old, hyphenated name.)
"""
from . import _lib, bodies, mailbox, sharding  # noqa: F401
from ._lib import (OPT_GRAPH, ARITH_FMA3, ARITH_REFERENCE, ARITH_REFERENCE_STRICT, ARITH_STRICT, COMM_ALLGATHER, COMM_AUTO, COMM_DIRECT, COMM_RING, OPT_ARITH, OPT_COMM, OPT_IBLOCK,  # noqa: F401
                   OPT_FUSE_COMBINE, OPT_ISA_LONG_BUFFERS, OPT_ISA_PHASE, OPT_JSLICES, OPT_JSUB, OPT_OVERLAP, OPT_SUM_BLOCK, OPT_SUM_ORDER, OPT_TIMING, OPT_VARIANT,
                   OPT_WAVES_PER_SIMD, OPT_WSPLIT, OPT_XCD_MAP, SUM_BLOCKED, SUM_FPGA16, SUM_SEQ,
                   VARIANT_AUTO, VARIANT_ISA, VARIANT_LDS, VARIANT_READLANE, VARIANT_SMEM, NBodyError)
from .bodies import make_bodies  # noqa: F401
from .engine import NBody, comm_plan, rsqrt_selftest, rsqrt_strict, strict_proof, unique_id  # noqa: F401
from .mailbox import Mailbox  # noqa: F401

__all__ = ["NBody", "Mailbox", "NBodyError", "make_bodies", "unique_id", "bodies", "mailbox", "sharding"]
