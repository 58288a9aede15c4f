"""The TEST build of the HIP engine: the product's sources compiled with -DLT_TEST_HOOKS into
``csrc/libltrans_hip_test.so`` (built by ``__graft_entry__.build()``).  It exports, beside the whole ABI,
the test aids ``lt_test_*`` (device leaf numerics on caller-supplied inputs, the raw random stream, ...)
that the product library ``libltrans_hip.so`` does not carry.  Only the leaf tests load it; every parity
test of the hot path runs through the product library."""
import ctypes as C
import os

import ltrans_amd

HERE = os.path.dirname(os.path.abspath(__file__))
TEST_LIBRARY = os.path.join(os.path.dirname(HERE), "ltransv.2b_amd", "csrc", "libltrans_hip_test.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(TEST_LIBRARY):
            raise RuntimeError(f"{TEST_LIBRARY} not found: python -c 'import __graft_entry__ as g; g.build()'")
        _lib = C.CDLL(TEST_LIBRARY, mode=C.RTLD_LOCAL)
    return _lib


class HipTestEngine(ltrans_amd.Engine):
    """an engine of the test build (for the aids that take an engine handle)"""

    def __init__(self, cfg):
        self._open(cfg, load(), "lt_")

    def allgather_layout(self, nranks, rank, counts, others):
        """lt_test_allgather_layout: this engine as rank `rank` of `nranks`; `others` = xyz(sum of the other
        counts, 3) of the other ranks in rank order -> xyz(sum(counts), 3) as lt_allgather_positions assembles it"""
        import numpy as np
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        others = np.asfortranarray(others, dtype=np.float64)
        out = np.full((int(counts.sum()), 3), -7.0, order="F")
        f = self.lib.lt_test_allgather_layout
        f.restype = C.c_int
        self._check(f(self._h, int(nranks), int(rank), counts.ctypes.data_as(C.c_void_p),
                      others.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "test_allgather_layout")
        return out
