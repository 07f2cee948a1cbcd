"""Runs the device per-read code (tests/host_emul) and the oracle under ASan/UBSan on the
golden fixtures.  Started by tests/test_sanitizers.py in a subprocess with libasan preloaded."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from decombinator_amd import _native as nat  # noqa: E402
from tests import golden_util as gu  # noqa: E402
from tests import parity_util as pu  # noqa: E402


def main():
    emul = C.CDLL(os.path.join(ROOT, "tests", "host_emul", "build", "libdcrx_emul_asan.so"))
    emul.emul_decombine.restype = C.c_int
    emul.emul_decombine.argtypes = [C.POINTER(nat.TagSetC), C.POINTER(nat.CfgC), C.POINTER(nat.BatchC), C.c_void_p,
                                    C.c_void_p, C.c_char_p, C.c_int]
    n = 0
    for path in gu.golden_files():
        fx = gu.load(path)
        tsc, keep = pu.tagset_c(fx["tagset"])
        for orientation in ("reverse", "forward", "both"):
            reads = [c["read"] for c in fx["cases"] if len(c["read"]) <= 320][:1500]
            batch = nat.pack_reads(reads)
            for flags in (0, nat.F_ONE_BASE_SCAN, nat.F_FORCE_SLOW_READER):
                cfg = nat.make_cfg(orientation, False, 130, flags)
                rec = np.zeros(batch.n_reads, dtype=nat.RECORD_DTYPE)
                cnt = np.zeros(nat.N_COUNTERS, dtype=np.uint64)
                b = batch.as_c()
                err = C.create_string_buffer(256)
                rc = emul.emul_decombine(C.byref(tsc), C.byref(cfg), C.byref(b), rec.ctypes.data, cnt.ctypes.data, err, 256)
                assert rc == 0, err.value
                n += batch.n_reads
    print("ASAN_OK", n)


if __name__ == "__main__":
    main()
