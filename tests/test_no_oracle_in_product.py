"""The product package must not route through the oracle or any CPU stand-in."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_never_touches_oracle_or_emulation():
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "decombinator_amd")):
        for fn in fns:
            if not fn.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                continue
            src = open(os.path.join(dp, fn), errors="replace").read()
            if re.search(r"(from|import)\s+oracle|oracle/|dcr_oracle|host_emul/|libdcrx_emul|refshim", src):
                if fn == "dcrx_dcr_device.h" and "tests/host_emul/" in src:
                    # a comment naming the test-only harness that includes this header
                    src2 = re.sub(r"//.*", "", src)
                    if not re.search(r"oracle|host_emul|refshim", src2):
                        continue
                bad.append(os.path.join(dp, fn))
    assert not bad, bad
