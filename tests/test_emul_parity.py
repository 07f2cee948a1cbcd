"""CPU run of the device per-read code (tests/host_emul, a test-only host compile
of decombinator_amd/csrc/dcrx_dcr_device.h) against the golden vectors and the
oracle.  Debug/sanitizer aid; the parity tests proper are tests/test_gpu_parity.py."""
import pytest

from decombinator_amd import _native as nat
from tests import golden_util as gu
from tests import parity_util as pu


@pytest.mark.parametrize("path", gu.golden_files(), ids=lambda p: p.split("/")[-1])
@pytest.mark.parametrize("flags", [0, nat.F_ONE_BASE_SCAN, nat.F_FORCE_SLOW_READER], ids=["pairscan", "onebase", "slowreader"])
def test_emul_matches_golden_and_oracle(path, flags):
    assert pu.check_fixture("emul", path, flags) > 500


def test_emul_synthetic_orientations_and_ragged_lengths():
    from tests import emul_extended
    assert emul_extended.run(60000)
