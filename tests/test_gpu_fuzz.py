"""A short randomised parity run of the MSM against the C oracle (tools/fuzz_parity.py): sizes across the
front ends and their boundaries, both curves, scalars on both sides of r/2, special values, identity bases,
overridden window widths."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [11, 12])
def test_msm_randomised_parity(ctx, coracle, seed):
    import fuzz_parity

    cases, bad = fuzz_parity.run(ctx, coracle, 8.0, seed)
    assert bad is None, bad
    assert cases >= 20
