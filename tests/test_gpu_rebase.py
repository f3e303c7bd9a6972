"""GPU test of the sessions that replace their generators by the folded ones (the path of large
inner-product arguments), forced at a small size so that the Python oracle can follow every round."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_ipa_session_rebase_against_oracle():
    env = dict(os.environ, BP_IPA_REBASE_MIN="16", BP_IPA_REBASE_M="4")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rebase_check.py")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0 and "rebase check passed" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_ipa_session_rebase_from_tables_one_lane_per_output():
    """the replacement over window tables with one list of steps for all lanes (what large sessions run)"""
    env = dict(os.environ, BP_IPA_REBASE_MIN="16", BP_IPA_REBASE_M="4", BP_IPA_REBASE_LOG_S="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rebase_check.py")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0 and "rebase check passed" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
