"""Lifecycle and failure-injection cases, ONE CHILD PROCESS EACH (tests/hazard_cases.py): a segmentation fault, a hang or an
abort inside the library is then one failed test with the child's output attached, and the parity suite's process --
and its pass count -- survives.  Collected last (tests/conftest.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    "destroyed_stream_then_another_stream",
    "stream_churn",
    "destroy_context_after_its_last_stream_is_gone",
    "fleet_rank_fails_before_the_collective",
    "fleet_rccl_rank_fails_before_the_collective",
    "fleet_rccl_rank_fails_inside_the_collective_phase",
    "fleet_host_sum_rank_fails_after_the_rendezvous",
]


def test_the_case_list_is_complete():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hazard_cases
    assert sorted(hazard_cases.CASES) == sorted(CASES)


@pytest.mark.parametrize("case", CASES)
def test_hazard_case_in_a_child_process(case, hwmc_mod):
    env = dict(os.environ)
    env.pop("HWMC_COMM_INJECT_FAIL", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hazard_cases.py"), case], capture_output=True, text=True,
                       timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0 and ("OK " + case) in r.stdout, "rc=%d\n%s\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
