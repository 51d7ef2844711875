"""pytest configuration: registers the `gpu` marker and puts the product package on sys.path.

`-m "not gpu"` tests: oracle vs golden vectors, host logic, C-ABI symbol export (no compute calls).
`-m gpu` tests: parity of the HIP path against the oracle, through the C-ABI, on a real MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "humanoid-gym_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The Philox-vs-oracle tests forgive (count, re-synchronise, bound) `low_speed` threshold flips (tests/synth_common.py);
    the counts are printed here so that they are in the log of a PASSING run too."""
    sc = sys.modules.get("synth_common")
    rep = getattr(sc, "REPORT", None) if sc is not None else None
    if rep:
        terminalreporter.write_sep("-", "internal-Philox parity runs: forgiven low_speed threshold flips")
        for what, flips in rep:
            terminalreporter.write_line("flips forgiven: %d  | %s" % (flips, what))
    br = sys.modules.get("bf16_report")
    brep = getattr(br, "REPORT", None) if br is not None else None
    if brep:
        terminalreporter.write_sep("-", "bf16 MFMA path: measured error vs the asserted bar (SURVEY.md 8c: 1e-2)")
        for what, err, bar in brep:
            terminalreporter.write_line("bf16 err %.3e  (bar %.0e)  | %s" % (err, bar, what))
