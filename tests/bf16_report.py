"""Measured errors of the bf16 MFMA path against fp32 references, collected while the `-m gpu` tests run and printed in pytest's terminal
summary (tests/conftest.py) -- so the log of a PASSING run shows how far below (or above) SURVEY.md 8c's 1e-2 bar each tensor sits.

`check(what, err, bar)` asserts err <= bar and records (what, err, bar).  The bar is 1e-2 wherever the measured error holds it; a site that
needs more says so with its own `bar=` and a comment giving the measured value and the reason."""
BF16_BAR = 1e-2          # "bf16 MFMA path <= 1e-2 relative"
REPORT = []


def check(what, err, bar=BF16_BAR):
    err = float(err)
    REPORT.append((what, err, bar))
    assert err <= bar, "%s: measured %.3e > bar %.1e" % (what, err, bar)
    return err
