"""Collect the worst cases of the borrowed-decision audit (tests/_pins.py) over the pinned -m gpu tests WITHOUT asserting its bounds --
the mode used to set NOISE_C / SHARE_K / MIN_PROPOSAL_MATCH.  It lives here, not behind an environment switch inside the test
suite: `pytest tests/` always asserts (round-4 verdict, Next 8).

    python tools/pins_measure.py [-k expr]          # on a GPU box; prints one summary line per audited test + the overall worst

It runs pytest in-process with `assert_borrowed_decisions_are_noise` replaced by the collecting form, so the tests' OTHER assertions
(losses, gradients) still apply.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(argv):
    import pytest
    import _pins
    collected = []

    def collecting(holder, label=""):
        summary, problems = _pins.audit_borrowed_decisions(holder, label)
        collected.append((label, summary, problems))
        return summary

    _pins.assert_borrowed_decisions_are_noise = collecting
    os.environ["HD_PINS_AUDIT"] = "1"
    rc = pytest.main([os.path.join(ROOT, "tests"), "-m", "gpu", "-q", "-s", "-x"] + argv)
    print("\n==== %d audits ====" % len(collected))
    for label, summary, problems in collected:
        print(("OUTSIDE " if problems else "ok      ") + summary)
        for pr in problems[:8]:
            print("         ", pr)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
