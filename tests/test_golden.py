"""Regression pin: the oracle reproduces the committed golden rollout (tests/golden/make_golden.py)."""
import os

import numpy as np

from tests.golden import make_golden


def test_oracle_reproduces_golden_rollout():
    ref = np.load(os.path.join(os.path.dirname(make_golden.__file__), "oracle_rollout.npz"))
    cur = make_golden.make()
    for k in ("actions", "ep0"):
        assert np.array_equal(ref[k], cur[k]), k
    # fp64 oracle on the same machine class: tight; contact events make later steps sensitive, so compare the early part tightly
    assert np.allclose(ref["aux"][:6], cur["aux"][:6], atol=1e-6)
    assert np.allclose(ref["actor"][:6], cur["actor"][:6], atol=1e-6)
    assert np.allclose(ref["reward"][:5], cur["reward"][:5], atol=1e-6)
    assert np.array_equal(ref["aux"][:, :, 70], cur["aux"][:, :, 70])
