"""The failure probability each bootstrapping-key flavour is documented with follows from the
committed noise measurement (profiles/r03/noise_probe.json, tools/noise_probe.py on the GPU) and
the constants the reference's ClientKey fixture stores (tests/golden/privateKey.bin: log2_p_fail,
the modulus-switch noise-reduction bound, sigma factor and input variance)."""
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# what include/lfhe.h and the project's documentation state per flavour: variance of a bootstrap's output
# relative to the one-bit key over 2^64, and log2 p_fail of a candidate the noise reduction accepts
# at its bound (the worst case it lets through) at noise level 5
DOCUMENTED = {
    "q64": (1.00, -130.2),
    "prime": (1.01, -130.0),
    "grouped": (1.38, -129.2),
    "q64_grouped": (1.38, -129.1),
    "grouped3": (1.46, -128.9),
    "q64_grouped3": (1.43, -129.1),
}


@pytest.fixture(scope="module")
def analysis():
    import pfail
    return pfail.analyse()


def test_published_constants(analysis):
    pub = analysis["published"]
    assert pub["log2_p_fail"] == pytest.approx(-129.15284804376165)
    assert pub["max_noise_level"] == 5 and pub["bound"] == 2.0 ** 58
    # the set's input variance is its key-switch noise plus 25 bootstrap outputs: the formula value of the
    # key switch leaves room for a bootstrap output of about 2^98.2
    room = 2.0 ** pub["input_variance_log2"] - 2.0 ** analysis["ks_variance_log2_formula"]
    assert 97.5 < math.log2(room / 25) < 98.6


def test_every_flavour_meets_the_sets_promise(analysis):
    for name, k in analysis["keys"].items():
        assert k["samples"] >= 1 << 17, name
        assert k["log2_p_fail"] < -128.0, (name, k["log2_p_fail"])
        assert k["log2_p_fail_worst_accepted"] < -128.0, (name, k["log2_p_fail_worst_accepted"])
        # measured key-switch noise agrees with its formula (the probe measures what it says it measures)
        assert abs(k["v_ks_log2_measured"] - analysis["ks_variance_log2_formula"]) < 0.05, name
        # decision noise actually observed: every one of the samples decoded with a margin
        assert k["decision_noise_log2_max"] < 57.0, name


def test_documented_figures_follow_from_the_probe(analysis):
    for name, (ratio, worst) in DOCUMENTED.items():
        k = analysis["keys"][name]
        assert k["v_br_ratio_to_q64"] == pytest.approx(ratio, abs=0.015), name
        assert k["log2_p_fail_worst_accepted"] == pytest.approx(worst, abs=0.06), name
    # the pairs of key bits double the key-noise term of the blind rotation and leave the rounding term:
    # predicted ratio (2 * 0.385 + 0.615) with the measured split of the one-bit key's variance
    assert 1.30 < analysis["keys"]["grouped"]["v_br_ratio_to_q64"] < 1.45


def test_measured_bootstrap_noise_agrees_with_its_formula(analysis):
    """Output variance of a bootstrap by formula (key noise of every GGSW of a step + decomposition
    rounding, tools/pfail.py) against the GPU measurement, for one, two and three key bits per step and
    both moduli: within 0.06 in the exponent (2^98.00, 2^98.48, 2^98.55)."""
    import pfail
    assert pfail.blind_rotation_variance_log2(1) == pytest.approx(97.995, abs=0.01)
    assert pfail.blind_rotation_variance_log2(2) == pytest.approx(98.48, abs=0.01)
    assert pfail.blind_rotation_variance_log2(3) == pytest.approx(98.547, abs=0.01)
    for name, k in analysis["keys"].items():
        assert abs(k["v_br_log2"] - k["v_br_log2_formula"]) < 0.06, (name, k["v_br_log2"], k["v_br_log2_formula"])


def test_pfail_formula_against_a_closed_form():
    import numpy as np
    import pfail
    # zero rounding error, input noise sigma = 2^58 / 10: p = 2 Q(10)
    got = pfail.log2_pfail(2 * (58 - math.log2(10)), np.zeros(4, dtype=np.float32), np.zeros(4, dtype=np.float32))
    assert got == pytest.approx(math.log2(math.erfc(10 / math.sqrt(2))), abs=1e-6)
