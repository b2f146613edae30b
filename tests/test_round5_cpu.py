"""CPU: host logic added in round 5 -- the train steps' default arithmetic policy, the policy context, plan caches per policy."""
import os

import pytest

from tpgsr_amd import kernels as K


def test_train_step_policy_precedence(monkeypatch):
    # nobody chose: the benchmarked, gated policy
    monkeypatch.setattr(K, "_POLICY_EXPLICIT", False)
    assert K.train_step_policy(None) == "x2"
    # the step's own argument wins over everything
    assert K.train_step_policy("x3") == "x3"
    with pytest.raises(ValueError):
        K.train_step_policy("fp8")
    # an explicit choice (TPGSR_CONV_PREC / set_conv_prec) is honoured
    prev = K.POLICY
    try:
        K.set_conv_prec("x3b2")
        assert K.train_step_policy(None) == "x3b2"
    finally:
        K.set_conv_prec(prev)
    with pytest.raises(ValueError):
        K.set_conv_prec("nope")


def test_policy_context_restores_and_is_not_an_explicit_choice(monkeypatch):
    monkeypatch.setattr(K, "_POLICY_EXPLICIT", False)
    prev = (K.POLICY, K.CONV_TERMS)
    with K.policy("x2"):
        assert K.POLICY == "x2" and K.CONV_TERMS == 2
        with K.policy("f32"):
            assert K.CONV_TERMS == 0
        assert K.POLICY == "x2"
    assert (K.POLICY, K.CONV_TERMS) == prev and not K._POLICY_EXPLICIT


def test_package_import_asks_for_eight_hardware_queues():
    import tpgsr_amd  # noqa: F401
    assert os.environ.get("GPU_MAX_HW_QUEUES")            # set by the package unless the caller chose a value
