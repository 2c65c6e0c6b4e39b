"""GPU tier, end to end: geo model + agent loop through the HIP path vs oracle and golden."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["e2e_small", "e2e_native"])
def test_registration_iteration(case):
    import parity_e2e
    parity_e2e.run_case(case, check_golden=True, verbose=True)


def test_registration_iteration_baseline_config0():
    """BASELINE.json configs[0] sizes (batch 1, 4096 points, 176x608 -> 192x608, 1 agent step) against the oracle."""
    import parity_e2e
    parity_e2e.run_case("e2e_config0", check_golden=False, verbose=True)


def test_registration_iteration_op_level_paths(monkeypatch):
    """Same iteration with every layer-level fusion and the side streams switched off: the op-level composition
    must meet the same oracle / golden bars."""
    import parity_e2e
    from cmr_agent_amd.models import LinearAttention as LA, PointNN, _vit, CMRAgent
    from cmr_agent_amd.utils import streams
    monkeypatch.setattr(LA.LinearAttention, "FUSED", False)
    monkeypatch.setattr(PointNN, "FUSED_FRONT", False)
    monkeypatch.setattr(_vit.Block, "FUSED", False)
    monkeypatch.setattr(CMRAgent, "FUSED_TAIL", False)
    monkeypatch.setattr(streams, "ENABLED", False)
    parity_e2e.run_case("e2e_small", check_golden=True, verbose=False)
