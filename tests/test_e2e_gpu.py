"""GPU tier, end to end: geo model + agent loop through the HIP path vs oracle and golden."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["e2e_small", "e2e_native"])
def test_registration_iteration(case):
    import parity_e2e
    parity_e2e.run_case(case, check_golden=True, verbose=True)
