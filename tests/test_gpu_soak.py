"""A fixed-seed slice of the randomised differential soak (tests/gpu_soak.py; 1 500 iterations of it ran in round 3) under pytest -m gpu:
60 iterations of random batch sizes (1 .. 3 000), context capacities (1 .. 4 096: slices), chain counts, both context modes, host /
device / serial / many-batches entry points, valid batches and batches with a changed message, swapped signatures, an infinity key or
an infinity signature - verdict AND final GT value equal to the C restatement of the reference every time."""
import pytest

pytestmark = pytest.mark.gpu


def test_soak_slice_fixed_seed():
    import __graft_entry__ as ge
    ge.build()
    import gpu_soak
    assert gpu_soak.soak(60, 2024, m=ge.load_package(), verbose=False) == 60
