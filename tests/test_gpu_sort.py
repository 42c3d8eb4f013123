"""The queue-ordering step (rt_debug_sort_keys) returns a stable permutation for
every size class of the underlying device sort."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 4096, 4353, 49152, 300000, 1 << 20, (1 << 22) - 1, 1 << 22, 5_000_000])
@pytest.mark.parametrize("bits", [21, 30, 32])
def test_sort_is_a_stable_permutation(ctx, n, bits):
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 1 << 30, size=n, dtype=np.uint32)
    keys[rng.random(n) < 0.1] = 0xFFFFFFFF                      # unused queue tail
    keys &= np.uint32((0xFFFFFFFF << (32 - bits)) & 0xFFFFFFFF)  # callers zero the bits they do not sort by
    perm = ctx.debug_sort_keys(keys, bits)
    ref = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(perm, ref)
