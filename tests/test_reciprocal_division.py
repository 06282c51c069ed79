"""The bounded-loss kernels divide the running sum by the ring's length with a multiplication (lossy_kernels.hip: lossy_div_magic /
lossy_div): floor(sum * M / 2^32), M = floor(2^32 / n) + 1.  The claim that this is the exact quotient for every sum the state can
hold (at most 64 images of 16 bits: sum < 2^22) and every length 1..64 is checked here exhaustively, on the CPU."""
import numpy as np


def test_reciprocal_division_is_exact_for_every_sum_and_length():
    sums = np.arange(0, 64 * 65535 + 1, dtype=np.uint64)
    assert int(sums[-1]) < 1 << 22
    for n in range(1, 65):
        if n == 1:
            q = sums  # lossy_div: magic 0 means "the sum itself"
        else:
            magic = np.uint64((1 << 32) // n + 1)
            q = (sums * magic) >> np.uint64(32)
        assert np.array_equal(q, sums // np.uint64(n)), n
