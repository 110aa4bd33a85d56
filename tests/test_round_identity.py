"""The two-instruction rounding of the reconstruction scatter (csrc/fsg_slice_acq.hip, sa_round_pos): floor(x + pred(0.5)) is
roundf(x) -- half away from zero, what the reference's CUDA kernel calls (slice_acq_cuda_kernel.cu:540-542) -- for EVERY float
0 <= x < 2^23.  Checked exhaustively once (1 258 291 200 floats, 68 s of numpy, 0 mismatches); here the dangerous neighbourhoods
(ties n + 0.5 and the floats around them in every binade) plus a few million random values."""
import numpy as np

PRED_HALF = np.uint32(0x3EFFFFFF).view(np.float32)  # 0.49999997, the float below 0.5: the constant in the kernel


def _roundf_pos(x):
    t = np.trunc(x)
    return t + ((x - t) >= np.float32(0.5)).astype(np.float32)


def _fast(x):
    return np.floor(x + PRED_HALF)  # float32 add (round to nearest even), then floor


def test_constant_is_the_float_below_one_half():
    assert np.nextafter(np.float32(0.5), np.float32(0)) == PRED_HALF


def test_ties_and_their_neighbours_in_every_binade():
    n = np.concatenate([np.arange(0, 4096), 2 ** np.arange(12, 23) - 1, 2 ** np.arange(12, 23), 2 ** np.arange(12, 22) + 1])
    ties = (n.astype(np.float64) + 0.5).astype(np.float32)  # exact below 2^23
    bits = ties.view(np.uint32).astype(np.int64)[:, None] + np.arange(-8, 9)[None, :]
    x = bits.clip(0, None).astype(np.uint32).view(np.float32).ravel()
    x = x[(x >= 0) & (x < np.float32(2 ** 23))]
    assert np.array_equal(_fast(x), _roundf_pos(x))
    # the one value that breaks floor(x + 0.5): the float below 0.5 must round to 0
    below = np.array([PRED_HALF], np.float32)
    assert np.floor(below + np.float32(0.5))[0] == 1.0 and _fast(below)[0] == 0.0 and _roundf_pos(below)[0] == 0.0


def test_random_floats_by_bit_pattern():
    rng = np.random.default_rng(7)
    hi = int(np.float32(2 ** 23).view(np.uint32))
    x = rng.integers(0, hi, 4_000_000, dtype=np.uint32).view(np.float32)
    assert np.array_equal(_fast(x), _roundf_pos(x))
