"""The exactness arguments behind round 5's limb_assign_kernel / nms_refine_kernel (csrc/decode.hip): the reference's
double-precision steps (lib/pafprocess/pafprocess.cpp:220-246 `get_score`, `roundpaf`; paf_to_pose.py:382 for the x8
nearest index; cv2.resize's (dx + 0.5) * scale - 0.5) are evaluated there WITHOUT double-precision instructions wherever
that is exact.  These tests check the identities themselves, in numpy, over the value ranges the kernels see - the GPU
parity tests (tests/test_decode_gpu.py) check the kernels."""
import numpy as np
import pytest


def _kernel_round(v):
    """limb_assign_kernel: lx = trunc(v); if (v - (float)lx >= 0.5f) ++lx   (all fp32)"""
    v = np.asarray(v, np.float32)
    t = v.astype(np.int32)                                  # truncation; v >= 0
    frac = (v - t.astype(np.float32)).astype(np.float32)
    return t + (frac >= np.float32(0.5))


def _reference_round(v):
    """pafprocess.cpp roundpaf: (int)(v + 0.5) with v widened to double"""
    return (np.asarray(v, np.float32).astype(np.float64) + 0.5).astype(np.int64)   # trunc; v >= 0


def test_round_half_up_of_a_float_without_doubles():
    rng = np.random.default_rng(0)
    v = rng.uniform(0, 4096, 2_000_000).astype(np.float32)
    assert np.array_equal(_kernel_round(v), _reference_round(v))
    # the values a line integral actually produces: A + i * ((B - A) / 10) in fp32, integer end points
    a = rng.integers(0, 3000, 400_000).astype(np.float32)
    b = rng.integers(0, 3000, 400_000).astype(np.float32)
    step = ((b - a) / np.float32(10.0)).astype(np.float32)
    for i in range(10):
        x = (a + np.float32(i) * step).astype(np.float32)
        assert x.min() >= 0
        assert np.array_equal(_kernel_round(x), _reference_round(x)), i
    # every half and its fp32 neighbours, across binades (2047.5 -> 2048.5: the spacing doubles)
    k = np.arange(0, 8192, dtype=np.float32)
    for base in (k + np.float32(0.5), np.nextafter(k + np.float32(0.5), np.float32(0)),
                 np.nextafter(k + np.float32(0.5), np.float32(1e9)), k, np.nextafter(k + 1, np.float32(0))):
        assert np.array_equal(_kernel_round(base), _reference_round(base))


@pytest.mark.parametrize("up", [1, 2, 4, 8, 16])
def test_nearest_upsampling_index_is_a_shift_for_powers_of_two(up):
    lx = np.arange(0, 70_000, dtype=np.int64)
    ref = np.floor(lx.astype(np.float64) * (1.0 / up)).astype(np.int64)   # floor((double)lx * inv_up)
    assert np.array_equal(ref, lx >> int(np.log2(up)))


def test_length_penalty_is_zero_for_limbs_up_to_half_the_image():
    """crit2 = (float)((double)(scores / 10.f) + min(0.5 * h1 / norm - 1, 0)): the kernel takes the double path only when
    2.f * norm > (float)h1, and `(double)norm < 1e-12` is `!(norm > 0)` (norm = sqrtf of a sum of integer squares)."""
    dx, dy = np.meshgrid(np.arange(-400, 401), np.arange(-400, 401))
    n2 = (dx.astype(np.float32) * dx.astype(np.float32) + dy.astype(np.float32) * dy.astype(np.float32)).astype(np.float32)
    norm = np.sqrt(n2).astype(np.float32).ravel()
    assert np.array_equal(norm.astype(np.float64) < 1e-12, ~(norm > 0))
    norm = norm[norm > 0]
    s = np.random.default_rng(1).uniform(-2, 2, norm.size).astype(np.float32)
    for h1 in (368, 184, 46 * 8, 100, 1):
        pen = np.minimum(0.5 * h1 / norm.astype(np.float64) - 1.0, 0.0)
        ref = (s.astype(np.float64) + pen).astype(np.float32)
        long_limb = (np.float32(2.0) * norm) > np.float32(h1)
        assert np.all(pen[~long_limb] == 0.0) and np.all(pen[long_limb] < 0.0)
        got = np.where(long_limb, ref, s)                    # short limbs: the fp32 value as it is
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), h1


@pytest.mark.parametrize("up", [1, 2, 4, 8, 16])
def test_resize_source_coordinate_in_fp32_for_powers_of_two(up):
    """nms_refine_kernel: fx = (float)((tid + 0.5) * inv_up - 0.5) evaluated in double == the same in fp32 when up is a
    power of two (every intermediate is exactly representable)."""
    tid = np.arange(0, 5 * up)
    ref = ((tid.astype(np.float64) + 0.5) * (1.0 / up) - 0.5).astype(np.float32)
    got = ((tid.astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / up) - np.float32(0.5)).astype(np.float32)
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
