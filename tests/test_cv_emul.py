"""Self-checks of the OpenCV fixed-point emulation (oracle/cv_emul.py; PARITY UNPINNED: cv2 is absent) and the stated
deviation of a float bilinear warp from it -- the bound SURVEY 8(f) N1 asks for."""
import numpy as np

from centerpose_amd.lib.utils.image import get_affine_transform, warp_affine_bilinear
from oracle import cv_emul as cv


def test_identity_and_integer_translation_are_exact():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (40, 56, 3)).astype(np.uint8)
    np.testing.assert_array_equal(cv.warp_affine_u8(img, [[1, 0, 0], [0, 1, 0]], (56, 40)), img)
    out = cv.warp_affine_u8(img, [[1, 0, 5], [0, 1, -3]], (56, 40))
    np.testing.assert_array_equal(out[0:37, 5:], img[3:40, :51])
    assert out[:, :5].max() == 0 and out[37:].max() == 0          # constant border 0
    np.testing.assert_array_equal(cv.resize_linear_u8(img, (56, 40)), img)


def test_half_pixel_shift_rounds_to_nearest_of_the_average():
    img = np.array([[10, 20, 41, 0]], np.uint8).repeat(4, 0)
    out = cv.warp_affine_u8(img, [[1, 0, 0.5], [0, 1, 0]], (4, 4))     # dst x samples src x - 0.5
    assert out[1].tolist() == [5, 15, 31, 21]                          # (0+10)/2, (10+20)/2, (20+41)/2 -> 30.5 -> 31, 20.5 -> 21


def test_resize_of_a_ramp_and_downscale_by_two():
    ramp = np.arange(0, 64, dtype=np.uint8)[None, :].repeat(8, 0)
    up = cv.resize_linear_u8(ramp, (128, 8))
    # fx = (dx + 0.5) / 2 - 0.5: samples at quarter positions, clamped at the ends
    expect = np.clip(np.round((np.arange(128) + 0.5) / 2 - 0.5 + 1e-9), 0, 63)
    assert np.abs(up[0].astype(int) - expect).max() <= 1
    down = cv.resize_linear_u8(ramp, (32, 4))
    np.testing.assert_array_equal(down[0], ((ramp[0, 0::2].astype(int) + ramp[0, 1::2] + 1) // 2))


def test_float_bilinear_deviates_by_at_most_one_grey_level():
    """The bound for a float warp (the mirror's CPU stand-in) against the 5-bit fixed-point one: coordinates are
    quantised to 1/32 px (|grad| <= 255 per px -> <= 255/64 in theory on a step edge; on natural / random frames the
    observed maximum is what matters for the network input), and the result is rounded to an integer (0.5)."""
    rng = np.random.RandomState(3)
    base = rng.randint(0, 256, (60, 80, 3)).astype(np.float64)
    # smooth the random frame a little so that it resembles an image rather than white noise
    img = ((base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) / 4).astype(np.uint8)
    c = np.array([40.0, 30.0], np.float32)
    M = get_affine_transform(c, 80.0, 0, [128, 128])
    fixed = cv.warp_affine_u8(img, M, (128, 128)).astype(np.float64)
    flt = warp_affine_bilinear(img, M, 128, 128)
    d = np.abs(fixed - flt)
    assert d.max() <= 2.5 and d.mean() < 0.5, (d.max(), d.mean())
