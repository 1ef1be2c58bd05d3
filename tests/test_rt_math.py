"""include/rt_math.h: accuracy of the fixed-arithmetic exp2 / exp / log2 / acos against double
precision, special values, and a few known answers that pin the bit patterns (the same header is
compiled into the kernels, so these are the values the GPU computes as well)."""
import numpy as np

import oracle_lib as O


def ulps(y, ref):
    ref32 = ref.astype(np.float32)
    return np.abs(y.astype(np.float64) - ref) / np.spacing(np.abs(ref32)).astype(np.float64)


def test_accuracy_against_double():
    rng = np.random.default_rng(7)
    x = rng.uniform(-125, 127, 400000).astype(np.float32)
    assert ulps(O.math_apply("exp2", x), np.exp2(x.astype(np.float64))).max() < 2.0
    x = rng.uniform(-86, 88, 400000).astype(np.float32)
    assert ulps(O.math_apply("exp", x), np.exp(x.astype(np.float64))).max() < 2.0
    x = np.exp(rng.uniform(-80, 80, 400000)).astype(np.float32)
    ref = np.log2(x.astype(np.float64))
    far = np.abs(ref) > 0.01
    assert ulps(O.math_apply("log2", x)[far], ref[far]).max() < 4.0
    x = rng.uniform(0.5, 2.0, 400000).astype(np.float32)
    assert np.abs(O.math_apply("log2", x) - np.log2(x.astype(np.float64))).max() < 2e-7
    x = rng.uniform(-1, 1, 400000).astype(np.float32)
    assert ulps(O.math_apply("acos", x), np.arccos(x.astype(np.float64))).max() < 4.0
    x = (1.0 - np.exp(rng.uniform(-20, -1, 200000))).astype(np.float32)
    ref = np.arccos(x.astype(np.float64))
    assert ulps(O.math_apply("acos", x)[ref > 0], ref[ref > 0]).max() < 4.0


def test_division_by_10000_is_exact():
    """The generator's `float(seed % 10000) / 10000.0` (Random.glsl:11) is computed as a product and one fused
    correction step: every one of the 10 000 dividends must give the correctly rounded quotient."""
    k = np.arange(10000, dtype=np.float32)
    assert np.array_equal(O.math_apply("div10000", k).view(np.uint32), (k / np.float32(10000.0)).view(np.uint32))


def test_special_values():
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    e2 = O.math_apply("exp2", np.array([0, 1, -1, 10, -126, -149, -151, 128, 200, nan], np.float32))
    assert list(e2[:8]) == [1.0, 2.0, 0.5, 1024.0, np.float32(2.0 ** -126), np.float32(2.0 ** -149), 0.0, inf] and e2[8] == inf and np.isnan(e2[9])
    ex = O.math_apply("exp", np.array([0, -200, 100, nan], np.float32))
    assert ex[0] == 1.0 and ex[1] == 0.0 and ex[2] == inf and np.isnan(ex[3])
    lg = O.math_apply("log2", np.array([1, 2, 0.5, 8, 0, -1, inf, nan, 2.0 ** -140], np.float32))
    assert list(lg[:5]) == [0.0, 1.0, -1.0, 3.0, -inf] and np.isnan(lg[5]) and lg[6] == inf and np.isnan(lg[7]) and lg[8] == -140.0
    ac = O.math_apply("acos", np.array([1, -1, 0, 1.5, -1.5, nan], np.float32))
    assert ac[0] == 0.0 and ac[1] == np.float32(np.pi) and ac[2] == np.float32(np.pi / 2) and np.isnan(ac[3:]).all()


def test_bit_patterns_are_pinned():
    """Known answers (hex of the binary32 results) for inputs of the kind the renderer feeds in."""
    x = np.array([0.3, -2.75, 5.5], np.float32)
    got = {k: [hex(v) for v in O.math_apply(k, a).view(np.uint32)] for k, a in
           (("exp2", x), ("exp", x), ("log2", np.abs(x)), ("acos", np.array([0.3, -0.75, 0.999], np.float32)))}
    assert got == PINNED, got


PINNED = {'exp2': ['0x3f9d9624', '0x3e1837f0', '0x423504f4'], 'exp': ['0x3facc82c', '0x3d82ec9c', '0x4374b122'], 'log2': ['0xbfde54e3', '0x3fbacea7', '0x401d6754'], 'acos': ['0x3fa20fb0', '0x401ace94', '0x3d37315a']}
