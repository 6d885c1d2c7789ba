"""The error bound of the block-scaled two-piece fp16 arithmetic of the MLP GEMMs, as a test (CPU).

csrc/gemm_split.hip's RQHIP_SPLIT_F16X2 path (the product: ONE block per row -- the whole row) and round 5's image experiment
(tools/experiments/gemm_img_r05.hip: one block per 256-column segment, accumulators rescaled at block boundaries) replace an fp32
product sum  sum_k a_k b_k  by

    sum_k (ah_k bh_k + ah_k bm_k + am_k bh_k) 2^(Ea(k) + Eb),     a_k 2^-Ea = ah_k + am_k + ra_k   (fp16 pieces, ra the remainder),

accumulated in fp32 by v_mfma_f32_32x32x16_f16, where Ea is the exponent of the (row, column-segment) block a_k lies in -- the block's
largest |value| times 2^-Ea lies in [2^14, 2^15) -- and Eb the exponent of the weight row.  This file

  1. restates the split exactly (numpy float16 = round to nearest even with subnormals, what v_cvt_f16_f32 does) and proves, by
     running over EVERY fp32 mantissa at every position below the block maximum, the representation bound
         |v - (h + m) 2^E|  <=  max(2^-23 |v|, 2^-25 2^E)                                                        (R)
     i.e. 2^-23 relative for every entry within 2^-16 of its block's maximum (where the low piece is a normal fp16 number -- the reason
     the block maximum is put at the TOP of fp16's range), an absolute floor of 2^-40 of the block's scale below that;
  2. derives from (R) the bound of a product sum,
         |sum a b - computed|  <=  sum_k [ (c_r + c_d + c_acc) |a_k| |b_k| ]  +  floors,     c_r = 2^-22 + 2^-45, c_d = 2^-22 (1 + 2^-9),
     with c_acc = n_terms * 2^-23 under assumption (H) of tests/test_filter_bound.py (every addition inside the matrix-instruction chain
     has relative error <= 2^-23), and floors = sum_k (fa_k |b_k| + fb |a_k| + fa_k fb), fa_k = 2^-25 2^Ea(k), fb = 2^-25 2^Eb;
  3. emulates the kernel's arithmetic on operands built to sit on the worst case (every mantissa on the coherent worst case of the
     11 + 11-bit split, all products of one sign), on blocks whose scales differ by decades inside one row (the accumulator rescale at
     block boundaries is exact: a power of two), and on random operands over many decades (hypothesis), and asserts the bound --
     reporting how much of it the worst case uses.  With ONE exponent per row instead of one per block the same statements hold with
     Ea constant: the round-4 arithmetic is the special case.
"""
import numpy as np
from hypothesis import given, settings, strategies as st

TOP = 14   # the block maximum is scaled into [2^TOP, 2^(TOP+1))


def exp_of_max(mx):
    """exponent E of a block whose largest |value| is mx > 0 (finite): mx 2^-E in [2^14, 2^15)  (csrc/gemm_split.hip:gs_exp_of_bits)"""
    m, e = np.frexp(np.float32(mx))          # mx = m 2^e, m in [0.5, 1)
    return int(e) - 1 - TOP


def split(v, E):
    """(h, m) as float64 arrays: h = RN16(v 2^-E), m = RN16(v 2^-E - h)  (csrc/gemm_split.hip:gs_split2_f16; v 2^-E is exact in fp32)"""
    a = np.ldexp(np.asarray(v, np.float32), -E).astype(np.float32)
    h = a.astype(np.float16)
    m = (a - h.astype(np.float32)).astype(np.float16)          # a - h is exact in fp32
    return h.astype(np.float64), m.astype(np.float64)


def represented(v, E):
    h, m = split(v, E)
    return np.ldexp(h + m, E)


# ------------------------------------------------------------------------------------------------------------------
# 1. the representation bound (R), over every mantissa at every distance below the block maximum
# ------------------------------------------------------------------------------------------------------------------
def test_representation_bound_every_mantissa_every_position():
    mant = np.arange(1 << 23, dtype=np.uint32)
    worst_rel, worst_floor = 0.0, 0.0
    E = 0                                            # block maximum in [2^14, 2^15): exponent 0
    for below in list(range(0, 30)) + [34, 40, 48]:  # the entry's binade lies `below` binades under the block maximum's
        bits = (np.uint32(127 + TOP - below) << np.uint32(23)) | mant
        v = bits.view(np.float32)
        err = np.abs(v.astype(np.float64) - represented(v, E))
        bound = np.maximum(2.0 ** -23 * np.abs(v.astype(np.float64)), 2.0 ** -25)
        assert (err <= bound).all(), (below, float((err / bound).max()))
        worst_rel = max(worst_rel, float((err / np.abs(v.astype(np.float64))).max()) if below <= 16 else 0.0)
        worst_floor = max(worst_floor, float(err.max()) if below > 16 else 0.0)
    # entries within 2^-16 of the block maximum are represented to 2^-23 relative; the floor below that is 2^-25 (2^-40 of the scale)
    assert worst_rel <= 2.0 ** -23 and worst_floor <= 2.0 ** -25
    print(f"worst relative representation error within 2^-16 of the block maximum: 2^{np.log2(worst_rel):.2f}; floor 2^{np.log2(worst_floor):.2f}")
    # with the maximum in [1, 2) instead (round 4's first form) entries a quarter of the maximum already lose bits: the reason for TOP
    bits = (np.uint32(127 - 3) << np.uint32(23)) | mant
    v = bits.view(np.float32)
    a = v.astype(np.float32)
    h = a.astype(np.float16)
    m = (a - h.astype(np.float32)).astype(np.float16)
    err = np.abs(v.astype(np.float64) - (h.astype(np.float64) + m.astype(np.float64)))
    assert (err / np.abs(v.astype(np.float64))).max() > 1.9 * 2.0 ** -23      # one bit gone at a quarter of the maximum, more below


def test_exponent_puts_the_block_maximum_at_the_top_of_fp16():
    rng = np.random.default_rng(0)
    for mx in np.concatenate([rng.lognormal(0, 20, 2000).astype(np.float32), np.float32([1.0, 2.0 ** -126, 3.4e38, 65504.0, 2.0 ** 14])]):
        if not np.isfinite(mx) or mx == 0:
            continue
        E = exp_of_max(mx)
        s = np.ldexp(np.float64(mx), -E)
        assert 2.0 ** TOP <= s < 2.0 ** (TOP + 1), (mx, E, s)
        assert np.isfinite(np.float16(s))            # never overflows fp16 (largest finite 65504 > 2^15)


# ------------------------------------------------------------------------------------------------------------------
# 2. + 3. the product-sum bound, on emulated kernel arithmetic
# ------------------------------------------------------------------------------------------------------------------
C_R = 2.0 ** -22 + 2.0 ** -45           # |a b - a^ b^| <= |a| db + |b| da + da db with da <= 2^-23 |a|, db <= 2^-23 |b| (relative parts)
C_D = 2.0 ** -22 * (1 + 2.0 ** -9)      # the dropped am bm: |am| <= 2^-11 |a| (1 + 2^-11) each
H_U = 2.0 ** -23                        # assumption (H): relative error of one accumulation step on the matrix cores


def emulate(a, b, seg):
    """The kernel's product sum for one output: a [R] (blocks of `seg` columns, one exponent each), b [R] (one exponent for the row).
    Products of two fp16 values are exact in fp32; the accumulation is modelled in float64 (its rounding is bounded analytically by
    assumption (H)); the accumulator is rescaled by exact powers of two at block boundaries.  Returns (value, floors)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    Eb = exp_of_max(np.abs(b).max()) if np.abs(b).max() > 0 else 0
    bh, bm = split(b, Eb)
    total, floors = 0.0, 0.0
    fb = 2.0 ** (Eb - 25)
    for s0 in range(0, a.size, seg):
        blk = a[s0:s0 + seg]
        if np.abs(blk).max() == 0:
            continue
        Ea = exp_of_max(np.abs(blk).max())
        ah, am = split(blk, Ea)
        acc = (ah * bh[s0:s0 + seg] + ah * bm[s0:s0 + seg] + am * bh[s0:s0 + seg]).sum()
        total += np.ldexp(acc, Ea + Eb)
        fa = 2.0 ** (Ea - 25)
        floors += (fa * np.abs(b[s0:s0 + seg].astype(np.float64)) + fb * np.abs(blk.astype(np.float64)) + fa * fb).sum()
    return total, floors


def check(a, b, seg):
    got, floors = emulate(a, b, seg)
    exact = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
    mag = float(np.dot(np.abs(a.astype(np.float64)), np.abs(b.astype(np.float64))))
    bound_repr = (C_R + C_D) * mag + floors
    err = abs(got - exact)
    assert err <= bound_repr * (1 + 1e-12) + 1e-300, (err, bound_repr)
    return err, mag, floors


def worst_mantissa(n, rng, positive=False):
    """every value on the coherent worst case of the 11 + 11-bit split (tests/test_gpu_gemm_split.py:worst_mantissa)"""
    a = rng.integers(0, 1024, n).astype(np.float64)
    j = rng.integers(0, 256, n).astype(np.float64)
    sgn = np.ones(n) if positive else rng.integers(0, 2, n) * 2.0 - 1
    v = sgn * (1 + a * 2.0 ** -10 + 2.0 ** -12 + (4 * j + 1) * 2.0 ** -23)
    out = v.astype(np.float32)
    assert (out.astype(np.float64) == v).all()
    return out


def test_product_sum_bound_on_the_worst_case_and_what_it_uses():
    rng = np.random.default_rng(1)
    used = 0.0
    for R, seg in ((768, 256), (512, 256), (256, 128), (2048, 256)):
        for _ in range(20):
            a, b = worst_mantissa(R, rng, positive=True), worst_mantissa(R, rng, positive=True) * np.float32(2.0 ** -5)
            err, mag, floors = check(a, b, seg)
            used = max(used, err / ((C_R + C_D) * mag))
            assert floors <= 2.0 ** -36 * mag                      # no entry is far below its block maximum here: the floors are nothing
            a, b = worst_mantissa(R, rng), worst_mantissa(R, rng)
            check(a, b, seg)
    print(f"the coherent worst case uses {used:.2f} of the representation + dropped-term bound (c_r + c_d = 2^{np.log2(C_R + C_D):.2f})")
    assert 0.2 < used <= 1.0
    # with assumption (H) the accumulation of 3 R products adds at most 3 R 2^-23 of the sum of magnitudes; together, for R = 768:
    total_c = C_R + C_D + 3 * 768 * H_U
    print(f"total constant for R = 768: |err| <= {total_c / 2.0 ** -22:.1f} x 2^-22 sum|a||b| (worst case; the measured GPU error on "
          "these operands is tests/test_gpu_gemm_split.py's 'worst-case mantissas' family: below the library fp32 GEMM's)")


def test_blocks_of_very_different_scale_inside_one_row():
    """Per-block exponents: every block keeps its own 22 bits (with ONE exponent for the row the small blocks would sit on the floor)."""
    rng = np.random.default_rng(2)
    for _ in range(50):
        a = rng.standard_normal(768).astype(np.float32)
        a[:256] *= np.float32(10.0 ** rng.integers(-9, 10))
        a[256:512] *= np.float32(10.0 ** rng.integers(-9, 10))
        a[512:] *= np.float32(10.0 ** rng.integers(-9, 10))
        b = (rng.standard_normal(768) / 28).astype(np.float32)
        err, mag, floors = check(a, b, 256)
        assert floors <= 2.0 ** -30 * mag + 1e-300
        # the same row under ONE exponent (segment = the whole row): still inside ITS bound, but the bound's floor term is no longer small
        _, mag1, floors1 = check(a, b, 768)
        assert floors1 >= floors


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.sampled_from([(768, 256), (512, 256), (256, 128), (128, 128)]), st.integers(-30, 30), st.integers(0, 6))
def test_product_sum_bound_random_operands(seed, shape, scale, inner_decades):
    R, seg = shape
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal(R) * 10.0 ** scale * 10.0 ** (-rng.integers(0, inner_decades + 1, R))).astype(np.float32)
    b = (rng.standard_normal(R) * 10.0 ** (-rng.integers(0, inner_decades + 1, R))).astype(np.float32)
    if not (np.isfinite(a).all() and np.isfinite(b).all()) or np.abs(b).max() == 0:
        return
    check(a, b, seg)
