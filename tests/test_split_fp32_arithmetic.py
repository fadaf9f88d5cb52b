"""CPU check of the arithmetic claims behind the split-fp32 ("bf16x6") kernels (csrc/ppo_x6.h, ppo_policy_bwd_x6.hip,
ppo_policy_fwd_x6.hip) -- no GPU: numpy restates the piece split and the six-term product.
  * an fp32 number is the EXACT sum of three bfloat16 pieces (each the RNE rounding of what the previous ones left);
  * a product of two bf16 numbers is exact in fp32;
  * the six piece products kept (h h, h m, m h, h l, m m, l h) miss the exact product by at most ~2^-23 |a b|;
  * the k-slot order the weight pieces are packed in (acc_kslot, ppo_optim.hip) is a bijection of a 32-wide tile and is the
    register order of a packed 32x32 accumulator tile."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(a):
    h = bf16_rne(a)
    r1 = (a - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    l = bf16_rne(r2)
    return h, m, l, r1, r2


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    x = (rng.normal(size=n) * np.exp(rng.uniform(-20, 20, size=n))).astype(np.float32)
    # (numbers whose bf16 rounding overflows, |a| >= 3.39e38, are outside what a weight, an activation or a gradient can be)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, -3.0e38, 1.1754944e-38, 127.0, -128.0,
                     0.01, 2.0 ** -100, 1.9999999], np.float32)
    return np.concatenate([x, edge])


def test_three_bf16_pieces_sum_to_the_fp32_number_exactly():
    a = _samples(200000, 1)
    h, m, l, r1, r2 = split3(a)
    # the two subtractions are exact in fp32 (checked in float64) ...
    assert np.array_equal(r1.astype(np.float64), a.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
    # ... and the third piece takes what is left: the sum is the number (float64 sum of three floats is exact here)
    s = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.array_equal(s.astype(np.float32), a)
    big = np.abs(a) > 1e-30                               # (pieces of numbers near the bottom of the range go subnormal)
    assert np.all(np.abs(s[big] - a[big].astype(np.float64)) <= 2.0 ** -25 * np.abs(a[big].astype(np.float64)))
    # piece magnitudes: |m| <= 2^-8 |a|, |l| <= 2^-16 |a| (one more bit each from round-to-nearest)
    assert np.all(np.abs(m[big]) <= 2.0 ** -8 * np.abs(a[big]) * (1 + 2.0 ** -7))
    assert np.all(np.abs(l[big]) <= 2.0 ** -16 * np.abs(a[big]) * (1 + 2.0 ** -6))


def test_products_of_pieces_are_exact_in_fp32_and_six_terms_reach_fp32_accuracy():
    rng = np.random.default_rng(2)
    a = (rng.normal(size=100000) * 3).astype(np.float32)
    b = (rng.normal(size=100000) * 0.3).astype(np.float32)
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    for x, y in ((ah, bh), (ah, bm), (am, bh), (ah, bl), (am, bm), (al, bh)):
        p64 = x.astype(np.float64) * y.astype(np.float64)
        assert np.array_equal((x * y).astype(np.float64), p64), "8-bit x 8-bit significands: the fp32 product is exact"
    six = (ah.astype(np.float64) * bh + (ah.astype(np.float64) * bm + am.astype(np.float64) * bh)
           + (ah.astype(np.float64) * bl + am.astype(np.float64) * bm + al.astype(np.float64) * bh))
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -23, rel.max()            # the dropped terms (m l, l m, l l): of the order of one fp32 rounding
    assert np.median(rel) <= 2.0 ** -25
    # with an operand that is exact in bf16 (the int8 state rows) three products are the whole product
    x8 = rng.integers(-128, 128, size=a.size).astype(np.float32)
    assert np.array_equal(bf16_rne(x8), x8)
    three = ah.astype(np.float64) * x8 + am.astype(np.float64) * x8 + al.astype(np.float64) * x8
    assert np.all(np.abs(three - a.astype(np.float64) * x8) <= 2.0 ** -25 * np.abs(a.astype(np.float64) * x8) + 1e-300)


def _acc_kslot(kk):
    """ppo_optim.hip: k-step s, lane half hh and element jj with 16 s + 8 (jj >> 2) + 4 hh + (jj & 3) == kk"""
    q = kk & 15
    return kk >> 4, (q >> 2) & 1, 4 * (q >> 3) + (q & 3)


def test_packed_accumulator_order_is_the_kslot_order_of_the_weight_pieces():
    seen = set()
    for kk in range(32):
        s, hh, jj = _acc_kslot(kk)
        assert 16 * s + 8 * (jj >> 2) + 4 * hh + (jj & 3) == kk
        seen.add((s, hh, jj))
    assert len(seen) == 32 and all(0 <= s < 2 and 0 <= hh < 2 and 0 <= jj < 8 for s, hh, jj in seen)
    # register r of lane half hh of a 32x32 accumulator tile holds row (r & 3) + 8 (r >> 2) + 4 hh (ppo_device.h dfeat); packing
    # registers 8 s .. 8 s + 7 pairwise gives element jj = r - 8 s of k-step s: the same map
    for hh in range(2):
        for r in range(16):
            row = (r & 3) + 8 * (r >> 2) + 4 * hh
            assert _acc_kslot(row) == (r >> 3, hh, r & 7)
    # X image of the backward: tile row R sits in k-slot R with bits 2 and 3 swapped -- the row order of those registers
    for R in range(32):
        slot = (R & 19) | ((R & 4) << 1) | ((R & 8) >> 1)
        s, hh, e = slot >> 4, (slot >> 3) & 1, slot & 7
        assert 16 * s + 8 * (e >> 2) + 4 * hh + (e & 3) == R
