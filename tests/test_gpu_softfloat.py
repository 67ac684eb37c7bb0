"""The gfx950 build of the device softfloat (whisper_amd/csrc/wsp_softfloat.h) against the reference's own SoftFloat 3e
(oracle/_ref/libsoftfloat_ref.so through the oracle's rvo_sf_ref), results AND exception flags, bit-exact.

The host build of the same header is swept by oracle/_ref/sf_check (60 M cases), but there div128by64, mul64x64 and clz64 take
their __int128 / builtin branches; the schoolbook division and __umul64hi products that run on the GPU are only reached here.
Operands: the directed special values of oracle/sf_check.cpp crossed with each other, plus seeded random operands with the same
exponent-class mix (near-cancelling fused multiply-adds, equal-magnitude pairs, subnormals, huge / tiny exponents)."""
import ctypes as C
import random

import pytest

import simutil as S
from simutil import rvoracle
import whisper_amd as W

pytestmark = pytest.mark.gpu

S64 = [0, 1, 2, 0x000fffffffffffff, 0x0010000000000000, 0x0010000000000001, 0x001fffffffffffff, 0x3ff0000000000000, 0x3ff0000000000001,
       0x3fefffffffffffff, 0x4000000000000000, 0x3fe0000000000000, 0x7fefffffffffffff, 0x7fe0000000000000, 0x7ff0000000000000,
       0x7ff8000000000000, 0x7ff0000000000001, 0x7ff4000000000000, 0x7fffffffffffffff, 0x4340000000000000, 0x433fffffffffffff,
       0x43e0000000000000, 0x43dfffffffffffff, 0x43f0000000000000, 0x41dfffffffc00000, 0x41e0000000000000, 0x41efffffffe00000,
       0x41f0000000000000, 0x3fdfffffffffffff, 0x3fe0000000000001, 0x3ff8000000000000, 0x4004000000000000, 0x36a0000000000000,
       0x3690000000000000, 0x380fffffffffffff, 0x3810000000000000, 0x47efffffe0000000, 0x47effffff0000000]
S64 += [v | (1 << 63) for v in S64]
S32 = [0, 1, 2, 0x007fffff, 0x00800000, 0x00800001, 0x00ffffff, 0x3f800000, 0x3f800001, 0x3f7fffff, 0x40000000, 0x3f000000, 0x7f7fffff,
       0x7f000000, 0x7f800000, 0x7fc00000, 0x7f800001, 0x7fa00000, 0x7fffffff, 0x4b000000, 0x4affffff, 0x4f000000, 0x4effffff, 0x4f800000,
       0x5f000000, 0x5effffff, 0x5f800000, 0x3effffff, 0x3f000001, 0x3fc00000, 0x40200000, 0x33800000, 0x34000000, 0x0c000000]
S32 += [v | (1 << 31) for v in S32]
M64 = (1 << 64) - 1


def _rand64(rng):
    r = rng.getrandbits(64); k = rng.randrange(8); man = r & 0x800fffffffffffff
    if k == 0: return r
    if k == 1: return man | ((1023 + rng.randrange(120) - 60) << 52)
    if k == 2: return man | (rng.randrange(4) << 52)
    if k == 3: return man | ((2043 + rng.randrange(4)) << 52)
    if k == 4: return (r & 0xfff0000000000000) | (0 if rng.randrange(2) else 1 << rng.randrange(52))
    if k == 5: return man | ((1023 + rng.randrange(8) - 4) << 52)
    if k == 6: return man | ((1 + rng.randrange(70)) << 52)
    return man | ((1023 + 40 + rng.randrange(30)) << 52)


def _rand32(rng):
    r = rng.getrandbits(32); k = rng.randrange(8); man = r & 0x807fffff
    if k == 0: return r
    if k == 1: return man | ((127 + rng.randrange(60) - 30) << 23)
    if k == 2: return man | (rng.randrange(4) << 23)
    if k == 3: return man | ((251 + rng.randrange(4)) << 23)
    if k == 4: return (r & 0xff800000) | (0 if rng.randrange(2) else 1 << rng.randrange(23))
    if k == 5: return man | ((127 + rng.randrange(8) - 4) << 23)
    if k == 6: return man | ((1 + rng.randrange(40)) << 23)
    return man | ((127 + 20 + rng.randrange(45)) << 23)


def _ref(op, rm, a, b, c):
    L = rvoracle.lib()
    L.rvo_sf_ref.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t,
                             C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    n = len(a)
    A = (C.c_uint64 * n)(*a); B = (C.c_uint64 * n)(*b); Cc = (C.c_uint64 * n)(*c)
    out = (C.c_uint64 * n)(); fl = (C.c_uint32 * n)()
    assert L.rvo_sf_ref(op, rm, A, B, Cc, n, out, fl) == 0
    return list(out), list(fl)


def _operands(dbl, n_rand, seed):
    rng = random.Random(seed)
    sp, rnd = (S64, _rand64) if dbl else (S32, _rand32)
    a, b, c = [], [], []
    for x in sp:
        for y in sp:
            a.append(x); b.append(y); c.append(sp[rng.randrange(len(sp))])
    sign = 1 << (63 if dbl else 31)
    for _ in range(n_rand):
        x, y, z = rnd(rng), rnd(rng), rnd(rng)
        if rng.randrange(8) == 0:
            y = (x ^ (sign if rng.randrange(2) else 0) ^ rng.randrange(4)) & (M64 if dbl else 0xffffffff)
        a.append(x); b.append(y); c.append(z)
    return a, b, c


@pytest.mark.parametrize("dbl", [True, False])
def test_device_softfloat_matches_reference_library(dbl):
    if not rvoracle.has_softfloat():
        pytest.skip("oracle/_ref/libsoftfloat_ref.so not built")
    a, b, c = _operands(dbl, 60000, 7 if dbl else 8)
    # near-cancelling fused multiply-adds: c = -(a * b) +- a few ulps, product from the reference library
    prod, _ = _ref((0x100 if dbl else 0) | 2, 0, a, b, c)
    rng = random.Random(99)
    sign = 1 << (63 if dbl else 31); mask = M64 if dbl else 0xffffffff
    cfma = [((p ^ sign) + rng.randrange(5) - 2) & mask if i % 3 == 0 else z for i, (p, z) in enumerate(zip(prod, c))]
    ints = [rng.getrandbits(64) >> rng.randrange(64) if i % 2 else (-(rng.getrandbits(63) >> rng.randrange(63))) & M64 for i in range(len(a))]
    bad = []
    total = 0
    for op in range(15):
        cc = cfma if op == 5 else ints if 10 <= op <= 13 else c
        code = (0x100 if dbl else 0) | op
        for rm in range(5):
            if op == 14 and not dbl and rm:       # f32 -> f64 is exact: one mode
                continue
            got, gfl = W.sf_eval(code, rm, a, b, cc)
            exp, efl = _ref(code, rm, a, b, cc)
            total += len(a)
            for i in range(len(a)):
                if got[i] != exp[i] or gfl[i] != efl[i]:
                    bad.append((op, rm, hex(a[i]), hex(b[i]), hex(cc[i]), hex(got[i]), gfl[i], hex(exp[i]), efl[i]))
    assert not bad, f"{len(bad)} of {total} cases differ; first: {bad[:5]}"
    print(f"{'f64' if dbl else 'f32'}: {total} device softfloat cases bit-exact against the reference library")


def _rand64_ordinary(rng):
    """Operands the way programs have them: moderate exponents, often few significant bits."""
    r = rng.getrandbits(64); e = (1023 + rng.randrange(40) - 20) << 52; m = r & 0xfffffffffffff
    k = rng.randrange(4)
    if k == 0: m &= ~((1 << rng.randrange(53)) - 1)
    elif k == 1: m = 0
    return (r & (1 << 63)) | e | m


def test_device_fp_unit_paths_match_reference_library():
    """The step kernel's double-precision add / sub / mul / div / fused multiply-add / integer -> double under round-to-nearest-even
    (hw_* in wsp_softfloat.h: the FP unit's result plus an error-free transformation for the inexact flag, the integer routine
    outside their domain), gfx950 build, against the reference's SoftFloat: results and flags, and the share the FP unit took."""
    if not rvoracle.has_softfloat():
        pytest.skip("oracle/_ref/libsoftfloat_ref.so not built")
    a, b, c = _operands(True, 60000, 17)
    rng = random.Random(5)
    for _ in range(120000):
        x, y, z = _rand64_ordinary(rng), _rand64_ordinary(rng), _rand64_ordinary(rng)
        a.append(x); b.append(y); c.append(z)
    prod, _ = _ref(0x100 | 2, 0, a, b, c)
    sign = 1 << 63
    # exact quotients (a := a * b when that product was exact-ish) and cancelling fused multiply-adds
    cfma = [((p ^ sign) + rng.randrange(5) - 2) & M64 if i % 3 == 0 else z for i, (p, z) in enumerate(zip(prod, c))]
    adiv = [p if i % 2 else x for i, (p, x) in enumerate(zip(prod, a))]
    ints = [rng.getrandbits(64) >> rng.randrange(64) if i % 2 else (-(rng.getrandbits(63) >> rng.randrange(63))) & M64 for i in range(len(a))]
    bad, total, hw = [], 0, 0
    for op in (0, 1, 2, 3, 5, 10, 11, 12, 13):
        aa = adiv if op == 3 else a
        cc = cfma if op == 5 else ints if op >= 10 else c
        got, gfl = W.sf_eval(0x300 | op, 0, aa, b, cc)
        exp, efl = _ref(0x100 | op, 0, aa, b, cc)
        total += len(a)
        for i in range(len(a)):
            hw += gfl[i] >> 31
            if got[i] != exp[i] or (gfl[i] & 0x7fffffff) != efl[i]:
                bad.append((op, hex(aa[i]), hex(b[i]), hex(cc[i]), hex(got[i]), hex(gfl[i]), hex(exp[i]), efl[i]))
    assert not bad, f"{len(bad)} of {total} cases differ; first: {bad[:5]}"
    assert hw > total // 2, (hw, total)
    print(f"{total} cases bit-exact, {hw} on the FP unit")


def _rand32_ordinary(rng):
    r = rng.getrandbits(32); e = (127 + rng.randrange(40) - 20) << 23; m = r & 0x7fffff
    k = rng.randrange(4)
    if k == 0: m &= ~((1 << rng.randrange(24)) - 1)
    elif k == 1: m = 0
    return (r & (1 << 31)) | e | m


def test_device_fp_unit_binary32_paths_match_reference_library():
    """The binary32 add / sub / mul / div / fused multiply-add / int32 -> float paths of the FP unit (hw_*32 in wsp_softfloat.h: computed in
    binary64 and rounded once more, which is innocuous for the four basic operations; the fma ties through the unit's own binary32
    fma) that the scalar F instructions and the vector unit's SEW 32 loops use under round-to-nearest-even: gfx950 build against
    the reference's SoftFloat, results and flags, and the share the FP unit took."""
    if not rvoracle.has_softfloat():
        pytest.skip("oracle/_ref/libsoftfloat_ref.so not built")
    a, b, c = _operands(False, 60000, 27)
    rng = random.Random(15)
    for _ in range(120000):
        a.append(_rand32_ordinary(rng)); b.append(_rand32_ordinary(rng)); c.append(_rand32_ordinary(rng))
    prod, _ = _ref(2, 0, a, b, c)
    sign = 1 << 31
    cfma = [((p ^ sign) + rng.randrange(5) - 2) & 0xffffffff if i % 3 == 0 else z for i, (p, z) in enumerate(zip(prod, c))]
    adiv = [p if i % 2 else x for i, (p, x) in enumerate(zip(prod, a))]
    ints = [rng.getrandbits(32) >> rng.randrange(32) if i % 2 else (-(rng.getrandbits(31) >> rng.randrange(31))) & 0xffffffff for i in range(len(a))]
    bad, total, hw = [], 0, 0
    for op in (0, 1, 2, 3, 5, 10, 11):
        aa = adiv if op == 3 else a
        cc = cfma if op == 5 else ints if op >= 10 else c
        got, gfl = W.sf_eval(0x200 | op, 0, aa, b, cc)
        exp, efl = _ref(op, 0, aa, b, cc)
        total += len(a)
        for i in range(len(a)):
            hw += gfl[i] >> 31
            if got[i] != exp[i] or (gfl[i] & 0x7fffffff) != efl[i]:
                bad.append((op, hex(aa[i]), hex(b[i]), hex(cc[i]), hex(got[i]), hex(gfl[i]), hex(exp[i]), efl[i]))
    assert not bad, f"{len(bad)} of {total} cases differ; first: {bad[:5]}"
    assert hw > total // 2, (hw, total)
    print(f"{total} cases bit-exact, {hw} on the FP unit")
