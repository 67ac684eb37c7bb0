"""The vector half of the oracle pinned against the reference's own code where that compiles here: VecRegs (VecRegs.hpp / .cpp:
group multipliers, VLMAX / elemMax, legal SEW x LMUL configurations, element index validity, tail / mask agnostic fill of
isDestActive, mask-register bits) and the floating-point helpers of float-util.hpp (doFadd .. doFsqrt, fusedMultiplyAdd, doNegate,
fpConvertTo with its 16-bit clamps), all driven through oracle/ref_glue_vec.cpp.  What stays unpinned is the instruction bodies in
vector*.cpp (Hart members: Hart.hpp needs Boost)."""
import ctypes as C
import random
import struct

import pytest

from simutil import rvoracle

R = rvoracle.ref()
pytestmark = pytest.mark.skipif(R is None or not hasattr(R, "ref_vec_new"), reason="oracle/_ref (built from /root/reference) is absent")

C_VL, C_VTYPE, C_MSTATUS, C_FCSR = 0xC20, 0xC21, 0x300, 0x003


def _setup():
    u64p = C.POINTER(C.c_uint64)
    R.ref_vec_new.restype = C.c_void_p
    R.ref_vec_new.argtypes = [C.c_uint] * 3
    R.ref_vec_free.argtypes = [C.c_void_p]
    R.ref_vec_update.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_uint]
    R.ref_vec_legal.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    R.ref_vec_valid_index.argtypes = [C.c_void_p, C.c_uint, C.c_uint64, C.c_uint, C.c_uint]
    R.ref_vec_vlmax.argtypes = [C.c_void_p]
    R.ref_vec_elem_max.argtypes = [C.c_void_p, C.c_uint]
    R.ref_vec_groupx8.argtypes = [C.c_uint]
    R.ref_vec_set_reg.argtypes = [C.c_void_p, C.c_uint, C.c_char_p]
    R.ref_vec_get_reg.argtypes = [C.c_void_p, C.c_uint, C.c_char_p]
    R.ref_vec_dest_active.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, u64p]
    R.ref_vec_write_mask.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int]
    R.ref_vec_write_elem.argtypes = [C.c_void_p, C.c_uint, C.c_uint64, C.c_uint, C.c_uint, C.c_uint64]
    R.ref_fp_arith.restype = C.c_uint64
    R.ref_fp_arith.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.POINTER(C.c_uint)]
    R.ref_fp_convert.restype = C.c_uint64
    R.ref_fp_convert.argtypes = [C.c_int, C.c_uint64, C.c_uint, C.POINTER(C.c_uint)]
    L = rvoracle.lib()
    L.rvo_vec_probe.argtypes = [C.c_void_p, C.c_uint, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u64p]
    L.rvo_vf_probe.restype = C.c_uint64
    L.rvo_vf_probe.argtypes = [C.c_uint, C.c_uint, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.POINTER(C.c_uint)]
    return L


def _probe(L, h, what, a=0, b=0, c=0, d=0, val=0):
    out = C.c_uint64(val)
    r = L.rvo_vec_probe(h.h, what, a, b, c, d, C.byref(out))
    return r, out.value


@pytest.mark.parametrize("vlenb", [16, 32, 64])
def test_vecregs_geometry_and_agnostic_fill(vlenb):
    L = _setup()
    rng = random.Random(vlenb)
    v = R.ref_vec_new(vlenb, 1, 8)
    h = rvoracle.Hart(isa="rv64imafdcv_zicsr", vlenb=vlenb)
    try:
        for gm in range(8):
            assert _probe(L, h, 6, gm)[0] == R.ref_vec_groupx8(gm)
            for sew in range(8):                                  # the reserved multiplier is caught by its group size of 0, not by the table
                if gm != 4:
                    assert _probe(L, h, 7, sew, gm)[0] == R.ref_vec_legal(v, sew, gm), (sew, gm)
        assert R.ref_vec_groupx8(4) == 0
        checked = inactive = 0
        for _ in range(300):
            sew, gm = rng.randrange(4), rng.choice([0, 1, 2, 3, 5, 6, 7])
            ta, ma = rng.randrange(2), rng.randrange(2)
            gx8 = R.ref_vec_groupx8(gm)
            vlmax = gx8 * vlenb // (8 << sew)
            if vlmax == 0:
                continue
            vl = rng.choice([0, 1, vlmax, rng.randrange(vlmax + 1)])
            R.ref_vec_update(v, sew, gm, ma, ta, 0, vl)
            h.set_csr(C_VTYPE, gm | (sew << 3) | (ta << 6) | (ma << 7)); h.set_csr(C_VL, vl)
            for reg in range(32):
                data = bytes(rng.getrandbits(8) for _ in range(vlenb))
                R.ref_vec_set_reg(v, reg, data); h.set_vreg(reg, data)
            assert _probe(L, h, 3)[0] == R.ref_vec_vlmax(v) == vlmax
            for eew in range(4):
                assert _probe(L, h, 2, eew)[0] == R.ref_vec_elem_max(v, eew)
            for _ in range(40):
                reg, ix, g, esz = rng.randrange(34), rng.choice([rng.randrange(4 * vlenb), rng.getrandbits(40)]), rng.choice([1, 2, 4, 8, 16, 32, 64]), rng.randrange(4)
                assert _probe(L, h, 1, reg, ix, g, esz)[0] == R.ref_vec_valid_index(v, reg, ix, g, 1 << esz), (reg, ix, g, esz)
            # isDestActive over the whole destination group, body / tail / masked-off elements (VecRegs.hpp:738-757)
            group = max(gx8, 8) // 8
            vd = rng.randrange(0, 32, group)
            emax = R.ref_vec_elem_max(v, sew)
            for ix in range(emax):
                for masked in (0, 1):
                    out = C.c_uint64()
                    ra = R.ref_vec_dest_active(v, vd, ix, sew, max(gx8, 8), masked, C.byref(out))
                    oa, ov = _probe(L, h, 0, vd, ix, sew, masked)
                    assert ra in (0, 1) and (oa, ov) == (ra, out.value), (sew, gm, ta, ma, vl, vd, ix, masked)
                    checked += 1; inactive += not ra
        assert checked > 10000 and inactive > 2000
        # mask-register bits and element writes land in the same bytes (VecRegs.hpp:187-211, 817-838)
        for _ in range(400):
            reg, i, val = rng.randrange(32), rng.randrange(8 * vlenb), rng.randrange(2)
            assert R.ref_vec_write_mask(v, reg, i, val) == 0
            _probe(L, h, 5, reg, i, val)
            esz = rng.randrange(4); ix = rng.randrange(vlenb >> esz); x = rng.getrandbits(64)
            assert R.ref_vec_write_elem(v, reg, ix, esz, 8, x) == 0
            _probe(L, h, 8, reg, ix, esz, 8, val=x)
        for reg in range(32):
            buf = C.create_string_buffer(vlenb)
            R.ref_vec_get_reg(v, reg, buf)
            assert buf.raw == h.vreg(reg), reg
    finally:
        R.ref_vec_free(v)


def _fp_operands(rng, dbl, n):
    ebits, mbits = (11, 52) if dbl else (8, 23)
    top = (1 << ebits) - 1
    special = [0, 1, top - 1, top, top // 2, top // 2 + 1, 1 + mbits]
    out = []
    for _ in range(n):
        e = rng.choice(special) if rng.random() < 0.5 else rng.randrange(top + 1)
        m = rng.choice([0, 1, (1 << mbits) - 1, 1 << (mbits - 1), rng.getrandbits(mbits)])
        out.append((rng.randrange(2) << (ebits + mbits)) | (e << mbits) | m)
    return out


def test_fp_helpers_match_float_util():
    """doFadd / doFsub / doFmul / doFdiv / doFsqrt / fusedMultiplyAdd / doNegate (float-util.hpp:743-931) under every rounding mode:
    results and fflags of the oracle's vector element arithmetic."""
    L = _setup()
    rng = random.Random(11)
    n = 0
    for dbl in (0, 1):
        ops = _fp_operands(rng, dbl, 1500)
        for i in range(0, len(ops) - 2, 3):
            a, b, c = ops[i], ops[i + 1], ops[i + 2]
            for rm in range(5):
                for kind, rop, args in ((0, 0, (a, b, 0)), (1, 1, (a, b, 0)), (2, 2, (a, b, 0)), (3, 3, (a, b, 0)), (4, 1, (b, a, 0)), (5, 3, (b, a, 0)),
                                        (6, 5, (a, b, c)), (7, 6, (a, 0, 0)), (8, 4, (a, 0, 0))):
                    fr, fo = C.c_uint(), C.c_uint()
                    want = R.ref_fp_arith(rop, dbl, args[0], args[1], args[2], rm, C.byref(fr))
                    oa = (a, b, c) if kind <= 6 else (a, 0, 0)
                    got = L.rvo_vf_probe(kind, 3 if dbl else 2, oa[0], oa[1], oa[2], rm, C.byref(fo))
                    assert (got, fo.value) == (want, fr.value), (dbl, kind, rm, hex(a), hex(b), hex(c))
                    n += 1
    assert n > 40000


def _vf_unary0(code, vd, vs2):
    return (0x12 << 26) | (1 << 25) | (vs2 << 20) | (code << 15) | (1 << 12) | (vd << 7) | 0x57


def test_vector_conversions_match_fp_convert_to():
    """vfwcvt / vfncvt between integers and floats, one instruction at a time on the oracle hart, element by element against
    fpConvertTo (float-util.hpp:229-520): the clamps of the 16-bit results, the rtz forms, every rounding mode, flags into fcsr."""
    _setup()
    rng = random.Random(5)
    vlenb = 64
    # (VFUNARY0 code, SEW log2, source log2, destination log2, fpConvertTo code, rtz)
    cases = [(0x11, 1, 2, 1, 0, 0), (0x10, 1, 2, 1, 1, 0), (0x17, 1, 2, 1, 0, 1), (0x16, 1, 2, 1, 1, 1),      # f32 -> i16 / u16
             (0x11, 2, 3, 2, 2, 0), (0x10, 2, 3, 2, 3, 0), (0x17, 2, 3, 2, 2, 1),                            # f64 -> i32 / u32
             (0x09, 2, 2, 3, 4, 0), (0x08, 2, 2, 3, 5, 0), (0x0f, 2, 2, 3, 4, 1), (0x0e, 2, 2, 3, 5, 1),      # f32 -> i64 / u64
             (0x0b, 1, 1, 2, 6, 0), (0x0a, 1, 1, 2, 7, 0), (0x0b, 2, 2, 3, 8, 0), (0x0a, 2, 2, 3, 9, 0),      # i16 -> f32, i32 -> f64
             (0x13, 2, 3, 2, 10, 0), (0x12, 2, 3, 2, 11, 0)]                                                  # i64 -> f32
    total = 0
    for code, sew, ssz, dsz, rcode, rtz in cases:
        for rm in range(5):
            h = rvoracle.Hart(isa="rv64imafdcv_zicsr", vlenb=vlenb)
            h.set_csr(C_MSTATUS, h.csr(C_MSTATUS) | (3 << 13) | (3 << 9))
            h.set_csr(C_FCSR, rm << 5)
            n = (vlenb >> max(ssz, dsz)) * 2                         # LMUL 1 on the narrow side, 2 on the wide side
            h.set_csr(C_VTYPE, 0 | (sew << 3) | (1 << 6)); h.set_csr(C_VL, n)
            if ssz >= 2 and rcode < 6 or rcode >= 12:
                src = _fp_operands(rng, ssz == 3, n)
                for k in range(0, n, 5):                             # values around the integer limits
                    x = rng.choice([2.0 ** 15, -2.0 ** 15 - 1, 65535.5, 2.0 ** 31, -2.0 ** 31 - 0.5, 2.0 ** 32 - 0.5, 2.0 ** 63, -2.0 ** 63, 0.5, -0.5, 1.5, 2.5]) + rng.choice([0, 0.25, -0.75])
                    src[k] = struct.unpack("<Q", struct.pack("<d", x))[0] if ssz == 3 else struct.unpack("<I", struct.pack("<f", x))[0]
            else:
                src = [rng.choice([0, 1, (1 << (8 << ssz)) - 1, 1 << ((8 << ssz) - 1), rng.getrandbits(8 << ssz)]) for _ in range(n)]
            raw = b"".join(s.to_bytes(1 << ssz, "little") for s in src)
            for k in range(len(raw) // vlenb):
                h.set_vreg(8 + k, raw[k * vlenb:(k + 1) * vlenb])
            h.write_mem(0x80000000, _vf_unary0(code, 16, 8).to_bytes(4, "little"))
            st, _ = h.run(1)
            assert h.csr(0x342) == 0, (hex(code), sew)                # no trap
            out = b"".join(h.vreg(16 + k) for k in range(2))
            flags = 0
            for i in range(n):
                fl = C.c_uint()
                want = R.ref_fp_convert(rcode, src[i], 1 if rtz else rm, C.byref(fl))
                got = int.from_bytes(out[i << dsz:(i + 1) << dsz], "little")
                assert got == want & ((1 << (8 << dsz)) - 1), (hex(code), sew, rm, i, hex(src[i]), hex(got), hex(want))
                flags |= fl.value
                total += 1
            assert h.csr(C_FCSR) & 0x1f == flags, (hex(code), sew, rm)
    assert total > 1500
