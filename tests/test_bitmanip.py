"""Zba / Zbb / Zbs / Zbc / Zicond (row f1: bitmanip.cpp:10-1110, Hart.cpp:14728-14760, Decoder.cpp:2490-2557, 2596-2627, 2771-2864).

Pinned: the decoder, word for word against the reference's own Decoder (oracle/_ref) over the whole OP / OP-IMM / OP-32 / OP-IMM-32
function space of both widths.  The arithmetic lives in Hart member functions (Hart.cpp cannot be built here), so it is "parity
unpinned" against the reference itself; it is checked three ways instead: a plain Python statement of the ratified specification's
definitions, the oracle (C, follows bitmanip.cpp line by line) and the device core must agree on corner operands and on random
programs."""
import ctypes as C
import random

import pytest

import simutil as S
import whisper_amd as W
from simutil import rvoracle
from whisper_amd import workloads as WL

needs_ref = pytest.mark.skipif(rvoracle.ref() is None, reason="oracle/_ref not built")
ALL = "zba_zbb_zbs_zbc_zicond"
ZB_FN = ["shadd", "slli.uw", "andn", "orn", "xnor", "clz", "ctz", "cpop", "min", "minu", "max", "maxu", "sext.b", "sext.h", "zext.h", "rol", "ror",
         "rolw", "rorw", "orc.b", "rev8", "bclr", "bext", "binv", "bset", "czero.eqz", "czero.nez", "clmul", "clmulh", "clmulr"]


class Dec(C.Structure):
    _fields_ = [("w0", C.c_uint32), ("imm", C.c_int32), ("pad", C.c_uint32), ("raw", C.c_uint32)]


def _host_decode(flags, words):
    L = S.shim()
    L.hs_decode.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(Dec)]
    out = (Dec * len(words))()
    L.hs_decode(flags, (C.c_uint32 * len(words))(*words), len(words), out)
    return out


def _expected_name(rec, rv64):
    """The reference mnemonic an OP_ZB record stands for."""
    fn, imm_form, par = ZB_FN[(rec.w0 >> 23) & 31], bool(rec.w0 & (1 << 29)), rec.imm
    if fn == "shadd":
        return ("add.uw" if par == 8 else f"sh{par & 3}add" + (".uw" if par & 8 else ""))
    if fn in ("clz", "ctz", "cpop"):
        return fn + ("w" if par and rv64 else "")
    if fn == "rev8":
        return "rev8" if rv64 else "rev8_32"
    if fn == "zext.h":
        return "packw" if rv64 else "pack"
    if fn in ("rolw", "rorw") and not rv64:
        fn = fn[:-1]
    if imm_form and fn in ("ror", "rorw", "bclr", "bext", "binv", "bset"):
        return {"ror": "rori", "rorw": "roriw"}.get(fn, fn + "i")
    return fn


ZB_NAMES = set("""sh1add sh2add sh3add add.uw sh1add.uw sh2add.uw sh3add.uw slli.uw andn orn xnor clz ctz cpop clzw ctzw cpopw max maxu min minu
sext.b sext.h rol ror rori rolw rorw roriw orc.b rev8 rev8_32 bclr bclri bext bexti binv binvi bset bseti czero.eqz czero.nez clmul clmulh
clmulr""".split())
RV64_ONLY = set("add.uw sh1add.uw sh2add.uw sh3add.uw slli.uw clzw ctzw cpopw rolw rorw roriw rev8 packw".split())


@needs_ref
@pytest.mark.parametrize("rv64", [True, False])
def test_bitmanip_decoder_matches_reference_decoder(rv64):
    """Every funct7 x funct3 of OP / OP-32 and every 12-bit immediate x funct3 {1, 5} of OP-IMM / OP-IMM-32, plus random words:
    what decodes here as OP_ZB is that very instruction in the reference's Decoder, with the same registers and shift amount; and
    every such instruction of the reference that its execute function accepts on this width decodes here."""
    rng = random.Random(11)
    words = []
    for op in (0x13, 0x1b):
        for f3 in (1, 5):
            for imm in range(4096):
                words.append((imm << 20) | (rng.getrandbits(5) << 15) | (f3 << 12) | (rng.getrandbits(5) << 7) | op)
    for op in (0x33, 0x3b):
        for f7 in range(128):
            for f3 in range(8):
                for rs2 in (0, rng.randrange(1, 32)):
                    words.append((f7 << 25) | (rs2 << 20) | (rng.getrandbits(5) << 15) | (f3 << 12) | (rng.getrandbits(5) << 7) | op)
    words += [(rng.getrandbits(25) << 7) | rng.choice([0x13, 0x1b, 0x33, 0x3b]) for _ in range(60000)]
    flags = rvoracle.isa_flags("rv64imac_zicsr_" + ALL) | (0 if rv64 else 512)
    out = _host_decode(flags, words)
    op_zb = _host_decode(flags, [WL.zb_word("andn", 1, 2, 3)])[0].w0 & 0xff
    seen = set()
    for i, w in enumerate(words):
        _, name, ops = rvoracle.ref_decode(w, rv64=rv64)
        rec = out[i]
        op = rec.w0 & 0xff
        rd, rs1, rs2 = (w >> 7) & 31, (w >> 15) & 31, (w >> 20) & 31
        if op == op_zb:
            assert name == _expected_name(rec, rv64), f"{w:#x}: decoded as {_expected_name(rec, rv64)}, reference says {name}"
            assert ((rec.w0 >> 8) & 31, (rec.w0 >> 13) & 31) == (rd, rs1) == (ops[0], ops[1]), hex(w)
            if rec.w0 & (1 << 29):
                assert rec.imm == ops[2], f"{w:#x}: shift amount {rec.imm} vs {ops[2]}"
            elif name not in ("clz", "ctz", "cpop", "clzw", "ctzw", "cpopw", "sext.b", "sext.h", "orc.b", "rev8", "rev8_32", "pack", "packw"):
                assert (rec.w0 >> 18) & 31 == rs2 == ops[2], hex(w)
            seen.add(name)
        elif name in ZB_NAMES or name in ("pack", "packw"):
            assert op == 0, f"{w:#x}: {name} decoded as a base instruction here"
            # rejected here at decode: only what the reference's execute function rejects (bitmanip.cpp)
            amt_bad = name in ("rori", "bclri", "bexti", "binvi", "bseti") and ops[2] > (63 if rv64 else 31)          # checkShiftImmediate
            pack_bad = name in ("pack", "packw") and (rs2 != 0 or (name == "pack" and rv64))                         # Zbkb unless zext.h
            width_bad = (not rv64 and name in RV64_ONLY) or (rv64 and name == "rev8_32")
            assert amt_bad or pack_bad or width_bad, f"{w:#x}: {name} rejected at decode"
    want = (ZB_NAMES | {"packw"}) - {"rev8_32"} if rv64 else (ZB_NAMES | {"pack"}) - RV64_ONLY
    assert want <= seen, want - seen


# ---- the ratified definitions, stated on Python integers (xlen-bit unsigned values)
def _spec(name, a, b, xlen):
    m = (1 << xlen) - 1
    sx = lambda v, bits: (v & ((1 << bits) - 1)) - ((v >> (bits - 1) & 1) << bits)
    s = lambda v: sx(v, xlen)
    w = lambda v: sx(v, 32) & m
    rot = lambda v, n, bits: ((v >> (n % bits)) | (v << ((bits - n % bits) % bits))) & ((1 << bits) - 1)

    def clm(x, y):                                        # carry-less product, 2 * xlen bits
        r = 0
        for i in range(xlen):
            if (y >> i) & 1:
                r ^= x << i
        return r
    a32, ix = a & 0xffffffff, b & (xlen - 1)
    table = {
        "sh1add": lambda: (a << 1) + b, "sh2add": lambda: (a << 2) + b, "sh3add": lambda: (a << 3) + b,
        "add.uw": lambda: a32 + b, "sh1add.uw": lambda: (a32 << 1) + b, "sh2add.uw": lambda: (a32 << 2) + b, "sh3add.uw": lambda: (a32 << 3) + b,
        "slli.uw": lambda: a32 << (b & 63),
        "andn": lambda: a & ~b, "orn": lambda: a | ~b, "xnor": lambda: ~(a ^ b),
        "clz": lambda: xlen - a.bit_length(), "ctz": lambda: xlen if a == 0 else (a & -a).bit_length() - 1, "cpop": lambda: bin(a).count("1"),
        "clzw": lambda: 32 - a32.bit_length(), "ctzw": lambda: 32 if a32 == 0 else (a32 & -a32).bit_length() - 1, "cpopw": lambda: bin(a32).count("1"),
        "max": lambda: a if s(a) > s(b) else b, "maxu": lambda: max(a, b), "min": lambda: a if s(a) < s(b) else b, "minu": lambda: min(a, b),
        "sext.b": lambda: sx(a, 8), "sext.h": lambda: sx(a, 16), "zext.h": lambda: a & 0xffff, "zext.h32": lambda: a & 0xffff,
        "rol": lambda: rot(a, xlen - ix, xlen), "ror": lambda: rot(a, ix, xlen), "rori": lambda: rot(a, ix, xlen),
        "rolw": lambda: w(rot(a32, 32 - (b & 31), 32)), "rorw": lambda: w(rot(a32, b & 31, 32)), "roriw": lambda: w(rot(a32, b & 31, 32)),
        "orc.b": lambda: sum(0xff << i for i in range(0, xlen, 8) if (a >> i) & 0xff),
        "rev8": lambda: int.from_bytes(a.to_bytes(xlen // 8, "little"), "big"), "rev8_32": lambda: int.from_bytes(a.to_bytes(xlen // 8, "little"), "big"),
        "bclr": lambda: a & ~(1 << ix), "bclri": lambda: a & ~(1 << ix), "bext": lambda: (a >> ix) & 1, "bexti": lambda: (a >> ix) & 1,
        "binv": lambda: a ^ (1 << ix), "binvi": lambda: a ^ (1 << ix), "bset": lambda: a | (1 << ix), "bseti": lambda: a | (1 << ix),
        "czero.eqz": lambda: 0 if b == 0 else a, "czero.nez": lambda: 0 if b != 0 else a,
        "clmul": lambda: clm(a, b), "clmulh": lambda: clm(a, b) >> xlen, "clmulr": lambda: clm(a, b) >> (xlen - 1),
    }
    return table[name]() & m


@pytest.mark.parametrize("xlen", [64, 32])
def test_bitmanip_arithmetic_oracle_and_device_core_match_the_specification(xlen):
    """Each mnemonic on corner and random operands, one instruction at a time: the specification's value == the oracle's == the
    device core's (host build)."""
    rng = random.Random(xlen)
    m = (1 << xlen) - 1
    isa = f"rv{xlen}imac_zicsr_" + ALL
    corner = [0, 1, 2, m, m >> 1, 1 << (xlen - 1), 0x80, 0xff, 0x7f, 0x8000, 0xffff, 0x7fffffff, 0x80000000, 0xffffffff, 0x0102040810204080 & m,
              0xff00ff00ff00ff00 & m, xlen - 1, xlen, 31, 32, 33, 63, 64]
    skip32 = {"add.uw", "sh1add.uw", "sh2add.uw", "sh3add.uw", "slli.uw", "clzw", "ctzw", "cpopw", "rolw", "rorw", "roriw", "rev8", "zext.h"}
    skip64 = {"rev8_32", "zext.h32"}
    names = [n for n in sorted(WL.ZB_ENC) if n not in (skip32 if xlen == 32 else skip64)]
    base = 0x80000000
    h = rvoracle.Hart(isa=isa, mem_base=base, mem_size=1 << 12, tohost=None, reset_pc=base)
    L = S.shim()
    hs = L.hs_create3(rvoracle.isa_flags(isa) | (512 if xlen == 32 else 0), base, 1 << 12, 2**64 - 1, 0, 0, 0)
    L.hs_get_x.restype = C.c_uint64
    n_done = 0
    try:
        for name in names:
            form = WL.ZB_ENC[name][3]
            for k in range(60):
                a = rng.choice(corner) if k < 40 else rng.getrandbits(xlen)
                b = rng.choice(corner) if k % 3 else rng.getrandbits(xlen)
                if form in ("s", "w5"):
                    b = rng.randrange(32 if form == "w5" else xlen)
                word = WL.zb_word(name, 7, 5, 6 if form == "r" else b)
                want = _spec(name, a, b, xlen)
                # oracle
                h.write_mem(base, word.to_bytes(4, "little"))
                h.pc = base
                h.set_x(5, a); h.set_x(6, b); h.set_x(7, 0x5a5a)
                h.resume()
                h.run(n_done + 1)
                assert h.pc == base + 4, (name, hex(word), "trapped in the oracle")
                assert h.x(7) == want, (name, hex(a), hex(b), hex(h.x(7)), hex(want))
                n_done += 1
                # device core, host build
                assert L.hs_write_mem(hs, base, word.to_bytes(4, "little"), 4) == 0
                L.hs_set_pc(hs, base)
                L.hs_set_x(hs, 5, C.c_uint64(a)); L.hs_set_x(hs, 6, C.c_uint64(b)); L.hs_set_x(hs, 7, C.c_uint64(0x5a5a))
                L.hs_run(hs, C.c_uint64(n_done), 0)
                assert L.hs_get_x(hs, 7) == want, (name, hex(a), hex(b), hex(L.hs_get_x(hs, 7)), hex(want))
    finally:
        L.hs_destroy(hs)


@pytest.mark.parametrize("xlen,ext", [(64, ALL), (32, ALL), (64, "zba_zbb"), (32, "zbs_zicond"), (64, "zbc"), (64, "zicntr")])
def test_bitmanip_programs_oracle_equals_device_core(xlen, ext):
    """Random programs (traps for every encoding whose extension is off or whose shift amount is too large included): the oracle
    and the device core (host build) agree on the whole result record, commit digest included."""
    for seed in range(8):
        p = WL.bitmanip_program(seed, xlen=xlen, ext=ext)
        ro, _, _ = S.run_oracle(p)
        rh = S.run_hostsim(p)[0]
        assert ro["status"] == rvoracle.STOP_TOHOST, (xlen, ext, seed)
        assert rh == ro, (xlen, ext, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("xlen", [64, 32])
def test_bitmanip_programs_on_the_gpu(xlen):
    """64 distinct programs per width in one batch: result records (state, memory and commit digests) == the oracle's; and the text
    log of one of them == the oracle's commit log with the library's disassembly."""
    from test_disasm import _expected_text
    progs = [WL.bitmanip_program(s, xlen=xlen, ext=ALL if s % 4 else "zba_zbb_zbs") for s in range(64)]
    for ext in (ALL, "zba_zbb_zbs"):
        sel = [p for p in progs if p.isa.endswith(ext)]
        b = W.Batch(len(sel), isa=sel[0].isa, mem_base=sel[0].mem_base, mem_size=sel[0].mem_size)
        try:
            b.load_elfs([p.elf for p in sel])
            b.set_stop(max_inst=sel[0].max_inst)
            b.trace_ring(1 << 16)
            assert b.run(trace_hash=True) == 0
            res = b.results()
            for i, p in enumerate(sel):
                ro, _, log = S.run_oracle(p, hart_id=i, log_cap=4000)
                assert res[i].as_dict() == ro, (xlen, ext, i)
                if i == 1:
                    text = b.trace_text(i)
                    got = text.rstrip("\n").split("\n")
                    assert got == _expected_text(p, log, text)      # (every hart is hart 0 of its own one-hart system)
        finally:
            b.close()
