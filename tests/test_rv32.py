"""RV32 (System<uint32_t>, BASELINE config 0: the README tohost program built for rv32imc).

CPU: the oracle against the reference's own RV32 trace (tests/golden/appendix_d.json `rv32`), the
device decoder (host build) against the reference's Decoder in RV32 mode, the oracle's 32-bit CSR views against the
reference's CsRegs<uint32_t>, and oracle == device semantics (host build of wsp_core.h) on the README program and on random
RV32 programs.  GPU: the same programs through the C ABI (ELF32 loader, xlen = 32 batches), bit-exact against the oracle."""
import ctypes as C
import json
import os
import random
import struct

import pytest

import simutil as S
from simutil import rvoracle
from whisper_amd import workloads as WL

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "appendix_d.json")))
R = rvoracle.ref()
needs_ref = pytest.mark.skipif(R is None, reason="oracle/_ref not built")

RV64_ONLY = set("""addiw addw subw sllw srlw sraw slliw srliw sraiw mulw divw divuw remw remuw ld sd lwu lr.d sc.d amoadd.d amoand.d
amomax.d amomaxu.d amomin.d amominu.d amoor.d amoswap.d amoxor.d fcvt.l.s fcvt.lu.s fcvt.l.d fcvt.lu.d fcvt.s.l fcvt.s.lu fcvt.d.l
fcvt.d.lu fmv.x.d fmv.d.x c.addiw c.addw c.subw c.ld c.sd c.ldsp c.sdsp""".split())


def test_oracle_matches_reference_rv32_trace():
    """The reference's captured rv32imc run (README-style x = 1; y = 2; z = x + y; tohost = 1): every commit record's pc, opcode,
    change kind / key / value, the stop reason and the final pc / a0 / misa / mstatus."""
    g = GOLD["rv32"]
    code = b"".join(struct.pack("<H", p) for p in g["parcels"])
    h = rvoracle.Hart(isa=g["isa"], mem_base=0, mem_size=g["base"] + 0x10000, tohost=g["tohost"], reset_pc=g["base"])
    h.write_mem(g["base"], code)
    st, log = h.run(100, log_cap=32)
    assert st == rvoracle.STOP_TOHOST
    assert len(log) == len(g["records"])
    kinds = {rvoracle.CH_R: "r", rvoracle.CH_M: "m"}
    for c, r in zip(log, g["records"]):
        (kind, key, val), = c.changes()
        op = f"{c.inst:08x}" if c.size == 4 else f"{c.inst & 0xffff:04x}"
        assert (c.tag, c.pc, op, kinds[kind], key, val) == (r["tag"], r["pc"], r["op"], r["kind"], r["key"], r["val"]), r["line"]
    f = g["final"]
    assert (h.pc, h.x(10), h.csr(0x301), h.csr(0x300)) == (f["pc"], f["a0"], f["misa"], f["mstatus"])


class Dec(C.Structure):
    _fields_ = [("w0", C.c_uint32), ("imm", C.c_int32), ("pad", C.c_uint32), ("raw", C.c_uint32)]


def _host_decode(flags, words):
    L = S.shim()
    L.hs_decode.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(Dec)]
    n = len(words)
    out = (Dec * n)()
    L.hs_decode(flags, (C.c_uint32 * n)(*words), n, out)
    return out


@needs_ref
def test_rv32_decoder_matches_reference_decoder():
    """Device decoder with ISA_RV32 vs the reference Decoder with enableRv64(false) on every 16-bit word and 60000 random 32-bit
    words: reference-illegal is illegal here; what decodes here is a supported mnemonic there and never an RV64-only one; what
    the reference decodes but rejects at execute time on RV32 (`if (not isRv64())`, Hart.cpp / float.cpp / amo.cpp) is rejected
    here at decode."""
    from test_cpu_host import SUPPORTED
    rng = random.Random(32)
    words = [rng.getrandbits(32) | 3 for _ in range(60000)] + list(range(0, 65536))
    words = [w for w in words if w < 65536 and (w & 3) != 3 or (w & 3) == 3 and (w & 0x1c) != 0x1c]
    out = _host_decode(rvoracle.isa_flags("rv32imafdc_zicsr") | 512, words)
    ok32 = (SUPPORTED - RV64_ONLY) | {"c.jal", "c.flw", "c.fsw", "c.flwsp", "c.fswsp"}
    seen = set()
    for i, w in enumerate(words):
        _, name, _ = rvoracle.ref_decode(w, rv64=False)
        mine_ill = (out[i].w0 & 0xff) == 0
        if name == "illegal":
            assert mine_ill, f"{w:#x}: reference illegal, decoded here"
        elif not mine_ill:
            assert name in ok32, f"{w:#x}: decoded here, reference says {name}"
            seen.add(name)
        elif name in SUPPORTED or name in ok32:            # (other names belong to extensions that are off: rejected at execute time there)
            fp_rm = name.startswith("f") and ((w >> 12) & 7) in (5, 6)
            assert name in RV64_ONLY or fp_rm or name.startswith("c.") or name in ("slli", "srli", "srai") or name.startswith(("csr", "amo", "lr.", "sc.")) or \
                name in ("sret", "sfence.vma", "wfi", "mret", "ecall", "ebreak", "fence", "fence.i"), f"{w:#x}: {name} rejected at decode"
    assert {"c.jal", "c.fsw", "c.flwsp", "c.fswsp", "mulh", "amoadd.w", "fadd.d"} <= seen


@needs_ref
@pytest.mark.parametrize("rv64", [True, False])
def test_compressed_expansion_matches_reference(rv64):
    """Every 16-bit word, both widths: the 32-bit instruction the oracle executes for it == the reference's own
    Decoder::expandCompressedInst (registers and immediates included), wherever the oracle accepts the word; and the device decoder
    produces the same record for the 16-bit word as for that expansion."""
    L = rvoracle.lib()
    L.rvo_expand16.restype = C.c_uint32
    L.rvo_expand16.argtypes = [C.c_uint32, C.c_int]
    words = [w for w in range(1, 65536) if (w & 3) != 3]
    flags = rvoracle.isa_flags("rv64imafdc_zicsr") | (0 if rv64 else 512)
    d16 = _host_decode(flags, words)
    exp = [L.rvo_expand16(w, 0 if rv64 else 1) for w in words]
    d32 = _host_decode(flags, [e if e else 0 for e in exp])
    n = 0
    for i, w in enumerate(words):
        mine_ill = (d16[i].w0 & 0xff) == 0
        if exp[i] == 0:
            assert mine_ill, f"{w:#x}: oracle illegal, device decodes"
            continue
        ref = R.ref_expand16(1 if rv64 else 0, w)
        if not rv64 and (w & 0xE003) == 0xE000:
            # c.fsw: expandCompressedInst hands encodeFsw its registers the other way round than encodeSw gets them
            # (Decoder.cpp:1930-1936); execution goes through decode16, whose operands are checked here instead
            _, name, ops = rvoracle.ref_decode(w, rv64=False)
            assert name == "c.fsw" and (ops[0], ops[1], ops[2]) == ((exp[i] >> 20) & 31, (exp[i] >> 15) & 31, ((exp[i] >> 25) << 5) | ((exp[i] >> 7) & 31))
            ref = exp[i]
        assert exp[i] == ref, f"{w:#06x}: oracle expands to {exp[i]:#010x}, reference to {ref:#010x} ({rvoracle.ref_disassemble(w, rv64=rv64)})"
        if not mine_ill:                                      # same operation, registers and immediate as the expansion (size flag aside)
            assert (d16[i].w0 & ~(1 << 28), d16[i].imm) == (d32[i].w0 & ~(1 << 28), d32[i].imm), f"{w:#06x} vs {exp[i]:#010x}"
            n += 1
    assert n > 40000


CSR32 = [0x300, 0x301, 0x304, 0x305, 0x306, 0x30A, 0x310, 0x31A, 0x320, 0x340, 0x341, 0x342, 0x343, 0x344, 0x001, 0x002, 0x003,
         0xB00, 0xB02, 0xB80, 0xB82, 0xC00, 0xC01, 0xC02, 0xC80, 0xC81, 0xC82, 0xF11, 0xF12, 0xF13, 0xF14] + list(range(0x3A0, 0x3F0)) + \
        [0x323, 0x33F, 0x723, 0x73F, 0xB03, 0xB1F, 0xB83, 0xB9F, 0x747, 0x757]     # (hardwired zero)
CSR32_W = [0x300, 0x304, 0x305, 0x306, 0x30A, 0x310, 0x31A, 0x320, 0x340, 0x341, 0x342, 0x343, 0x001, 0x002, 0x003, 0x323, 0x723, 0xB03, 0xB83,
           0x747, 0x757]


@needs_ref
@pytest.mark.parametrize("isa,cfg", [("rv32imafdcu_zicsr_zicntr", (0, 1, 1, 1, 1)), ("rv32imac_zicsr", (0, 0, 0, 1, 0))])
def test_rv32_csr_views_match_reference_csregs32(isa, cfg):
    """Presence, reset values and csrrw -> csrrs read-back of the oracle's 32-bit CSR views against CsRegs<uint32_t>."""
    R.ref_csr32_new.restype = C.c_void_p
    R.ref_csr32_new.argtypes = [C.c_int] * 5
    R.ref_csr32_peek.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    R.ref_csr32_read.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    R.ref_csr32_write.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint32]
    R.ref_csr32_implemented.argtypes = [C.c_void_p, C.c_uint]
    R.ref_csr32_free.argtypes = [C.c_void_p]
    b = R.ref_csr32_new(*cfg)
    v = C.c_uint32()
    try:
        h = rvoracle.Hart(isa=isa)
        for n in CSR32:
            impl = bool(R.ref_csr32_implemented(b, n))
            try:
                o = h.csr(n)
            except KeyError:
                o = None
            assert (o is not None) == impl, f"{isa} csr {n:#x}: oracle models={o is not None} reference implemented={impl}"
            if impl:
                assert R.ref_csr32_peek(b, n, C.byref(v)) and o == v.value, f"{isa} csr {n:#x} reset value: oracle {o:#x} reference {v.value:#x}"
        rng = random.Random(0x32C5)
        plan = []
        for n in CSR32_W + [0x3A0, 0x3A1, 0x3A3, 0x3B0, 0x3B5, 0x3EF]:
            if not R.ref_csr32_implemented(b, n):
                continue
            for val in [0xFFFFFFFF, 0, 0x55555555, 0xAAAAAAAA] + [1 << i for i in range(0, 32, 3)] + [rng.getrandbits(32) for _ in range(8)]:
                if n == 0x300:
                    val &= ~(0xA | (1 << 17))            # no interrupts, no MPRV while the program runs
                if 0x3A0 <= n <= 0x3AF:
                    val &= 0x7F7F7F7F & ~0x18181818       # entries stay Off and unlocked: the program keeps running in M mode
                plan.append((n, val))
        rng.shuffle(plan)
        base, data = 0x80000000, 0x80000000 + (1 << 16)
        from whisper_amd.rvasm import Asm
        a = Asm(base, xlen=32)
        a.li(7, data)
        reads = []
        fp_csrs = (1, 2, 3)
        for n, val in plan:
            if n in fp_csrs:
                a.li(5, 0x6000); a.csrrs(0, 0x300, 5)
            a.li(5, val); a.csrrw(0, n, 5); a.csrrs(6, n, 0); a.sw(6, 0, 7); a.addi(7, 7, 4)
            reads.append((n, val))
        a.li(5, 1); a.li(6, data - 16); a.sw(5, 0, 6)
        hh = rvoracle.Hart(isa=isa, mem_base=base, mem_size=1 << 18, tohost=data - 16)
        hh.write_mem(base, a.assemble())
        st, _ = hh.run(1 << 22)
        assert st == rvoracle.STOP_TOHOST and hh.retired == hh.exec_count, (st, hex(hh.stop_value), hex(hh.pc))
        got = hh.read_mem(data, 4 * len(reads))
        for i, (n, val) in enumerate(reads):
            if n in fp_csrs:
                assert R.ref_csr32_read(b, 0x300, 3, C.byref(v)) and R.ref_csr32_write(b, 0x300, 3, v.value | 0x6000)
            assert R.ref_csr32_write(b, n, 3, val) and R.ref_csr32_read(b, n, 3, C.byref(v))
            o = int.from_bytes(got[4 * i:4 * i + 4], "little")
            if n in (0xB00, 0xB02):
                continue                                   # the counters advance between the write and the read (Hart level)
            assert o == v.value, f"{isa}: after csrrw {n:#x} <- {val:#x}: oracle {o:#x} reference {v.value:#x}"
    finally:
        R.ref_csr32_free(b)


def _progs():
    return [WL.readme_rv32_program()] + [WL.rv32_program(s) for s in range(40)] + [WL.rv32_program(s, fp=True) for s in range(40, 70)]


def test_rv32_oracle_equals_device_semantics_on_host():
    """README rv32imc program + 70 random RV32 programs: oracle == host build of the device header, with the pre-decoded image
    and decoding from RAM."""
    seen_trap = 0
    ended = 0
    for p in _progs():
        ro, h, _ = S.run_oracle(p)
        assert ro["status"] != rvoracle.STOP_UNSUPPORTED, (p.name, ro)     # (a random program may lock itself out with a PMP write: limit / illegal stop)
        ended += ro["status"] == rvoracle.STOP_TOHOST
        for pre in (True, False):
            rh, xs, fs = S.run_hostsim(p, predecode=pre)
            assert rh == ro, (p.name, pre, S.explain_diff(p, ro, h, xs, fs))
        seen_trap += ro["exec"] != ro["retired"]
    assert seen_trap > 30 and ended > 50
    p = WL.readme_rv32_program()
    ro, h, _ = S.run_oracle(p)
    assert h.x(10) == 3 and p.elf[4] == 1                  # ELF32


@pytest.mark.gpu
def test_rv32_programs_on_the_gpu():
    """Config 0 on the device: the README rv32imc ELF32 program alone and replicated, and the random RV32 programs (one hart per
    program, then mixed waves at 8 lanes per wave), results bit-exact against the oracle."""
    import whisper_amd as W
    p = WL.readme_rv32_program()
    ro, _, _ = S.run_oracle(p)
    for n in (1, 130):
        b = W.Batch(n, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size)
        try:
            b.load_elf_bytes(p.elf)
            assert b.run(trace_hash=True) == 0
            res = b.results()
            assert all(r.as_dict() == ro for r in res)
            assert b.peek(n - 1, W.RES_X, 10) == 3 and b.peek(0, W.RES_CSR, 0x301) == 0x4034112D and b.peek(0, W.RES_CSR, 0x300) == 0x1800
        finally:
            b.close()
    for fp in (False, True):
        progs = [WL.rv32_program(s, fp=fp) for s in range(96)]
        exp = [S.run_oracle(q, hart_id=i)[0] for i, q in enumerate(progs)]            # (the programs read mhartid)
        for lanes in (0, 8):
            b = W.Batch(len(progs), isa=progs[0].isa, mem_base=progs[0].mem_base, mem_size=progs[0].mem_size, lanes_per_wave=lanes)
            try:
                b.load_elfs([q.elf for q in progs])
                for i, q in enumerate(progs):
                    b.set_stop(max_inst=q.max_inst, lo=i, hi=i + 1)
                assert b.run(trace_hash=True) == 0
                res = [r.as_dict() for r in b.results()]
                for i, q in enumerate(progs):
                    assert res[i] == exp[i], (q.name, lanes)
            finally:
                b.close()
    with pytest.raises(W.WspError):
        W.Batch(1, isa="rv32imacsu_zicsr")                  # Sv32 is not modelled: refused, not guessed
