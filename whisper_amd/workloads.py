"""Seeded synthetic workloads for the BASELINE.json configs.

No RISC-V toolchain or riscv-arch-test checkout exists offline, so every program is generated
here with rvasm.Asm and wrapped into an ELF (elfio.write_elf).  The same bytes are fed to the
oracle and to the device.
  readme_program()      config 0: README-style tohost program (compressed forms)
  archtest_suite()      config 1: arch-test style I/M/C signature tests, one program per mnemonic
  compute_kernel()      config 2: CoreMark-style RV64IMAFD kernel, per-instance seed in a0
  fuzz_program()        random legal/illegal instruction mix with a skipping trap handler (M-mode)
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field

from .elfio import write_elf32, PF_R, PF_W, PF_X, write_elf
from .rvasm import CSR, Asm

MASK64 = (1 << 64) - 1
BASE = 0x80000000


@dataclass
class Program:
    name: str
    isa: str
    elf: bytes
    entry: int
    segments: list          # [(paddr, bytes, flags)]
    symbols: dict
    mem_base: int = BASE
    mem_size: int = 1 << 16
    max_inst: int = 1 << 22
    meta: dict = field(default_factory=dict)

    @property
    def tohost(self):
        return self.symbols.get("tohost")


def _finish(a: Asm, name, isa, data: Asm | None = None, mem_size=1 << 16, max_inst=1 << 22, extra_syms=None, meta=None):
    text = a.assemble()
    segs = [(a.base, text, PF_R | PF_X)]
    syms = dict(a.symbols)
    if data is not None:
        segs.append((data.base, data.assemble(), PF_R | PF_W))
        syms.update(data.symbols)
    if extra_syms:
        syms.update(extra_syms)
    entry = syms.get("_start", a.base)
    elf = (write_elf32 if isa.lower().startswith("rv32") else write_elf)(segs, entry, syms)
    return Program(name, isa, elf, entry, segs, syms, BASE, mem_size, max_inst, meta or {})


def _exit_tohost(a: Asm, tohost_addr, value=1, tmp1=5, tmp2=6):
    a.li(tmp1, value)
    a.li(tmp2, tohost_addr)
    a.sd(tmp1, 0, tmp2)
    a.jal(0, a.pc)   # j . (never reached after the stop)


# ------------------------------------------------------------------------------------------ config 0
def readme_program() -> Program:
    """README.md:167-185 style: x=1; y=2; z=x+y; write 1 to tohost (RV64 rendition, c.* forms)."""
    data_base = BASE + 0x1000
    a = Asm(BASE)
    a.label("_start")
    a.c_li(10, 1)
    a.c_li(11, 2)
    a.c_add(10, 11)
    a.li(5, data_base)
    a.c_li(6, 1)
    a.sd(6, 0, 5)
    a.jal(0, a.pc)
    d = Asm(data_base)
    d.label("tohost")
    d.dword(0)
    d.label("fromhost")
    d.dword(0)
    return _finish(a, "readme", "rv64imc_zicsr", d, mem_size=1 << 14, max_inst=1000)


# ------------------------------------------------------------------------------------------ config 1
_CORNERS = [0, 1, 2, MASK64, MASK64 - 1, 1 << 63, (1 << 63) - 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0x100000000,
            0x5555555555555555, 0xAAAAAAAAAAAAAAAA, 0x0123456789ABCDEF, 63, 64, 31, 32]


def _fnv(s: str) -> int:
    h = 0xCBF29CE484222325
    for ch in s.encode():
        h = ((h ^ ch) * 0x100000001B3) & MASK64
    return h


def _archtest_operands(mn: str, n_rand=6, part=0):
    rng = random.Random(_fnv(mn) ^ (part * 0x9E3779B97F4A7C15 & MASK64))
    cs = _CORNERS[5 * part % len(_CORNERS):] + _CORNERS[:5 * part % len(_CORNERS)]      # later parts: the other corner values first
    ops = [(x, y) for x in cs[:10] for y in cs[:10]]
    for i in range(64):
        ops.append((1 << i, rng.getrandbits(64)))
    for _ in range(n_rand * 8):
        ops.append((rng.getrandbits(64), rng.getrandbits(64)))
    return ops


_RR = ["add", "sub", "sll", "slt", "sltu", "xor", "srl", "sra", "or_", "and_", "addw", "subw", "sllw", "srlw", "sraw",
       "mul", "mulh", "mulhsu", "mulhu", "div", "divu", "rem", "remu", "mulw", "divw", "divuw", "remw", "remuw"]
_RI = ["addi", "slti", "sltiu", "xori", "ori", "andi", "addiw"]
_SH = [("slli", 64), ("srli", 64), ("srai", 64), ("slliw", 32), ("srliw", 32), ("sraiw", 32)]
_BR = ["beq", "bne", "blt", "bge", "bltu", "bgeu"]
_LD = [("lb", 1), ("lh", 2), ("lw", 4), ("ld", 8), ("lbu", 1), ("lhu", 2), ("lwu", 4)]
_ST = [("sb", 1), ("sh", 2), ("sw", 4), ("sd", 8)]
_CRR = ["c_sub", "c_xor", "c_or", "c_and", "c_subw", "c_addw", "c_mv", "c_add"]


def archtest_program(mn: str, part: int = 0) -> Program:
    """One arch-test style program: exercise `mn` over corner + seeded random operands, store every
    result into begin_signature..end_signature (model_test.h:33-41 conventions), then halt through
    tohost (RVMODEL_HALT, model_test.h:23-27).  `part` (the suite's <mnemonic>-01, -02, ... files): other operand values,
    immediates and offsets, and other registers."""
    data_base = BASE + 0x4000
    a = Asm(BASE)
    d = Asm(data_base)
    d.label("tohost"); d.dword(0)
    d.label("fromhost"); d.dword(0)
    d.label("scratch")
    pat = bytes((i * 37 + 11) & 0xFF for i in range(256))
    d.data(pat)
    d.align_data(16)
    d.label("begin_signature")
    sig_words = 0
    a.label("_start")
    SIG = 31
    T0, T1, T2 = [(5, 6, 7), (28, 29, 30), (10, 11, 12), (18, 19, 20)][part % 4]
    CA, CB = [(8, 9), (10, 11), (12, 13), (14, 15)][part % 4]                  # the compressed forms' register window
    sig_cap = 0x2000
    a.li(SIG, data_base + 0x200)      # patched below once begin_signature is known: same address by construction
    rng = random.Random(_fnv(mn) ^ 0x5EED ^ (part * 0x2545F4914F6CDD1D & MASK64))
    _ops = lambda m: _archtest_operands(m, part=part)

    def emit_store(reg):
        nonlocal sig_words
        a.sd(reg, 0, SIG)
        a.addi(SIG, SIG, 8)
        sig_words += 2

    if mn in _RR:
        for x, y in _ops(mn):
            a.li(T0, x); a.li(T1, y)
            getattr(a, mn)(T2, T0, T1)
            emit_store(T2)
    elif mn in _RI:
        for x, _ in _ops(mn)[:120]:
            a.li(T0, x)
            getattr(a, mn)(T2, T0, rng.randint(-2048, 2047))
            emit_store(T2)
    elif mn in dict(_SH):
        lim = dict(_SH)[mn]
        for x, _ in _ops(mn)[:120]:
            a.li(T0, x)
            getattr(a, mn)(T2, T0, rng.randrange(lim))
            emit_store(T2)
    elif mn in _BR:
        for i, (x, y) in enumerate(_ops(mn)[:120]):
            a.li(T0, x); a.li(T1, y); a.li(T2, 0)
            lbl = f"bt{i}"
            getattr(a, mn)(T0, T1, lbl)
            a.li(T2, 1)
            a.label(lbl)
            emit_store(T2)
    elif mn in dict(_LD):
        sz = dict(_LD)[mn]
        a.li(T0, data_base + 0x10 + 64)
        for i in range(96):
            off = rng.randrange(-64, 128)
            if i % 3:
                off -= off % sz        # mostly aligned, some misaligned
            getattr(a, mn)(T2, off, T0)
            emit_store(T2)
    elif mn in dict(_ST):
        sz = dict(_ST)[mn]
        a.li(T0, data_base + 0x10 + 64)
        for i in range(64):
            off = rng.randrange(-64, 120)
            if i % 3:
                off -= off % sz
            a.li(T1, rng.getrandbits(64))
            getattr(a, mn)(T1, off, T0)
            a.ld(T2, (off // 8) * 8, T0)
            emit_store(T2)
            a.ld(T2, (off // 8) * 8 + 8, T0)
            emit_store(T2)
    elif mn in ("lui", "auipc"):
        for i in range(64):
            getattr(a, mn)(T2, rng.getrandbits(20))
            if mn == "auipc":
                a.auipc(T1, 0); a.sub(T2, T2, T1)
            emit_store(T2)
    elif mn in ("jal", "jalr"):
        for i in range(32):
            lbl = f"jt{i}"
            if mn == "jal":
                a.jal(T2, lbl)
            else:
                a.la(T0, lbl); a.jalr(T2, T0, 0)
            a.li(T2, 0xBAD)
            a.label(lbl)
            a.auipc(T1, 0); a.sub(T2, T2, T1)
            emit_store(T2)
    elif mn in _CRR:
        for x, y in _ops(mn)[:100]:
            a.li(CA, x); a.li(CB, y)
            getattr(a, mn)(CA, CB)
            emit_store(CA)
    elif mn in ("c_addi", "c_addiw", "c_li", "c_andi", "c_slli", "c_srli", "c_srai", "c_lui"):
        for x, _ in _ops(mn)[:100]:
            a.li(CA, x)
            if mn in ("c_slli", "c_srli", "c_srai"):
                getattr(a, mn)(CA, rng.randrange(1, 64))
            elif mn == "c_lui":
                v = rng.choice([i for i in range(-32, 32) if i != 0]); a.c_lui(CA, v)
            else:
                getattr(a, mn)(CA, rng.randrange(-32, 32))
            emit_store(CA)
    elif mn in ("c_lw", "c_ld", "c_sw", "c_sd", "c_lwsp", "c_ldsp", "c_swsp", "c_sdsp"):
        a.li(CB, data_base + 0x10); a.li(2, data_base + 0x10)
        for i in range(32):
            a.li(CA, rng.getrandbits(64))
            if mn == "c_lw": a.c_lw(CA, CB, 4 * rng.randrange(32))
            elif mn == "c_ld": a.c_ld(CA, CB, 8 * rng.randrange(30))
            elif mn == "c_sw": o = 4 * rng.randrange(32); a.c_sw(CA, CB, o); a.ld(CA, (o // 8) * 8, CB)
            elif mn == "c_sd": o = 8 * rng.randrange(30); a.c_sd(CA, CB, o); a.ld(CA, o, CB)
            elif mn == "c_lwsp": a.c_lwsp(CA, 4 * rng.randrange(60))
            elif mn == "c_ldsp": a.c_ldsp(CA, 8 * rng.randrange(30))
            elif mn == "c_swsp": o = 4 * rng.randrange(60); a.c_swsp(CA, o); a.ld(CA, (o // 8) * 8, 2)
            else: o = 8 * rng.randrange(30); a.c_sdsp(CA, o); a.ld(CA, o, 2)
            emit_store(CA)
    elif mn in ("c_j", "c_beqz", "c_bnez", "c_jr", "c_jalr"):
        for i in range(32):
            lbl = f"ct{i}"
            a.li(CA, rng.choice([0, 1, rng.getrandbits(64)])); a.li(T2, 0)
            if mn == "c_j": a.c_j(lbl)
            elif mn == "c_beqz": a.c_beqz(CA, lbl)
            elif mn == "c_bnez": a.c_bnez(CA, lbl)
            elif mn == "c_jr": a.la(CB, lbl); a.c_jr(CB)
            else: a.la(CB, lbl); a.c_jalr(CB)
            a.li(T2, 1)
            a.label(lbl)
            emit_store(T2)
    else:
        raise ValueError(mn)
    assert sig_words * 4 <= sig_cap, (mn, sig_words)
    _exit_tohost(a, data_base)
    # signature region
    while d.pc < data_base + 0x200:
        d.dword(0)
    assert d.symbols["begin_signature"] <= data_base + 0x200
    d.symbols["begin_signature"] = data_base + 0x200
    d.zeros(sig_words * 4)
    d.label("end_signature")
    return _finish(a, f"arch-{mn}" + (f"-{part + 1:02d}" if part else ""), "rv64imc_zicsr", d, mem_size=1 << 16, max_inst=200000, meta=dict(mnemonic=mn, part=part))


def archtest_mnemonics():
    return (_RR + _RI + [s for s, _ in _SH] + _BR + [s for s, _ in _LD] + [s for s, _ in _ST] + ["lui", "auipc", "jal", "jalr"] +
            _CRR + ["c_addi", "c_addiw", "c_li", "c_andi", "c_slli", "c_srli", "c_srai", "c_lui", "c_lw", "c_ld", "c_sw",
                    "c_sd", "c_lwsp", "c_ldsp", "c_swsp", "c_sdsp", "c_j", "c_beqz", "c_bnez", "c_jr", "c_jalr"])


def archtest_suite(parts: int = 1):
    """parts = 1: one program per mnemonic (91); parts = 4: 364 programs, the size of the RV64 I / M / C arch-test suite."""
    return [archtest_program(m, k) for k in range(parts) for m in archtest_mnemonics()]


# ------------------------------------------------------------------------------------------ config 2
def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def compute_seed(i: int) -> int:
    return splitmix64(0xC0FFEE ^ i)


def compute_kernel(iters: int = 4, fp: bool = True, paged: bool = False) -> Program:
    """CoreMark-style mix, seed in a0 (per instance): LCG-filled 16x16 integer matrix multiply,
    linked-list walk, bit-serial CRC16 (data-dependent branches), then an FP section (daxpy,
    Horner fma, div/sqrt chain, fcvt round trips under every rounding mode, single precision mix).
    Result words are stored to `results`; exit by writing 1 to tohost.
    paged: the same kernel the way an operating system would run it - M-mode boot code opens the PMP, builds an Sv39 page
    table (one 2 MiB identity megapage over the arena), turns translation on and drops to S mode; everything after that is
    S-mode code whose every fetch, load and store is translated."""
    data_base = BASE + 0x2000
    a = Asm(BASE)
    d = Asm(data_base)
    d.label("tohost"); d.dword(0)
    d.label("fromhost"); d.dword(0)
    d.label("seed"); d.dword(0)
    d.label("results"); d.zeros(8 * 32)
    d.label("matA"); d.zeros(16 * 16 * 4)
    d.label("matB"); d.zeros(16 * 16 * 4)
    d.label("matC"); d.zeros(16 * 16 * 4)
    d.label("nodes"); d.zeros(64 * 16)
    d.label("buf"); d.zeros(256)
    d.label("vecx"); d.zeros(64 * 8)
    d.label("vecy"); d.zeros(64 * 8)
    d.label("fconst")
    import struct as _st
    for v in (1.0, 0.5, 3.0, 1e-3, 1.0000001, 7.25, 1e300, 1e-300):
        d.data(_st.pack("<d", v))
    S = d.symbols
    # registers
    SEED, RES, I, J, K, T0, T1, T2, T3, P, Q, ACC, ITER = 10, 11, 12, 13, 14, 5, 6, 7, 28, 15, 16, 17, 18

    def lcg(reg):  # reg = reg * 6364136223846793005 + 1442695040888963407
        a.mul(reg, reg, 20); a.add(reg, reg, 21)

    a.label("_start")
    a.li(T0, 0x6000 if fp else 0)
    if fp:
        a.csrs(CSR["mstatus"], T0)
    if paged:
        pt_root, pt_l1 = BASE + 0x8000, BASE + 0x9000
        a.li(T0, (1 << 54) - 1); a.csrw(0x3B0, T0); a.li(T0, 0x1F); a.csrw(0x3A0, T0)        # pmp0: NAPOT over everything, RWX
        a.li(T0, pt_root); a.li(T1, ((pt_l1 >> 12) << 10) | 1); a.sd(T1, 8 * ((BASE >> 30) & 511), T0)
        a.li(T0, pt_l1); a.li(T1, ((BASE >> 12) << 10) | 0xCF); a.sd(T1, 8 * ((BASE >> 21) & 511), T0)   # V R W X A D, 2 MiB leaf
        a.li(T0, (8 << 60) | (pt_root >> 12)); a.csrw(CSR["satp"], T0)
        a.li(T0, 3 << 11); a.csrc(CSR["mstatus"], T0); a.li(T0, 1 << 11); a.csrs(CSR["mstatus"], T0)  # MPP = S
        a.la(T0, "kernel"); a.csrw(CSR["mepc"], T0); a.mret()
        a.label("kernel")
    a.li(RES, S["results"])
    a.li(20, 6364136223846793005); a.li(21, 1442695040888963407)
    a.li(T0, S["seed"]); a.ld(T1, 0, T0); a.xor(SEED, SEED, T1)
    a.li(ITER, iters)
    a.label("outer")
    # ---- 1. fill A, B with LCG; C = A*B (32-bit), checksum
    a.li(P, S["matA"]); a.li(I, 512)
    a.label("fill")
    lcg(SEED); a.srli(T0, SEED, 40); a.sw(T0, 0, P); a.addi(P, P, 4); a.addi(I, I, -1); a.bnez(I, "fill")
    a.li(I, 0); a.li(ACC, 0)
    a.label("mm_i")
    a.li(J, 0)
    a.label("mm_j")
    a.li(K, 0); a.li(T3, 0)
    a.slli(P, I, 6); a.li(T0, S["matA"]); a.add(P, P, T0)          # &A[i][0]
    a.slli(Q, J, 2); a.li(T0, S["matB"]); a.add(Q, Q, T0)          # &B[0][j]
    a.label("mm_k")
    a.lw(T0, 0, P); a.lw(T1, 0, Q); a.mulw(T0, T0, T1); a.addw(T3, T3, T0)
    a.addi(P, P, 4); a.addi(Q, Q, 64); a.addi(K, K, 1); a.li(T0, 16); a.blt(K, T0, "mm_k")
    a.slli(T0, I, 6); a.slli(T1, J, 2); a.add(T0, T0, T1); a.li(T1, S["matC"]); a.add(T0, T0, T1); a.sw(T3, 0, T0)
    a.add(ACC, ACC, T3); a.addi(J, J, 1); a.li(T0, 16); a.blt(J, T0, "mm_j")
    a.addi(I, I, 1); a.li(T0, 16); a.blt(I, T0, "mm_i")
    a.sd(ACC, 0, RES)
    # ---- 2. linked list: node i -> (i*k+1) mod 64 with k odd from the seed; values from LCG; walk 256 steps
    a.li(P, S["nodes"]); a.li(I, 0); a.srli(K, SEED, 7); a.andi(K, K, 62); a.addi(K, K, 5)
    a.label("ll_build")
    a.mul(T0, I, K); a.addi(T0, T0, 1); a.andi(T0, T0, 63); a.slli(T0, T0, 4); a.li(T1, S["nodes"]); a.add(T0, T0, T1)
    a.slli(T1, I, 4); a.add(T1, T1, P); a.sd(T0, 0, T1); lcg(SEED); a.sd(SEED, 8, T1)
    a.addi(I, I, 1); a.li(T0, 64); a.blt(I, T0, "ll_build")
    a.li(I, 256); a.li(ACC, 0); a.mv(Q, P)
    a.label("ll_walk")
    a.ld(T0, 8, Q); a.andi(T1, T0, 1); a.beqz(T1, "ll_even"); a.add(ACC, ACC, T0); a.j("ll_next")
    a.label("ll_even"); a.xor(ACC, ACC, T0)
    a.label("ll_next"); a.ld(Q, 0, Q); a.addi(I, I, -1); a.bnez(I, "ll_walk")
    a.sd(ACC, 8, RES)
    # ---- 3. CRC16 (0xA001) over 64 LCG bytes, bit serial
    a.li(P, S["buf"]); a.li(I, 64)
    a.label("crc_fill"); lcg(SEED); a.srli(T0, SEED, 33); a.sb(T0, 0, P); a.addi(P, P, 1); a.addi(I, I, -1); a.bnez(I, "crc_fill")
    a.li(P, S["buf"]); a.li(I, 64); a.li(ACC, 0xFFFF); a.li(T3, 0xA001)
    a.label("crc_byte"); a.lbu(T0, 0, P); a.xor(ACC, ACC, T0); a.li(J, 8)
    a.label("crc_bit"); a.andi(T1, ACC, 1); a.srli(ACC, ACC, 1); a.beqz(T1, "crc_nox"); a.xor(ACC, ACC, T3)
    a.label("crc_nox"); a.addi(J, J, -1); a.bnez(J, "crc_bit")
    a.addi(P, P, 1); a.addi(I, I, -1); a.bnez(I, "crc_byte")
    a.sd(ACC, 16, RES)
    # ---- 4. atomics: amoadd/amoswap/lr-sc on the results area
    a.addi(T0, RES, 24); a.amoadd_d(T1, SEED, T0); a.amoxor_w(T1, ACC, T0); a.lr_d(T1, T0); a.addi(T1, T1, 3); a.sc_d(T2, T1, T0)
    a.sd(T2, 32, RES)
    if fp:
        # ---- 5. FP: x[i] = int->double(seed bits), y[i] = a*x[i] + y[i]; Horner; div/sqrt; fcvt under all rms
        a.li(P, S["vecx"]); a.li(Q, S["vecy"]); a.li(I, 64)
        a.li(T0, S["fconst"]); a.fld(0, 0, T0); a.fld(1, 8, T0); a.fld(2, 16, T0); a.fld(3, 24, T0); a.fld(4, 32, T0); a.fld(5, 40, T0)
        a.label("fp_fill")
        lcg(SEED); a.srai(T0, SEED, 20); a.fcvt_d_l(6, T0); a.fmul_d(6, 6, 3); a.fsd(6, 0, P)
        lcg(SEED); a.srli(T0, SEED, 30); a.fcvt_d_lu(7, T0); a.fdiv_d(7, 7, 5); a.fsd(7, 0, Q)
        a.addi(P, P, 8); a.addi(Q, Q, 8); a.addi(I, I, -1); a.bnez(I, "fp_fill")
        a.li(P, S["vecx"]); a.li(Q, S["vecy"]); a.li(I, 64); a.fmv_d_x(8, 0)
        a.label("daxpy")
        a.fld(6, 0, P); a.fld(7, 0, Q); a.fmadd_d(7, 4, 6, 7); a.fsd(7, 0, Q); a.fadd_d(8, 8, 7)
        a.fmul_d(9, 6, 6); a.fnmsub_d(9, 9, 1, 7); a.fsub_d(8, 8, 9)
        a.addi(P, P, 8); a.addi(Q, Q, 8); a.addi(I, I, -1); a.bnez(I, "daxpy")
        a.fsd(8, 40, RES)
        # Horner in x = vecx[3] scaled into (0,2): p = ((((c*x+1)*x+0.5)*x+3)*x+1e-3)
        a.li(P, S["vecx"]); a.fld(6, 24, P)
        a.fsgnjx_d(6, 6, 6); a.fadd_d(7, 6, 0); a.fdiv_d(6, 6, 7); a.fadd_d(6, 6, 6)     # x = 2|v|/(|v|+1)
        a.fmv_d_x(9, 0); a.fadd_d(9, 9, 4)
        for c in (0, 1, 2, 3):
            a.fmadd_d(9, 9, 6, c)
        a.fsd(9, 48, RES)
        # div / sqrt chain
        a.fsqrt_d(10, 7); a.fdiv_d(11, 0, 10); a.fmul_d(11, 11, 11); a.fmsub_d(11, 11, 7, 0); a.fsd(11, 56, RES)
        # conversions under every rounding mode + flags readback
        for rm in range(5):
            a.fcvt_l_d(T0, 9, rm); a.fcvt_w_d(T1, 8, rm); a.add(T0, T0, T1); a.fcvt_wu_d(T1, 11, rm); a.add(T0, T0, T1)
            a.fcvt_s_d(12, 9, rm); a.fcvt_d_s(13, 12); a.fsub_d(13, 13, 9); a.fmv_x_d(T1, 13); a.xor(T0, T0, T1)
            a.sd(T0, 64 + 8 * rm, RES)
        # single precision mix with dynamic rounding mode from the seed
        a.srli(T0, SEED, 13); a.andi(T0, T0, 3); a.csrw(CSR["frm"], T0)
        a.fcvt_s_d(14, 8); a.fcvt_s_d(15, 6); a.fmul_s(16, 14, 15); a.fadd_s(16, 16, 15); a.fdiv_s(17, 16, 14); a.fsqrt_s(18, 15)
        a.fmadd_s(19, 16, 17, 18); a.fnmadd_s(19, 19, 15, 14); a.fmin_s(20, 19, 18); a.fmax_s(21, 19, 18)
        a.flt_s(T0, 20, 21); a.feq_s(T1, 20, 20); a.fclass_s(T2, 19); a.slli(T1, T1, 8); a.or_(T0, T0, T1); a.slli(T2, T2, 16); a.or_(T0, T0, T2)
        a.fmv_x_w(T1, 19); a.slli(T1, T1, 32); a.xor(T0, T0, T1); a.sd(T0, 104, RES)
        a.fsw(19, 112, RES); a.flw(22, 112, RES); a.fcvt_w_s(T0, 22); a.fcvt_lu_s(T1, 21); a.add(T0, T0, T1); a.sd(T0, 120, RES)
        a.csrr(T0, CSR["fcsr"]); a.sd(T0, 128, RES)
        a.csrw(CSR["frm"], 0)
    a.addi(ITER, ITER, -1)
    a.beqz(ITER, "done")
    a.j("outer")
    a.label("done")
    if not paged:
        a.csrr(T0, CSR["minstret"]); a.sd(T0, 136, RES)
    _exit_tohost(a, S["tohost"])
    isa = "rv64imafdc_zicsr" if fp else "rv64imac_zicsr"
    if paged:
        isa = isa.replace("c_zicsr", "csu_zicsr")
    return _finish(a, "compute-sv39" if paged else "compute", isa, d, mem_size=1 << 16, max_inst=1 << 26, meta=dict(iters=iters))


def variant(p: "Program", k: int) -> "Program":
    """A distinct program from a generated one: the k-th variant XORs k into the immediate field of one 32-bit word of the code
    (both in the ELF image and in the segment list the oracle loads).  Whatever the patched word decodes to - another immediate,
    another instruction, an illegal one - device and oracle run the same bytes."""
    if k == 0:
        return p
    import copy
    paddr, data, flags = p.segments[0]
    off = (len(data) // 2 + 4 * (k % 17)) & ~3
    word = int.from_bytes(data[off:off + 4], "little") ^ ((k & 0xfff) << 20)
    q = copy.copy(p)
    nd = data[:off] + word.to_bytes(4, "little") + data[off + 4:]
    q.segments = [(paddr, nd, flags)] + list(p.segments[1:])
    at = p.elf.find(data)
    assert at >= 0
    q.elf = p.elf[:at] + nd + p.elf[at + len(data):]
    q.name = f"{p.name}.v{k}"
    return q


# ------------------------------------------------------------------------------------------ fuzz
def fuzz_program(seed: int, n: int = 400, fp: bool = True, compressed: bool = True, interrupts: bool = False) -> Program:
    """Random instruction mix (M-mode, bare): ALU, loads/stores into mapped and unmapped addresses,
    forward branches, M, A, F/D, CSR accesses, ecall/ebreak/illegal words.  A trap handler skips the
    trapping instruction (length from the low two bits at mepc) and returns.  interrupts=True enables S/U in
    the ISA, lets the mix write mie/mip/mideleg/sie/sip and set mstatus.MIE so that software-raised
    supervisor-level interrupts are taken in M mode (direct or vectored mtvec by seed parity)."""
    rng = random.Random(seed)
    data_base = BASE + 0x4000
    a = Asm(BASE)
    d = Asm(data_base)
    d.label("tohost"); d.dword(0)
    d.label("fromhost"); d.dword(0)
    d.label("data")
    d.data(bytes(rng.getrandbits(8) for _ in range(1024)))
    fvals = [0, 0x8000000000000000, 0x3FF0000000000000, 0xBFF0000000000000, 0x7FF0000000000000, 0xFFF0000000000000,
             0x7FF8000000000000, 0x7FF0000000000001, 0x0000000000000001, 0x000FFFFFFFFFFFFF, 0x0010000000000000,
             0x7FEFFFFFFFFFFFFF, 0x4340000000000000, 0xC3E0000000000000, 0x3FE0000000000000, 0x41DFFFFFFFC00000,
             0xFFFFFFFF3F800000, 0xFFFFFFFF7F800000, 0xFFFFFFFF7FC00000, 0xFFFFFFFF7F800001, 0xFFFFFFFF00000001,
             0xFFFFFFFF00800000, 0xFFFFFFFF7F7FFFFF, 0xFFFFFFFFCF000000, 0xFFFFFFFF4F000000, 0x000000003F800000]
    d.label("fdata")
    for v in fvals:
        d.dword(v)
    for _ in range(32 - len(fvals)):
        d.dword(rng.getrandbits(64))
    S = d.symbols
    H1, H2 = 30, 31     # reserved for the trap handler
    a.label("_start")
    a.j("init")
    a.align(4)
    a.label("handler")
    if interrupts:
        # vectored layout (used with mtvec.MODE=1 on odd seeds): exceptions enter at base, interrupt cause c at base+4c
        a.jal(0, "exc")
        for _ in range(1, 14):
            a.jal(0, "irq")
        a.label("irq")           # software-raised supervisor-level interrupt taken in M mode: drop it and resume
        a.csrw(CSR["mip"], 0); a.mret()
        a.label("exc")
        a.csrr(H1, CSR["mcause"]); a.bltz(H1, "irq")
    a.csrr(H1, CSR["mepc"]); a.lhu(H2, 0, H1); a.andi(H2, H2, 3); a.addi(H2, H2, -3)
    a.bnez(H2, "h2"); a.addi(H1, H1, 2)
    a.label("h2"); a.addi(H1, H1, 2); a.csrw(CSR["mepc"], H1); a.mret()
    a.label("init")
    a.la(5, "handler")
    if interrupts and (seed & 1):
        a.addi(5, 5, 1)
    a.csrw(CSR["mtvec"], 5)
    if fp:
        a.li(5, 0x2000); a.csrs(CSR["mstatus"], 5)
        a.li(5, S["fdata"])
        for i in range(32):
            a.fld(i, 8 * i, 5)
    for r in range(1, 30):
        a.li(r, rng.choice([rng.getrandbits(64), rng.getrandbits(12), S["data"] + 8 * rng.randrange(100), (1 << 63), MASK64, 0]))
    regs = list(range(0, 30))
    ptrs = [8, 9, 18, 19]
    for r in ptrs:
        a.li(r, S["data"] + 256 + 8 * rng.randrange(32))
    rr_ops = _RR
    labels = 0
    pending = []   # (countdown, label)
    amo = ["amoswap", "amoadd", "amoxor", "amoand", "amoor", "amomin", "amomax", "amominu", "amomaxu"]
    fp3 = ["fadd", "fsub", "fmul", "fdiv", "fsgnj", "fsgnjn", "fsgnjx", "fmin", "fmax"]
    safe_csrs = [CSR["mscratch"], CSR["mepc"], CSR["mcause"], CSR["mtval"], CSR["fflags"], CSR["frm"], CSR["fcsr"],
                 CSR["mcycle"], CSR["minstret"], CSR["mhartid"], CSR["misa"], CSR["mie"], CSR["mip"], CSR["mcounteren"],
                 CSR["cycle"], CSR["sstatus"], CSR["sepc"], 0x7C0, CSR["mvendorid"], CSR["menvcfg"], CSR["mcountinhibit"], CSR["mstatus"],
                 0xB03, 0x33F, 0x747, 0x7B0]         # mhpmcounter3 / mhpmevent31 / mseccfg: hardwired zero; dcsr: debug mode only
    if interrupts:
        safe_csrs += [CSR["mie"], CSR["mip"], CSR["mip"], CSR["mideleg"], CSR["medeleg"], CSR["sie"], CSR["sip"], CSR["mstatus"]]
    for i in range(n):
        for p in list(pending):
            p[0] -= 1
            if p[0] <= 0:
                a.label(p[1]); pending.remove(p)
        k = rng.random()
        rd = rng.choice([r for r in regs if r not in ptrs]); rs1 = rng.choice(regs); rs2 = rng.choice(regs)
        if k < 0.30:
            getattr(a, rng.choice(rr_ops))(rd, rs1, rs2)
        elif k < 0.42:
            getattr(a, rng.choice(_RI))(rd, rs1, rng.randint(-2048, 2047))
        elif k < 0.47:
            m, lim = rng.choice(_SH); getattr(a, m)(rd, rs1, rng.randrange(lim))
        elif k < 0.50:
            (a.lui if rng.random() < 0.5 else a.auipc)(rd, rng.getrandbits(20))
        elif k < 0.62:
            m, sz = rng.choice(_LD + _ST)
            base = rng.choice(ptrs) if rng.random() < 0.9 else rs1
            off = rng.randrange(-256, 256)
            if rng.random() < 0.8:
                off -= off % sz
            if m.startswith("l"):
                getattr(a, m)(rd, off, base)
            else:
                getattr(a, m)(rs2, off, base)
        elif k < 0.70:
            lbl = f"L{labels}"; labels += 1
            getattr(a, rng.choice(_BR))(rs1, rs2, lbl)
            pending.append([rng.randrange(1, 6), lbl])
        elif k < 0.72:
            lbl = f"L{labels}"; labels += 1
            a.jal(rd if rng.random() < 0.5 else 0, lbl)
            pending.append([rng.randrange(1, 4), lbl])
        elif k < 0.77:
            sfx = rng.choice(["w", "d"]); base = rng.choice(ptrs + [rs1])
            r = rng.random()
            if r < 0.25:
                getattr(a, f"lr_{sfx}")(rd, base)
            elif r < 0.5:
                getattr(a, f"sc_{sfx}")(rd, rs2, base)
            else:
                getattr(a, f"{rng.choice(amo)}_{sfx}")(rd, rs2, base)
        elif k < 0.90 and fp:
            f1, f2, f3, fd = (rng.randrange(32) for _ in range(4)); sfx = rng.choice("sd")
            rm = rng.choice([0, 1, 2, 3, 4, 7, 7, 7, 5])
            r = rng.random()
            if r < 0.35:
                op = rng.choice(fp3)
                if op in ("fadd", "fsub", "fmul", "fdiv"):
                    getattr(a, f"{op}_{sfx}")(fd, f1, f2, rm)
                else:
                    getattr(a, f"{op}_{sfx}")(fd, f1, f2)
            elif r < 0.50:
                getattr(a, f"{rng.choice(['fmadd', 'fmsub', 'fnmsub', 'fnmadd'])}_{sfx}")(fd, f1, f2, f3, rm)
            elif r < 0.56:
                getattr(a, f"fsqrt_{sfx}")(fd, f1, rm)
            elif r < 0.68:
                getattr(a, f"fcvt_{rng.choice(['w', 'wu', 'l', 'lu'])}_{sfx}")(rd, f1, rm)
            elif r < 0.78:
                getattr(a, f"fcvt_{sfx}_{rng.choice(['w', 'wu', 'l', 'lu'])}")(fd, rs1, rm)
            elif r < 0.84:
                getattr(a, f"{rng.choice(['feq', 'flt', 'fle'])}_{sfx}")(rd, f1, f2)
            elif r < 0.88:
                getattr(a, f"fclass_{sfx}")(rd, f1)
            elif r < 0.92:
                rng.choice([a.fmv_x_w, a.fmv_x_d])(rd, f1) if rng.random() < 0.5 else rng.choice([a.fmv_w_x, a.fmv_d_x])(fd, rs1)
            elif r < 0.96:
                (a.fcvt_s_d if rng.random() < 0.5 else a.fcvt_d_s)(fd, f1)
            else:
                base = rng.choice(ptrs); off = 8 * rng.randrange(-16, 16)
                rng.choice([a.flw, a.fld])(fd, off, base) if rng.random() < 0.5 else rng.choice([a.fsw, a.fsd])(f1, off, base)
        elif k < 0.94:
            csr = rng.choice(safe_csrs)
            op = rng.choice(["csrrw", "csrrs", "csrrc", "csrrwi", "csrrsi", "csrrci"])
            if interrupts and rng.random() < 0.35:      # raise / enable the supervisor software interrupt, or all three S bits
                which = rng.random()
                if which < 0.4:
                    a.csrrsi(0, CSR["mip"], 2)
                elif which < 0.8:
                    a.csrrsi(0, CSR["mie"], 2)
                else:
                    a.li(rd if rd else 1, 0x222); a.csrrs(0, CSR["mie"], rd if rd else 1); a.csrrs(0, CSR["mip"], rd if rd else 1)
                continue
            if csr == CSR["mstatus"]:
                op = rng.choice(["csrrs", "csrrc", "csrrsi", "csrrci"]); src = rng.choice([0, 3, 5])  # keep FS/MIE sane-ish
                if interrupts and rng.random() < 0.6:
                    op, src = "csrrsi", 8                                                             # mstatus.MIE on
                getattr(a, op)(rd, csr, src if op.endswith("i") else 0)
            else:
                getattr(a, op)(rd, csr, rng.randrange(32) if op.endswith("i") else rs1)
        elif k < 0.96:
            rng.choice([a.ecall, a.ebreak, a.fence, a.fence_i, a.wfi])()
        elif k < 0.98 and compressed:
            c = rng.random()
            if c < 0.3: a.c_addi(rd if rd else 1, rng.randrange(-32, 32))
            elif c < 0.5: a.c_li(rd if rd else 1, rng.randrange(-32, 32))
            elif c < 0.7: a.c_mv(rd if rd else 1, rs2 if rs2 else 1)
            elif c < 0.85: a.c_add(rd if rd else 1, rs2 if rs2 else 1)
            else: a.c_nop()
        else:
            w = rng.getrandbits(32)
            if rng.random() < 0.5:
                w |= 3
                a.word(w)        # random 32-bit word (mostly illegal, sometimes a valid instruction)
            else:
                a.half(rng.choice([0, 0x0000, 0xFFFF & ~3 & w, 0x8000, 0x6000, 0xE000]) & ~0x3 | rng.choice([0, 0, 1, 2]))
    for p in pending:
        a.label(p[1])
    a.align(4)
    _exit_tohost(a, S["tohost"], tmp1=30, tmp2=31)
    isa = ("rv64imafdc" if fp else "rv64imac") + ("su_zicsr" if interrupts else "_zicsr")
    return _finish(a, f"fuzz-{seed}", isa, d, mem_size=1 << 16, max_inst=20000, meta=dict(seed=seed))


# ------------------------------------------------------------------------------------------ PMP
def _napot(base: int, size: int) -> int:
    """pmpaddr value of the naturally aligned power-of-two region [base, base + size), size >= 8."""
    return ((base >> 2) | (size // 8 - 1)) & 0x003FFFFFFFFFFFFF


def pmp_program(seed: int, n: int = 300) -> Program:
    """Physical memory protection: M mode programs TOR / NA4 / NAPOT entries (some locked) over a data arena, a small
    code island and a background entry, drops to U (even seeds) or S (odd seeds) and runs a random mix of loads, stores,
    AMOs, LR/SC and FP accesses at region edges (aligned and not), jumps into executable and non-executable islands and
    a 4-byte instruction that straddles the end of an executable region.  `ecall` with a7 != 0 asks M mode to access
    memory itself (locked entries apply), under MPRV (U/S permissions apply), or to rewrite pmpcfg2 / pmpaddr9 (which may
    hit the reference's stale-TOR-bound quirk: an UNSUPPORTED stop on every implementation).  Every trap goes to one M
    handler that skips the faulting instruction (fetch faults: resumes at ra).  a7 == 0 exits through tohost."""
    rng = random.Random(0x706D70 ^ seed)
    data_base = BASE + 0x4000
    a = Asm(BASE)
    d = Asm(data_base)
    d.label("tohost"); d.dword(0)
    d.label("fromhost"); d.dword(0)
    d.org(data_base + 0x100)
    d.label("arena")
    d.data(bytes(rng.getrandbits(8) for _ in range(0x700)))
    d.org(data_base + 0x1000)
    isl = Asm(data_base + 0x1000)
    ISL = data_base + 0x1000
    # island 1 at ISL: executable [ISL, ISL+8); the 4-byte instruction at ISL+6 crosses into non-executable memory
    isl.c_nop(); isl.addi(6, 6, 1); isl.addi(6, 6, 2); isl.c_nop(); isl.ret()
    isl.org(ISL + 0x40)
    isl.addi(6, 6, 3); isl.ret()                    # island 2 at ISL+0x40: fully executable (16 bytes NAPOT)
    isl.org(ISL + 0x80)
    isl.addi(6, 6, 4); isl.ret()                    # island 3 at ISL+0x80: executable only if the background entry says so
    d.data(isl.assemble())
    S = d.symbols
    ARENA = S["arena"]
    perms = [0, 1, 3, 4, 5, 7]
    cfg = [0] * 64
    addr = [0] * 64
    edges = []

    def region(ix, kind, lo, size, p, lock=False):
        c = p | (0x80 if lock else 0)
        if kind == "tor":
            addr[ix - 1] = lo >> 2; addr[ix] = (lo + size) >> 2; cfg[ix] = c | (1 << 3)
        elif kind == "na4":
            addr[ix] = lo >> 2; cfg[ix] = c | (2 << 3); size = 4
        else:
            addr[ix] = _napot(lo, size); cfg[ix] = c | (3 << 3)
        edges.extend([lo, lo + size])

    region(0, "napot", BASE, 0x4000, 5)                                  # program text: r-x
    region(1, "napot", S["tohost"], 16, 0)                               # tohost / fromhost: nothing below M
    edges.clear()                                                        # (M-mode service stores stay away from text and tohost)
    ix = 2
    while ix < 16:                                                       # entries 2..7 over arena A, 8..15 over arena B
        lo_area = ARENA + (0x000 if ix < 8 else 0x400)
        kind = rng.choice(["tor", "na4", "napot", "napot", "off"])
        lock = rng.random() < 0.2
        p = rng.choice(perms)
        if kind == "tor" and ix + 1 < (8 if ix < 8 else 16):
            lo = lo_area + 4 * rng.randrange(0, 0x80); region(ix + 1, "tor", lo, 4 * rng.randrange(1, 0x40), p, lock); ix += 2
        elif kind == "na4":
            region(ix, "na4", lo_area + 4 * rng.randrange(0, 0xF0), 4, p, lock); ix += 1
        elif kind == "napot":
            size = rng.choice([8, 16, 32, 64, 128, 256]); region(ix, "napot", lo_area + size * rng.randrange(0, 0x300 // size), size, p, lock); ix += 1
        else:
            addr[ix] = (lo_area + 4 * rng.randrange(0, 0xF0)) >> 2; ix += 1
    region(17, "tor", ISL, 8, 5)                                         # island 1
    region(18, "napot", ISL + 0x40, 16, rng.choice([5, 4, 7]))           # island 2
    region(40, "na4", ARENA + 0x3F0, 4, rng.choice(perms), rng.random() < 0.3)
    region(63, "napot", 0, 1 << 57, rng.choice([3, 3, 7, 1, 0]))         # background
    edges = sorted(set(edges) - {0, 1 << 57})

    N1, H0, H3, H1, H2 = 27, 28, 29, 30, 31     # reserved for the handler
    MPRV = 1 << 17
    a.label("_start")
    a.j("init")
    a.align(4)
    a.label("handler")
    # a fault of M mode itself (locked entries, MPRV) can only come from a load / store of this handler, all 4 bytes long:
    # step over it with registers of its own so that the interrupted path carries on
    a.csrr(N1, CSR["mstatus"]); a.srli(N1, N1, 11); a.andi(N1, N1, 3); a.addi(N1, N1, -3); a.bnez(N1, "from_below")
    a.csrr(N1, CSR["mepc"]); a.addi(N1, N1, 4); a.csrw(CSR["mepc"], N1); a.mret()
    a.label("from_below")
    a.csrr(H1, CSR["mcause"])
    a.addi(H2, H1, -1); a.beqz(H2, "fetch_fault")
    a.addi(H2, H1, -8); a.sltiu(H2, H2, 4); a.bnez(H2, "is_ecall")
    a.label("skip")
    a.csrr(H1, CSR["mepc"]); a.lhu(H2, 0, H1); a.andi(H2, H2, 3); a.addi(H2, H2, -3)
    a.bnez(H2, "h2"); a.addi(H1, H1, 2)
    a.label("h2"); a.addi(H1, H1, 2); a.csrw(CSR["mepc"], H1); a.mret()
    a.label("fetch_fault")
    a.csrr(H1, CSR["mepc"]); a.beq(H1, 1, "exit")                        # ra itself is not executable (clobbered by arena code): give up
    a.csrw(CSR["mepc"], 1); a.mret()
    a.label("is_ecall")
    a.beqz(17, "exit")
    a.csrr(H0, CSR["mepc"]); a.csrr(H3, CSR["mstatus"])                  # the service may fault and nest
    for k, lbl in ((1, "svc_store"), (2, "svc_mprv"), (3, "svc_load"), (4, "svc_cfg"), (5, "svc_addr")):
        a.li(H2, k); a.beq(17, H2, lbl)
    a.j("svc_done")
    a.label("svc_load"); a.ld(H2, 0, 10); a.lbu(H2, 3, 10); a.j("svc_done")
    a.label("svc_store"); a.sd(11, 0, 10); a.sb(11, 5, 10); a.amoadd_w(H2, 11, 10); a.j("svc_done")
    a.label("svc_mprv")
    a.li(H2, MPRV); a.csrs(CSR["mstatus"], H2); a.lw(H2, 0, 10); a.sw(11, 4, 10)
    a.li(H2, MPRV); a.csrc(CSR["mstatus"], H2); a.j("svc_done")
    a.label("svc_cfg"); a.csrrw(H2, 0x3A2, 11); a.j("svc_done")
    a.label("svc_addr"); a.csrrw(H2, 0x3B9, 11)
    a.label("svc_done")
    a.csrw(CSR["mstatus"], H3); a.addi(H0, H0, 4); a.csrw(CSR["mepc"], H0); a.mret()
    a.label("exit")
    _exit_tohost(a, S["tohost"], tmp1=30, tmp2=31)

    a.label("init")
    a.la(5, "handler"); a.csrw(CSR["mtvec"], 5)
    a.li(5, 0x2000); a.csrs(CSR["mstatus"], 5)                           # FS on
    for i in range(64):
        if addr[i]:
            a.li(5, addr[i]); a.csrw(0x3B0 + i, 5)
    for k in range(8):
        v = 0
        for j in range(8):
            v |= cfg[8 * k + j] << (8 * j)
        if v:
            a.li(5, v); a.csrw(0x3A0 + 2 * k, 5)
    for k in (0, 2, 4, 10, 14):                                          # read everything back into the trace
        a.csrr(5, 0x3A0 + k)
    for i in (0, 1, 2, 9, 17, 63):
        a.csrr(5, 0x3B0 + i)
    a.li(5, 0x3A2); a.csrrw(6, 0x3A1, 5)                                 # odd pmpcfg: not implemented on RV64 (illegal)
    for r in range(1, 28):
        a.li(r, rng.choice([rng.getrandbits(64), rng.getrandbits(12), ARENA + 8 * rng.randrange(0xE0), 0]))
    a.li(5, 3 << 11); a.csrc(CSR["mstatus"], 5)
    if seed & 1:
        a.li(5, 1 << 11); a.csrs(CSR["mstatus"], 5)                      # MPP = S
    a.la(5, "user"); a.csrw(CSR["mepc"], 5); a.mret()
    a.align(4)
    a.label("user")

    def pick_addr(m_mode=False):
        r = rng.random()
        if r < 0.65 and edges:
            return rng.choice(edges) + rng.randrange(-9, 9)
        if r < 0.9 or m_mode:
            return ARENA + rng.randrange(0, 0x700)
        return rng.choice([S["tohost"], S["tohost"] + 8, S["tohost"] + 13, BASE + 0x100, BASE + 0x3FFC, BASE - 8, BASE + 0x10000 - 4, ISL + 0x104, 0x1000])

    P = 9            # address register
    amo = ["amoswap", "amoadd", "amoxor", "amoand", "amoor", "amomin", "amomax", "amominu", "amomaxu"]
    regs = [r for r in range(2, 27) if r not in (P, 10, 11, 17)]
    for i in range(n):
        k = rng.random()
        rd = rng.choice(regs); rs2 = rng.choice(regs)
        if k < 0.45:
            m, sz = rng.choice(_LD + _ST)
            t = pick_addr()
            if rng.random() < 0.6:
                t -= t % sz
            off = rng.randrange(-64, 64)
            a.li(P, (t - off) & MASK64)
            getattr(a, m)(rd if m.startswith("l") else rs2, off, P)
        elif k < 0.57:
            sfx = rng.choice(["w", "d"]); t = pick_addr()
            if rng.random() < 0.85:
                t -= t % (4 if sfx == "w" else 8)
            a.li(P, t & MASK64)
            r = rng.random()
            if r < 0.25:
                getattr(a, f"lr_{sfx}")(rd, P)
            elif r < 0.5:
                getattr(a, f"lr_{sfx}")(rd, P); getattr(a, f"sc_{sfx}")(rd, rs2, P)
            else:
                getattr(a, f"{rng.choice(amo)}_{sfx}")(rd, rs2, P)
        elif k < 0.65:
            t = pick_addr(); t -= t % rng.choice([1, 4, 8])
            a.li(P, t & MASK64)
            rng.choice([a.flw, a.fld, a.fsw, a.fsd])(rng.randrange(32), 0, P)
        elif k < 0.75:
            t = rng.choice([ISL, ISL, ISL + 0x40, ISL + 0x80, ISL + 2, ISL + 8, ARENA + 4 * rng.randrange(0x100), BASE + 0x20000])
            a.li(P, t); a.jalr(1, P, 0)
        elif k < 0.88:
            svc = rng.choice([1, 1, 2, 2, 3, 3, 4, 5])
            a.li(17, svc)
            if svc == 4:
                v = 0
                for j in range(8):
                    v |= (rng.choice(perms) | (rng.choice([0, 1, 2, 3]) << 3) | (0x80 if rng.random() < 0.1 else 0)) << (8 * j)
                a.li(11, v)
            elif svc == 5:
                a.li(11, (ARENA + 0x400 + 4 * rng.randrange(0xF0)) >> 2)
            else:
                t = pick_addr(True); t -= t % rng.choice([1, 8]); a.li(10, t & MASK64); a.li(11, rng.getrandbits(64))
            a.ecall()
        elif k < 0.92:
            rng.choice([lambda: a.csrr(rd, 0x3A0), lambda: a.csrw(0x3B3, rs2), lambda: a.csrr(rd, CSR["mstatus"]),
                        lambda: a.csrr(rd, CSR["sstatus"]), a.ebreak, a.wfi, a.fence_i])()
        else:
            getattr(a, rng.choice(_RR))(rd, rng.choice(regs), rs2)
    a.li(17, 0); a.ecall()
    a.label("hang"); a.j("hang")
    assert a.here() < BASE + 0x4000, "text outgrew its PMP region"
    return _finish(a, f"pmp-{seed}", "rv64imafdcsu_zicsr", d, mem_size=1 << 16, max_inst=40000, meta=dict(seed=seed))


# ------------------------------------------------------------------------------------------ Sv39 paging
_PTE_V, _PTE_R, _PTE_W, _PTE_X, _PTE_U, _PTE_G, _PTE_A, _PTE_D = 1, 2, 4, 8, 16, 32, 64, 128


def sv39_program(seed: int, n: int = 300, exec_body: bool = False, high: bool = False) -> Program:
    """Sv39 paging: M mode opens PMP, builds a three-level page table (4 KiB pages with mixed
    R/W/X/U/A/D bits, a 2 MiB window over the whole arena, a misaligned superpage, malformed PTEs), and drops to S (odd
    seeds) or U (even seeds) at a virtual address that is not the physical one.  The body mixes loads, stores, AMOs, LR/SC
    and FP accesses (aligned and not, also across page boundaries with different permissions), jumps into mapped,
    unmapped and non-executable pages, 4-byte instructions that straddle a page boundary, page-table edits through the
    window *without* sfence.vma (stale TLB entries are architecturally visible in the reference), the sfence.vma forms,
    sstatus.SUM / MXR toggles and satp rewrites (new ASID).  `ecall` services run in M mode: MPRV loads and stores (translated
    with the caller's privilege) and satp switches.  Every trap goes to one M handler that steps over the faulting
    instruction (fetch faults resume at ra).  Run with a small `tlb_entries` so the replacement policy matters.
    high: root entry 511 maps the code and data pages a second time in the UPPER half of the Sv39 address space
    (0xFFFFFFFFC0000000 + ...: bits 63:39 all ones), and accesses, jumps and service calls go there as well as to non-canonical
    addresses on either side of the hole (VirtMem.cpp: bits 63:39 must equal bit 38).
    exec_body: mark the body's segment executable in the ELF, so that the loader pre-decodes it too (the image then spans
    boot code .. body, about 80 KiB) and S / U-mode steps can take the TLB-hit fast path."""
    rng = random.Random(0x5339 ^ seed)
    smode = bool(seed & 1)
    MEM = 1 << 18
    TOHOST = BASE + 0x4000
    PT_ROOT, PT_L1, PT_L0A, PT_L0B = BASE + 0x8000, BASE + 0x9000, BASE + 0xA000, BASE + 0xB000
    PA_CODE, VA_CODE = BASE + 0x10000, 0x10000            # body code: 4 pages
    PA_DATA, VA_DATA = BASE + 0x20000, 0x400000           # data: 16 pages (L0B)
    VA_WIN = 0x40000000                                   # 2 MiB window onto the arena (L1[0] of root[1])
    U = 0 if smode else _PTE_U
    FULL = _PTE_V | _PTE_R | _PTE_W | _PTE_X | _PTE_A | _PTE_D

    def pte(pa, flags, hi=0):
        return (hi << 54) | ((pa >> 12) << 10) | flags

    pt = {}
    pt[PT_ROOT + 8 * 0] = pte(PT_L1, _PTE_V)
    pt[PT_ROOT + 8 * 1] = pte(PT_L1 + 0x800, _PTE_V)      # second half of the L1 page serves root[1]
    pt[PT_ROOT + 8 * 2] = pte(BASE + 0x1000, FULL | U)    # misaligned 1 GiB leaf
    pt[PT_ROOT + 8 * 3] = pte(PT_L1, _PTE_V | _PTE_A)     # non-leaf with A set
    HI = 0xFFFFFFFFC0000000
    if high:
        pt[PT_ROOT + 8 * 511] = pte(PT_L1, _PTE_V)        # the lower GiB's tables again, for VA 0xFFFF_FFFF_C000_0000 ..
    pt[PT_L1 + 8 * 0] = pte(PT_L0A, _PTE_V)               # VA 0x0000_0000 .. : code pages
    pt[PT_L1 + 8 * 2] = pte(PT_L0B, _PTE_V)               # VA 0x0040_0000 .. : data pages
    pt[PT_L1 + 8 * 3] = pte(BASE + 0x1000, FULL | U)      # misaligned 2 MiB leaf
    pt[PT_L1 + 8 * 4] = pte(BASE, FULL | U, hi=rng.choice([1, 0x80, 0x200]))   # reserved / PBMT / N bits
    pt[PT_L1 + 0x800 + 8 * 0] = pte(BASE, _PTE_V | _PTE_R | _PTE_W | _PTE_A | _PTE_D | U)   # the window (2 MiB leaf)
    for i in range(4):                                    # code pages: x, or rx; page 3 sometimes not executable
        f = _PTE_V | _PTE_X | _PTE_A | U | rng.choice([0, _PTE_R])
        if i == 3 and rng.random() < 0.5:
            f = _PTE_V | _PTE_R | _PTE_A | U
        pt[PT_L0A + 8 * (16 + i)] = pte(PA_CODE + 0x1000 * i, f)
    data_flags = []
    for i in range(16):
        k = rng.random()
        if k < 0.5:
            f = _PTE_V | _PTE_R | _PTE_W | _PTE_A | _PTE_D | U
        elif k < 0.8:
            f = _PTE_V | _PTE_A | rng.choice([_PTE_R, _PTE_R | _PTE_W, _PTE_X, _PTE_R | _PTE_X]) | rng.choice([0, _PTE_D]) | rng.choice([U, U, _PTE_U ^ U]) | rng.choice([0, _PTE_G])
        else:
            f = rng.randrange(256)
        data_flags.append(f)
        pt[PT_L0B + 8 * i] = pte(PA_DATA + 0x1000 * rng.randrange(16), f)

    d = Asm(TOHOST)
    d.label("tohost"); d.dword(0)
    d.label("fromhost"); d.dword(0)
    d.org(PT_ROOT)
    img = bytearray(0x4000)
    for addr, v in pt.items():
        img[addr - PT_ROOT:addr - PT_ROOT + 8] = v.to_bytes(8, "little")
    d.data(bytes(img))
    d.org(PA_DATA)
    d.data(bytes(rng.getrandbits(8) for _ in range(0x10000)))

    a = Asm(BASE)
    N1, H0, H3, H1, H2 = 27, 28, 29, 30, 31     # reserved for the handler
    MPRV, MXR = 1 << 17, 1 << 19
    a.label("_start")
    a.j("init")
    a.align(4)
    a.label("handler")
    # a fault of M mode itself can only come from an MPRV access of this handler (4-byte instructions): step over it
    a.csrr(N1, CSR["mstatus"]); a.srli(N1, N1, 11); a.andi(N1, N1, 3); a.addi(N1, N1, -3); a.bnez(N1, "from_below")
    a.csrr(N1, CSR["mepc"]); a.addi(N1, N1, 4); a.csrw(CSR["mepc"], N1); a.mret()
    a.label("from_below")
    a.csrr(H0, CSR["mepc"]); a.csrr(H3, CSR["mstatus"])                  # what follows may fault and nest
    a.csrr(H1, CSR["mcause"])
    a.addi(H2, H1, -1); a.beqz(H2, "fetch_fault")
    a.addi(H2, H1, -12); a.beqz(H2, "fetch_fault")
    a.addi(H2, H1, -8); a.sltiu(H2, H2, 4); a.bnez(H2, "is_ecall")
    # length of the faulting instruction: read it the way it was fetched (MPRV + MXR: the caller's translation)
    a.li(H2, 0); a.li(H1, MPRV | MXR); a.csrs(CSR["mstatus"], H1); a.lhu(H2, 0, H0); a.csrc(CSR["mstatus"], H1)
    a.andi(H2, H2, 3); a.addi(H2, H2, -3); a.bnez(H2, "len2"); a.addi(H0, H0, 2)
    a.label("len2"); a.addi(H0, H0, 2); a.j("resume")
    a.label("fetch_fault")
    a.beq(H0, 1, "exit")                                                 # ra itself cannot be fetched: give up
    a.mv(H0, 1); a.j("resume")
    a.label("is_ecall")
    a.beqz(17, "exit")
    a.addi(H0, H0, 4)
    for k, lbl in ((1, "svc_mprv_ld"), (2, "svc_mprv_st"), (3, "svc_satp"), (4, "svc_sfence")):
        a.li(H2, k); a.beq(17, H2, lbl)
    a.j("resume")
    a.label("svc_mprv_ld"); a.li(H1, MPRV); a.csrs(CSR["mstatus"], H1); a.ld(H2, 0, 10); a.lbu(H2, 5, 10); a.csrc(CSR["mstatus"], H1); a.j("resume")
    a.label("svc_mprv_st"); a.li(H1, MPRV); a.csrs(CSR["mstatus"], H1); a.sd(11, 0, 10); a.sh(11, 3, 10); a.csrc(CSR["mstatus"], H1); a.j("resume")
    a.label("svc_satp"); a.csrrw(H2, CSR["satp"], 11); a.j("resume")
    a.label("svc_sfence"); a.sfence_vma(10, 11)
    a.label("resume")
    a.csrw(CSR["mstatus"], H3); a.csrw(CSR["mepc"], H0); a.mret()
    a.label("exit")
    _exit_tohost(a, TOHOST, tmp1=30, tmp2=31)

    a.label("init")
    a.la(5, "handler"); a.csrw(CSR["mtvec"], 5)
    a.li(5, 0x2000); a.csrs(CSR["mstatus"], 5)                           # FS on
    a.li(5, (1 << 54) - 1); a.csrw(0x3B0, 5); a.li(5, 0x1F); a.csrw(0x3A0, 5)   # PMP: everything, for everyone
    if seed % 8 == 7:
        a.li(5, 0xB1FF); a.csrw(CSR["medeleg"], 5)                       # delegated faults land on stvec = 0, unmapped: the run ends in the
                                                                         # consecutive-fault stop after exercising sepc / scause / stval
    asid0 = rng.randrange(4)
    a.li(5, (8 << 60) | (asid0 << 44) | (PT_ROOT >> 12)); a.csrw(CSR["satp"], 5)
    a.csrr(6, CSR["satp"])
    a.li(5, (5 << 60) | 1); a.csrw(CSR["satp"], 5); a.csrr(6, CSR["satp"])    # unsupported MODE: write dropped
    for r in range(1, 27):
        a.li(r, rng.choice([rng.getrandbits(64), rng.getrandbits(12), VA_DATA + 8 * rng.randrange(0x1000), 0]))
    a.li(5, 3 << 11); a.csrc(CSR["mstatus"], 5)
    if smode:
        a.li(5, 1 << 11); a.csrs(CSR["mstatus"], 5)                      # MPP = S
    a.li(5, VA_CODE); a.csrw(CSR["mepc"], 5); a.mret()
    assert a.here() < BASE + 0x4000

    # ---- the body, linked at its virtual address
    b = Asm(VA_CODE)
    P = 9
    amo = ["amoswap", "amoadd", "amoxor", "amoand", "amoor", "amomin", "amomax", "amominu", "amomaxu"]
    regs = [r for r in range(2, 27) if r not in (P, 10, 11, 17)]

    def pick_va():
        if high and rng.random() < 0.35:
            return rng.choice([HI + VA_DATA + rng.randrange(16) * 0x1000 + rng.choice([rng.randrange(0x1000), 0xFF8 + rng.randrange(12)]),
                               HI + VA_DATA + rng.randrange(16) * 0x1000 + 8 * rng.randrange(512), HI + VA_CODE + rng.randrange(0x4000),
                               (HI + VA_DATA + rng.randrange(0x10000)) ^ (1 << (39 + rng.randrange(25))),      # one upper bit wrong: not canonical
                               (1 << 38) + VA_DATA + rng.randrange(0x1000), 0xFFFFFFBFFFFFF000 + rng.randrange(0x1000)])
        r = rng.random()
        if r < 0.6:
            return VA_DATA + rng.randrange(16) * 0x1000 + rng.choice([rng.randrange(0x1000), 0xFF8 + rng.randrange(12), rng.randrange(16)])
        if r < 0.75:
            return VA_WIN + rng.choice([0x20000 + rng.randrange(0x10000), 0x4000, 0x4008, MEM - 8, MEM + 8, 0xA000 + 8 * rng.randrange(16)])
        if r < 0.85:
            return VA_CODE + rng.randrange(0x4000)
        return rng.choice([0x80000000 + rng.randrange(0x1000), 0xC0000000 + rng.randrange(0x1000), 0x600000 + rng.randrange(0x2000),
                           0x800000 + rng.randrange(0x1000), (1 << 38) + 0x123, 0xFFFFFFC000000000 + rng.randrange(0x1000),
                           rng.getrandbits(64), VA_DATA + 0x10000 + rng.randrange(0x1000), BASE + 0x100])

    for i in range(n):
        k = rng.random()
        rd = rng.choice(regs); rs2 = rng.choice(regs)
        if b.here() >= VA_CODE + 0x3000 - 0x80:
            break
        if k < 0.40:
            m, sz = rng.choice(_LD + _ST)
            t = pick_va()
            if rng.random() < 0.55:
                t -= t % sz
            off = rng.randrange(-64, 64)
            b.li(P, (t - off) & MASK64)
            getattr(b, m)(rd if m.startswith("l") else rs2, off, P)
        elif k < 0.50:
            sfx = rng.choice(["w", "d"]); t = pick_va()
            if rng.random() < 0.85:
                t -= t % (4 if sfx == "w" else 8)
            b.li(P, t & MASK64)
            r = rng.random()
            if r < 0.25:
                getattr(b, f"lr_{sfx}")(rd, P)
            elif r < 0.5:
                getattr(b, f"lr_{sfx}")(rd, P); getattr(b, f"sc_{sfx}")(rd, rs2, P)
            else:
                getattr(b, f"{rng.choice(amo)}_{sfx}")(rd, rs2, P)
        elif k < 0.56:
            t = pick_va(); t -= t % rng.choice([1, 4, 8])
            b.li(P, t & MASK64)
            rng.choice([b.flw, b.fld, b.fsw, b.fsd])(rng.randrange(32), 0, P)
        elif k < 0.64:                                                   # jumps: islands at the end of code page 2 / in page 3, data pages, unmapped
            t = rng.choice([VA_CODE + 0x2FF0, VA_CODE + 0x2FFA, VA_CODE + 0x3000, VA_CODE + 0x3010, VA_DATA + 0x1000 * rng.randrange(16), 0x800000, 0xC0000000])
            if high and rng.random() < 0.5:                              # the islands through their upper-half addresses (the pc itself is up there)
                t = rng.choice([HI + VA_CODE + 0x2FF0, HI + VA_CODE + 0x3010, HI + VA_CODE + 0x2FFA, (HI + VA_CODE + 0x3010) ^ (1 << 45)])
            b.li(P, t & MASK64); b.jalr(1, P, 0)
        elif k < 0.74:                                                   # page-table edit through the window, no sfence
            ix = rng.randrange(16)
            f = rng.choice([data_flags[ix], FULL | U, _PTE_V | _PTE_R | _PTE_A | U, _PTE_V | _PTE_R | _PTE_W | _PTE_A | U, 0, rng.randrange(256)])
            b.li(P, VA_WIN + (PT_L0B - BASE) + 8 * ix); b.li(rd, pte(PA_DATA + 0x1000 * rng.randrange(16), f)); b.sd(rd, 0, P)
        elif k < 0.82:
            r = rng.random()
            va = pick_va() if rng.random() < 0.8 else 0
            if r < 0.5:                                                  # sfence.vma in place (illegal from U) ...
                b.li(10, va & MASK64); b.li(11, rng.randrange(4))
                b.sfence_vma(rng.choice([0, 10]), rng.choice([0, 11]))
            else:                                                        # ... or through M
                b.li(10, va & MASK64); b.li(11, rng.randrange(4)); b.li(17, 4); b.ecall()
        elif k < 0.88:
            r = rng.random()
            if r < 0.4:
                b.li(rd, rng.choice([1 << 18, 1 << 19, 3 << 18])); rng.choice([b.csrs, b.csrc])(CSR["sstatus"], rd)
            elif r < 0.7:                                                # satp: new ASID (direct in S, illegal in U)
                b.li(rd, (8 << 60) | (rng.randrange(4) << 44) | (PT_ROOT >> 12)); b.csrw(CSR["satp"], rd)
            else:
                b.li(11, (8 << 60) | (rng.randrange(4) << 44) | (PT_ROOT >> 12)); b.li(17, 3); b.ecall()
        elif k < 0.94:
            t = pick_va(); t -= t % rng.choice([1, 8])
            b.li(10, t & MASK64); b.li(11, rng.getrandbits(64)); b.li(17, rng.choice([1, 2])); b.ecall()
        elif k < 0.96:
            b.c_nop()                                                    # shifts the following 4-byte instructions off word alignment
        else:
            getattr(b, rng.choice(_RR))(rd, rng.choice(regs), rs2)
    b.li(17, 0); b.ecall()
    b.label("hang"); b.j("hang")
    assert b.here() <= VA_CODE + 0x2FF0, "body outgrew its pages"
    b.org(VA_CODE + 0x2FF0)
    b.addi(6, 6, 1); b.addi(6, 6, 2); b.c_nop(); b.addi(6, 6, 3)
    b.ret()                                                              # at 0x2FFE: straddles code pages 2 and 3
    b.org(VA_CODE + 0x3010)
    b.addi(6, 6, 4); b.ret()
    code = b.assemble()
    text = a.assemble()
    segs = [(a.base, text, PF_R | PF_X), (d.base, d.assemble(), PF_R | PF_W), (PA_CODE, code, PF_R | (PF_X if exec_body else 0))]   # code lands inside the data image's span: last
    syms = dict(a.symbols); syms.update(d.symbols)
    elf = write_elf(segs, syms["_start"], syms)
    return Program(f"sv39{'hi' if high else ''}-{seed}", "rv64imafdcsu_zicsr", elf, syms["_start"], segs, syms, BASE, MEM, 60000, dict(seed=seed, tlb_entries=8))


# ------------------------------------------------------------------------------------------ RVV (config 3)
# funct6 of the modelled OP-V instructions
_V_INT_VVXI = {"vadd": 0x00, "vand": 0x09, "vor": 0x0A, "vxor": 0x0B, "vsll": 0x25, "vsrl": 0x28, "vsra": 0x29}   # vv vx vi
_V_INT_VVX = {"vsub": 0x02, "vminu": 0x04, "vmin": 0x05, "vmaxu": 0x06, "vmax": 0x07}                              # vv vx
_V_MUL = {"vmul": 0x25, "vmacc": 0x2D, "vnmsac": 0x2F, "vmadd": 0x29, "vnmsub": 0x2B, "vdivu": 0x20, "vdiv": 0x21, "vremu": 0x22, "vrem": 0x23,
          "vmulhu": 0x24, "vmulhsu": 0x26, "vmulh": 0x27}                              # OPMVV / OPMVX
_V_RED = {"vredsum": 0, "vredand": 1, "vredor": 2, "vredxor": 3, "vredminu": 4, "vredmin": 5, "vredmaxu": 6, "vredmax": 7}
_V_FP = {"vfadd": 0x00, "vfsub": 0x02, "vfmul": 0x24, "vfdiv": 0x20, "vfmadd": 0x28, "vfnmadd": 0x29, "vfmsub": 0x2A, "vfnmsub": 0x2B,
         "vfmacc": 0x2C, "vfnmacc": 0x2D, "vfmsac": 0x2E, "vfnmsac": 0x2F, "vfmin": 0x04, "vfmax": 0x06, "vfsgnj": 0x08, "vfsgnjn": 0x09,
         "vfsgnjx": 0x0A}                                                                                               # vv vf


def rvv_program(seed: int, n: int = 160, vlenb: int = 64) -> Program:
    """RVV (BASELINE config 3 style): strip-mined saxpy / daxpy (vle, vfmacc.vf, vse), then a
    random sequence over the modelled vector slice: vsetvli / vsetivli / vsetvl with SEW 8..64, LMUL mf8..8 (legal and
    not), ta / ma; unit-stride loads and stores (aligned and not, in and out of the arena, masked); integer vv / vx / vi
    arithmetic, multiply-add, merges and moves, vid, reductions, scalar moves; F/D arithmetic, vfmacc, ordered and unordered
    sums, FP moves; misaligned register groups, v0 overlaps and non-zero vstart (illegal), vector CSR reads and writes, a
    few unsupported encodings at the very end.  Traps go to an M handler that skips the instruction."""
    rng = random.Random(0x727676 ^ seed ^ (vlenb << 20))
    data_base = BASE + 0x4000
    a = Asm(BASE)
    d = Asm(data_base)
    d.label("tohost"); d.dword(0)
    d.label("fromhost"); d.dword(0)
    d.org(data_base + 0x100)
    d.label("xs")
    import struct
    nx = 200
    d.data(b"".join(struct.pack("<f", rng.uniform(-100, 100)) for _ in range(nx)))
    d.label("ys")
    d.data(b"".join(struct.pack("<f", rng.uniform(-100, 100)) for _ in range(nx)))
    d.align_data(8)
    d.label("xd")
    d.data(b"".join(struct.pack("<d", rng.uniform(-1e3, 1e3)) for _ in range(nx)))
    d.label("yd")
    d.data(b"".join(struct.pack("<d", rng.choice([rng.uniform(-1e3, 1e3), 1e300, -1e300, 5e-324, float("inf"), float("nan")])) for _ in range(nx)))
    d.align_data(64)
    d.label("arena")
    d.data(bytes(rng.getrandbits(8) for _ in range(0x2000)))
    S = d.symbols
    ARENA = S["arena"]

    H1, H2 = 30, 31
    a.label("_start")
    a.j("init")
    a.align(4)
    a.label("handler")
    a.csrr(H1, CSR["mepc"]); a.addi(H1, H1, 4); a.csrw(CSR["mepc"], H1); a.mret()      # every instruction of the body is 4 bytes
    a.label("init")
    a.la(5, "handler"); a.csrw(CSR["mtvec"], 5)
    if seed % 5 == 4:
        a.vsetvli(6, 0, Asm.vtype(32, 1)); a.vop(0x00, 0, 1, 2, 3)          # VS still Off: both illegal
    a.li(5, 0x2200); a.csrs(CSR["mstatus"], 5)                               # FS and VS on
    a.vop(0x00, 0, 1, 2, 3)                                                  # vtype.vill at reset: illegal
    a.csrr(6, 0xC21); a.csrr(6, 0xC22); a.csrr(6, 0xC20)

    # ---- saxpy: y[i] += a * x[i] (e32) then daxpy (e64), strip-mined
    for sew, xs, ys, ld in ((32, "xs", "ys", a.flw), (64, "xd", "yd", a.fld)):
        lm = rng.choice([1, 2, 4, 8])
        a.li(10, rng.randrange(60, nx)); a.li(11, S[xs]); a.li(12, S[ys]); a.li(13, S[xs] + 4 * (sew // 32)); ld(1, 0, 13)
        top = a.here()
        a.vsetvli(14, 10, Asm.vtype(sew, lm, ta=bool(rng.getrandbits(1)), ma=bool(rng.getrandbits(1))))
        a.vle(sew, 8, 11); a.vle(sew, 16, 12)
        a.vop(_V_FP["vfmacc"], 5, 16, 8, 1)                                  # vfmacc.vf v16, f1, v8
        a.vse(sew, 16, 12)
        a.slli(15, 14, sew // 32 + 1); a.add(11, 11, 15); a.add(12, 12, 15); a.sub(10, 10, 14)
        a.bnez(10, top)
        # ordered and unordered sums of what was just computed
        a.vsetvli(14, 0, Asm.vtype(sew, lm, ta=True)); a.li(12, S[ys]); a.vle(sew, 16, 12)
        a.vop(0x17, 4, 24, 0, 0)                                             # vmv.v.x v24, x0
        a.vop(0x03, 1, 25, 16, 24); a.vop(0x01, 1, 26, 16, 24)               # vfredosum / vfredusum
        a.vop(0x10, 1, 2, 25, 0)                                             # vfmv.f.s f2, v25

    regs = [r for r in range(5, 28)]
    P = 9
    for r in regs:
        a.li(r, rng.choice([rng.getrandbits(64), rng.getrandbits(8), ARENA + 8 * rng.randrange(0x300), 0, MASK64]))
    cur = dict(sew=8, lmul=1)

    def vreg(group_aligned=True):
        lm = cur["lmul"] if isinstance(cur["lmul"], int) else 1
        if group_aligned and rng.random() < 0.9:
            return rng.randrange(0, 32 // lm) * lm
        return rng.randrange(32)

    for i in range(n):
        k = rng.random()
        vm = 1 if rng.random() < 0.7 else 0
        if k < 0.14:
            sew = rng.choice([8, 16, 32, 64]); lmul = rng.choice([1, 1, 2, 4, 8, "mf2", "mf4", "mf8"])
            cur.update(sew=sew, lmul=lmul)
            vt = Asm.vtype(sew, lmul, ta=bool(rng.getrandbits(1)), ma=bool(rng.getrandbits(1)))
            r = rng.random()
            if r < 0.5:
                a.li(P, rng.choice([rng.randrange(0, 70), rng.randrange(0, 600), MASK64])); a.vsetvli(rng.choice([0, 6]), rng.choice([0, P, P]), vt)
            elif r < 0.8:
                a.vsetivli(6, rng.randrange(32), vt)
            else:
                a.li(P, rng.randrange(0, 300)); a.li(7, rng.choice([vt, vt, vt | 0x100, 4, 1 << 63, rng.getrandbits(8)])); a.vsetvl(6, rng.choice([0, P]), 7)
        elif k < 0.32:
            eew = rng.choice([8, 16, 32, 64, cur["sew"], cur["sew"]])
            t = rng.choice([ARENA + rng.randrange(0x1800), ARENA + 8 * rng.randrange(0x300), ARENA + 0x2000 - rng.randrange(64), BASE + (1 << 16) - rng.randrange(1, 40),
                            0x1000, ARENA + 64 * rng.randrange(0x40)])
            if rng.random() < 0.01:
                t = S["tohost"] - rng.randrange(1, 30)                       # a vector store that runs into tohost (may end the run)
            if rng.random() < 0.7:
                t -= t % (eew // 8)
            a.li(P, t)
            r = rng.random()
            if r < 0.3:
                a.vle(eew, vreg(), P, vm)
            elif r < 0.6:
                a.vse(eew, vreg(), P, vm)
            elif r < 0.7:                                                    # whole registers: vl<nf>re<eew>.v / vs<nf>r.v (also illegal nf / alignment / masked)
                nf = rng.choice([1, 1, 2, 2, 4, 8, 3]); st = rng.random() < 0.5
                v = rng.randrange(0, 32 // nf) * nf if rng.random() < 0.9 else rng.randrange(32)
                a.vls(st, 8 if st and rng.random() < 0.9 else eew, v, P, lumop=8, vm=rng.choice([1] * 9 + [0]), nf=nf - 1, mew=rng.choice([0] * 19 + [1]))
            elif r < 0.8 and rng.random() < 0.6:                             # indexed: offsets built with vid * k (or random register contents), index EEW 8..64
                ieew = rng.choice([8, 16, 32, 64, cur["sew"]])
                vi = vreg()
                if rng.random() < 0.8:
                    a.vop(0x14, 2, vi, 0, 0x11, 1)                           # vid.v vi
                    a.vop(0x25, 3, vi, vi, rng.choice([0, 1, 2, 3]), 1)      # vsll.vi vi, vi, k
                a.li(P, ARENA + 8 * rng.randrange(0x100))
                a.vls(rng.random() < 0.5, ieew, vreg(), P, lumop=vi, vm=vm, mop=rng.choice([1, 3]), nf=rng.choice([0, 0, 0, 1, 2, 3]))
            elif r < 0.8 and rng.random() < 0.5:                             # segments (unit-stride / strided) and fault-only-first loads
                nf = rng.choice([2, 2, 3, 4, 8])
                v = rng.randrange(0, 32)
                q = rng.random()
                if q < 0.45:
                    a.vls(rng.random() < 0.5, eew, v, P, lumop=0, vm=vm, nf=nf - 1)
                elif q < 0.7:
                    a.li(7, rng.choice([nf * (eew // 8), 64, 0, 3, -(eew // 8)]) & MASK64)
                    a.vls(rng.random() < 0.5, eew, v, P, lumop=7, vm=vm, nf=nf - 1, mop=2)
                else:
                    if rng.random() < 0.6:
                        a.li(P, BASE + (1 << 16) - rng.randrange(1, 200))    # runs off the end of memory: vl is trimmed
                    a.vls(False, eew, v, P, lumop=0x10, vm=vm, nf=rng.choice([0, 0, 0, nf - 1]))
            elif r < 0.76:                                                   # vlm.v / vsm.v
                a.vls(rng.random() < 0.5, 8, rng.choice([0, vreg(False)]), P, lumop=0xB, vm=rng.choice([1] * 9 + [0]))
            else:                                                            # strided: positive, zero, negative and unaligned strides
                a.li(7, rng.choice([eew // 8, 2 * (eew // 8), 24, 0, -(eew // 8), 3, 0x400]) & MASK64)
                (a.vlse if rng.random() < 0.5 else a.vsse)(eew, vreg(), P, 7, vm)
        elif k < 0.38:                                                   # whole-register moves, mask scans, extensions
            q = rng.random()
            if q < 0.2:
                nr = rng.choice([1, 2, 4, 8, 3, 5])
                al = (lambda: rng.randrange(0, 32 // nr) * nr) if nr in (1, 2, 4, 8) and rng.random() < 0.9 else (lambda: rng.randrange(32))
                a.vop(0x27, 3, al(), al(), nr - 1, rng.choice([1] * 9 + [0]))                 # vmv<nr>r.v
            elif q < 0.4:
                a.vop(0x10, 2, rng.choice(regs), vreg(False), rng.choice([0x10, 0x11, 0x10, 0x11, 0x12]), vm)   # vcpop.m / vfirst.m
            elif q < 0.6:
                a.vop(0x14, 2, rng.choice([0, vreg(False), vreg(False)]), vreg(False), rng.choice([1, 2, 3, 1, 2, 3, 4]), vm)   # vmsbf / vmsof / vmsif
            elif q < 0.75:
                a.vop(0x14, 2, vreg(), vreg(False), 0x10, vm)             # viota.m
            elif q < 0.85:
                a.vop(0x12, 2, vreg(), vreg(False) if rng.random() < 0.7 else vreg(), rng.randrange(2, 8), vm)   # vzext / vsext .vf8 .vf4 .vf2
            elif q < 0.90:                                                   # vrgather .vv / .vx / .vi
                f3 = rng.choice([0, 4, 3])
                if f3 == 4:
                    a.li(7, rng.choice([0, 1, 5, 17, 63, 200, 600, MASK64]))
                a.vop(0x0C, f3, vreg(), vreg(), vreg() if f3 == 0 else (7 if f3 == 4 else rng.randrange(32)), vm)
            elif q < 0.94:                                                   # vrgatherei16.vv, vcompress.vm, vwredsum[u].vs
                t = rng.random()
                if t < 0.4:
                    a.vop(0x0E, 0, vreg(), vreg(), vreg(), vm)
                elif t < 0.7:
                    a.vop(0x17, 2, vreg(), vreg(), rng.choice([0, vreg(False)]), rng.choice([1] * 9 + [0]))
                else:
                    a.vop(rng.choice([0x30, 0x31]), 0, vreg(False), vreg(), vreg(False), vm)
            else:                                                            # vslideup / vslidedown .vx / .vi, vslide1up / vslide1down .vx
                f3 = rng.choice([4, 3, 6])
                if f3 == 4:
                    a.li(7, rng.choice([0, 1, 2, 7, 16, 100, 3000, MASK64]))
                a.vop(rng.choice([0x0E, 0x0F]), f3, vreg(), vreg(), 7 if f3 == 4 else (rng.choice(regs) if f3 == 6 else rng.randrange(32)), vm)
        elif k < 0.60:
            name = rng.choice(list(_V_INT_VVXI) + list(_V_INT_VVX) + ["vrsub"])
            if name == "vrsub":
                f6, forms = 0x03, [4, 3]
            elif name in _V_INT_VVXI:
                f6, forms = _V_INT_VVXI[name], [0, 4, 3]
            else:
                f6, forms = _V_INT_VVX[name], [0, 4]
            f3 = rng.choice(forms)
            a.vop(f6, f3, vreg(), vreg(), vreg() if f3 == 0 else (rng.choice(regs) if f3 == 4 else rng.randrange(32)), vm)
        elif k < 0.62:                                                   # fixed point: vxrm set now and then, saturating / averaging / scaling ops
            if rng.random() < 0.25:
                a.li(7, rng.randrange(4)); a.csrw(0x00A, 7)
            q = rng.random()
            if q < 0.35:
                f6 = rng.choice([0x20, 0x21, 0x22, 0x23, 0x27, 0x2A, 0x2B]); forms = [0, 4] if f6 in (0x22, 0x23, 0x27) else [0, 4, 3]
                f3 = rng.choice(forms)
                a.vop(f6, f3, vreg(), vreg(), vreg() if f3 == 0 else (rng.choice(regs) if f3 == 4 else rng.randrange(32)), vm)
            elif q < 0.6:
                f3 = rng.choice([2, 6]); a.vop(rng.choice([0x08, 0x09, 0x0A, 0x0B]), f3, vreg(), vreg(), vreg() if f3 == 2 else rng.choice(regs), vm)
            elif q < 0.7:                                                    # vadc / vmadc / vsbc / vmsbc
                f6 = rng.choice([0x10, 0x11, 0x12, 0x13]); f3 = rng.choice([0, 4] if f6 >= 0x12 else [0, 4, 3])
                vmc = 0 if f6 in (0x10, 0x12) and rng.random() < 0.9 else rng.choice([0, 1])
                a.vop(f6, f3, vreg() if f6 in (0x10, 0x12) else rng.choice([vreg(False), vreg(False), 0]), vreg(), vreg() if f3 == 0 else (rng.choice(regs) if f3 == 4 else rng.randrange(32)), vmc)
            elif q < 0.8:                                                    # same-register operands reach the min * min and x - x corners
                v = vreg(); a.vop(rng.choice([0x27, 0x21, 0x23, 0x20]), 0, vreg(), v, v, vm)
            else:
                lm = cur["lmul"] if isinstance(cur["lmul"], int) else 1
                vsw = rng.randrange(0, 32 // min(2 * lm, 16)) * min(2 * lm, 16)
                f3 = rng.choice([0, 4, 3])
                a.vop(rng.choice([0x2E, 0x2F]), f3, vreg(), vsw, vreg() if f3 == 0 else (rng.choice(regs) if f3 == 4 else rng.randrange(32)), vm)
        elif k < 0.66:
            name = rng.choice(list(_V_MUL)); f3 = rng.choice([2, 6])
            a.vop(_V_MUL[name], f3, vreg(), vreg(), vreg() if f3 == 2 else rng.choice(regs), vm)
        elif k < 0.70:
            r = rng.random()
            if r < 0.6:                                                     # compares into a mask register (often v0), then used as masks
                f6 = rng.choice([0x18, 0x19, 0x1A, 0x1B, 0x1C, 0x1D, 0x1E, 0x1F])
                forms = [0, 4, 3] if f6 in (0x18, 0x19, 0x1C, 0x1D) else [0, 4] if f6 in (0x1A, 0x1B) else [4, 3]
                f3 = rng.choice(forms)
                a.vop(f6, f3, rng.choice([0, 0, vreg(False)]), vreg(), vreg() if f3 == 0 else (rng.choice(regs) if f3 == 4 else rng.randrange(32)), vm)
            elif r < 0.8:
                a.vop(rng.choice(range(0x18, 0x20)), 2, rng.choice([0, vreg(False)]), vreg(False), vreg(False), rng.choice([1, 1, 1, 0]))   # vmand.mm ...
            else:                                                            # widening add / sub / mul: destination group twice as wide
                lm = cur["lmul"] if isinstance(cur["lmul"], int) else 1
                vdw = rng.randrange(0, 32 // min(2 * lm, 16)) * min(2 * lm, 16) if rng.random() < 0.85 else rng.randrange(32)
                f3 = rng.choice([2, 6])
                if rng.random() < 0.7:
                    f6 = rng.choice([0x30, 0x31, 0x32, 0x33, 0x34, 0x35, 0x36, 0x37, 0x38, 0x3A, 0x3B, 0x3C, 0x3D, 0x3E, 0x3F])
                    f3 = 6 if f6 == 0x3E else f3                             # vwmaccus exists as .vx only
                    a.vop(f6, f3, vdw % 32, vdw % 32 if 0x34 <= f6 <= 0x37 and rng.random() < 0.6 else vreg(), vreg() if f3 == 2 else rng.choice(regs), vm)
                else:                                                        # narrowing shifts: the source group is twice as wide
                    f3 = rng.choice([0, 4, 3])
                    a.vop(rng.choice([0x2C, 0x2D]), f3, vreg(), vdw % 32, vreg() if f3 == 0 else (rng.choice(regs) if f3 == 4 else rng.randrange(32)), vm)
        elif k < 0.77:
            r = rng.random()
            if r < 0.35:
                f3 = rng.choice([0, 4, 3]); a.vop(0x17, f3, vreg(), vreg(), vreg() if f3 == 0 else (rng.choice(regs) if f3 == 4 else rng.randrange(32)), 0)   # vmerge
            elif r < 0.7:
                f3 = rng.choice([0, 4, 3]); a.vop(0x17, f3, vreg(), rng.choice([0, 0, 0, 3]), vreg() if f3 == 0 else (rng.choice(regs) if f3 == 4 else rng.randrange(32)), 1)   # vmv.v.*
            elif r < 0.8:
                a.vop(0x14, 2, vreg(), rng.choice([0, 0, 0, 2]), 0x11, vm)    # vid.v
            elif r < 0.9:
                a.vop(0x10, 2, rng.choice(regs), vreg(False), 0, rng.choice([1, 1, 1, 0]))      # vmv.x.s
            else:
                a.vop(0x10, 6, vreg(False), rng.choice([0, 0, 0, 1]), rng.choice(regs), rng.choice([1, 1, 1, 0]))   # vmv.s.x
        elif k < 0.82:
            a.vop(_V_RED[rng.choice(list(_V_RED))], 2, vreg(False), vreg(), vreg(False), vm)
        elif k < 0.93:
            r = rng.random()
            if r < 0.35:
                name = rng.choice(list(_V_FP)); f3 = rng.choice([1, 5])
                a.vop(_V_FP[name], f3, vreg(), vreg(), vreg() if f3 == 1 else rng.randrange(32), vm)
            elif r < 0.44:                                                   # vfsqrt.v, same-width vfcvt.*, vfmerge.vfm
                q = rng.random()
                if q < 0.3:
                    a.vop(0x13, 1, vreg(), vreg(), rng.choice([0, 0, 4, 5]), vm)
                elif q < 0.8:
                    a.vop(0x12, 1, vreg(), vreg(), rng.choice([0, 1, 2, 3, 6, 7]), vm)
                else:
                    a.vop(0x17, 5, vreg(), vreg(), rng.randrange(32), 0)
            elif r < 0.60:                                                   # widening FP (SEW 32 -> 64), f <-> f width changes, vfclass
                lm = cur["lmul"] if isinstance(cur["lmul"], int) else 1
                vdw = (rng.randrange(0, 32 // min(2 * lm, 16)) * min(2 * lm, 16)) % 32 if rng.random() < 0.85 else rng.randrange(32)
                q = rng.random(); f3 = rng.choice([1, 5])
                if q < 0.55:
                    f6 = rng.choice([0x30, 0x32, 0x34, 0x36, 0x38, 0x3C, 0x3D, 0x3E, 0x3F])
                    a.vop(f6, f3, vdw, vdw if f6 in (0x34, 0x36) and rng.random() < 0.6 else vreg(), vreg() if f3 == 1 else rng.randrange(32), vm)
                elif q < 0.7:                                                # widening / narrowing int <-> float conversions
                    code = rng.choice([0x08, 0x09, 0x0A, 0x0B, 0x0E, 0x0F, 0x10, 0x11, 0x12, 0x13, 0x16, 0x17])
                    a.vop(0x12, 1, vdw if code < 0x10 else vreg(), vreg() if code < 0x10 else vdw, code, vm)
                elif q < 0.78:
                    a.vop(0x12, 1, vdw, vreg(), 0x0C, vm)
                elif q < 0.86:
                    a.vop(0x12, 1, vreg(), vdw, rng.choice([0x14, 0x14, 0x15]), vm)
                elif q < 0.94:
                    a.vop(rng.choice([0x31, 0x33]), 1, vreg(False), vreg(), vreg(False), vm)
                else:
                    a.vop(0x13, 1, vreg(), vreg(), 0x10, vm)
            elif r < 0.62:                                                   # vmfeq vmfle vmflt vmfne (.vv / .vf), vmfgt vmfge (.vf)
                f6 = rng.choice([0x18, 0x19, 0x1B, 0x1C, 0x1D, 0x1F]); f3 = 5 if f6 in (0x1D, 0x1F) else rng.choice([1, 5])
                a.vop(f6, f3, rng.choice([0, 0, vreg(False)]), vreg(), vreg() if f3 == 1 else rng.randrange(32), vm)
            elif r < 0.7:
                a.vop(rng.choice([0x27, 0x21, 0x0E, 0x0F]), 5, vreg(), vreg(), rng.randrange(32), vm)   # vfrsub / vfrdiv / vfslide1up / vfslide1down
            elif r < 0.8:
                a.vop(rng.choice([0x01, 0x03, 0x05, 0x07]), 1, vreg(False), vreg(), vreg(False), vm)   # vfredusum / vfredosum / vfredmin / vfredmax
            elif r < 0.87:
                a.vop(0x17, 5, vreg(), 0, rng.randrange(32), 1)                                   # vfmv.v.f
            elif r < 0.94:
                a.vop(0x10, 1, rng.randrange(32), vreg(False), 0, 1)                              # vfmv.f.s
            else:
                a.vop(0x10, 5, vreg(False), 0, rng.randrange(32), 1)                              # vfmv.s.f
        elif k < 0.97:
            r = rng.random()
            if r < 0.3:
                a.li(P, rng.randrange(0, 40)); a.csrw(0x008, P)                                   # vstart: the next arithmetic op traps, a load / store resumes
            elif r < 0.5:
                a.li(P, rng.getrandbits(3)); a.csrw(rng.choice([0x009, 0x00A, 0x00F]), P)
            elif r < 0.7:
                a.li(P, rng.choice([0, 1, 2, 3, 4, 5]) << 5); a.csrw(CSR["fcsr"], P)              # rounding mode (5: invalid -> FP vector ops trap)
            else:
                a.csrr(rng.choice(regs), rng.choice([0x008, 0x009, 0x00A, 0x00F, 0xC20, 0xC21, 0xC22, CSR["mstatus"], CSR["fcsr"]]))
        else:
            getattr(a, rng.choice(_RR))(rng.choice(regs), rng.choice(regs), rng.choice(regs))
    if seed % 7 == 6:                                                        # an encoding outside the modelled slice: UNSUPPORTED stop on both sides
        a.vsetivli(6, 4, Asm.vtype(32, 1)); a.vop(0x12, 2, 4, 8, 8)          # vbrev8.v (Zvbb)
    _exit_tohost(a, S["tohost"], tmp1=30, tmp2=31)
    assert a.here() < data_base
    return _finish(a, f"rvv-{seed}-{vlenb}", "rv64imafdcv_zicsr", d, mem_size=1 << 16, max_inst=40000, meta=dict(seed=seed, vlenb=vlenb))


def rvv_kernel(n: int = 1024, iters: int = 2, vlenb: int = 64) -> Program:
    """BASELINE config 3 workload ("RVV vector kernels at VLEN 256 / 512, 16k harts"): the strip-mined loops of a vector
    library, per-instance data from the seed in a0 (xorshift-filled arrays).  Per pass: saxpy (e32, vfmacc.vf), daxpy (e64,
    vfmadd.vv with a splat), an integer pipeline (e32: vmul.vx, vadd.vv, vsrl.vi, vxor.vv, vmax.vv; e8: vadd / vand / vsub at
    LMUL 8), a masked update (vmslt.vx -> vmerge / masked vadd), a dot product (vfmul + vfredusum) and an integer checksum
    (vredsum).  Results go to `results`; exit by writing 1 to tohost.  Almost every dynamic instruction of the loops is a
    vector instruction or its scalar strip-mining companion (slli / add / sub / bnez)."""
    import struct
    data_base = BASE + 0x1000
    a = Asm(BASE)
    d = Asm(data_base)
    d.label("tohost"); d.dword(0)
    d.label("fromhost"); d.dword(0)
    d.label("seed"); d.dword(0)
    d.label("results"); d.zeros(8 * 16)
    d.label("fconst"); d.data(struct.pack("<f", 1.25)); d.data(struct.pack("<f", 0.5)); d.data(struct.pack("<d", 0.75)); d.data(struct.pack("<d", 1.0 / 1024))
    d.align_data(64)
    d.label("xs"); d.zeros(4 * n)
    d.label("ys"); d.zeros(4 * n)
    d.label("xd"); d.zeros(8 * n)
    d.label("yd"); d.zeros(8 * n)
    d.label("bi"); d.zeros(4 * n)
    d.label("bb"); d.zeros(4 * n)
    S = d.symbols
    SEED, RES, N, PX, PY, VL, T0, T1, T2, ITER = 10, 11, 12, 13, 14, 15, 5, 6, 7, 18

    def strip(sew, lmul, body, px=None, py=None, ta=True, ma=True):
        """for (n = N; n; n -= vl) { vl = vsetvli(n); body(); px += vl * sew / 8; py += ... }"""
        a.li(N, n)
        if px is not None: a.li(PX, S[px])
        if py is not None: a.li(PY, S[py])
        top = a.here()
        a.vsetvli(VL, N, Asm.vtype(sew, lmul, ta=ta, ma=ma))
        body()
        a.slli(T0, VL, {8: 0, 16: 1, 32: 2, 64: 3}[sew])
        if px is not None: a.add(PX, PX, T0)
        if py is not None: a.add(PY, PY, T0)
        a.sub(N, N, VL)
        a.bnez(N, top)

    a.label("_start")
    a.li(T0, 0x6600); a.csrs(CSR["mstatus"], T0)                               # FS and VS on
    a.li(RES, S["results"])
    a.li(T0, S["seed"]); a.ld(T1, 0, T0); a.xor(SEED, SEED, T1); a.ori(SEED, SEED, 1)
    # fill the integer arrays (and through them the FP ones) with a hash of (element index, seed): vid.v, multiply / shift / xor
    a.li(T1, 0x9E3779B1); a.li(T2, 0)
    def fill():
        a.vop(0x14, 2, 8, 0, 0x11)                                             # vid.v v8
        a.vop(0x00, 4, 8, 8, T2)                                               # vadd.vx v8, v8, t2 (index of the strip's first element)
        a.vop(_V_MUL["vmul"], 6, 8, 8, SEED)                                   # vmul.vx v8, v8, seed
        a.vop(0x28, 3, 16, 8, 15); a.vop(0x0B, 0, 8, 8, 16)                    # v8 ^= v8 >> 15
        a.vop(_V_MUL["vmul"], 6, 8, 8, T1)                                     # vmul.vx v8, v8, golden
        a.vop(0x28, 3, 16, 8, 13); a.vop(0x0B, 0, 8, 8, 16)                    # v8 ^= v8 >> 13
        a.vse(32, 8, PX); a.add(T2, T2, VL)
    strip(32, 8, fill, "bi"); strip(32, 8, fill, "bb")
    a.li(T0, S["fconst"]); a.flw(1, 0, T0); a.flw(2, 4, T0); a.fld(3, 8, T0); a.fld(4, 16, T0)
    # xs / ys = float(int32 >> 8), xd / yd = double(int32) / 1024: vector conversions
    def cvt32():
        a.vle(32, 8, PX); a.vop(0x29, 3, 8, 8, 8)                              # vsra.vi v8, v8, 8
        a.vop(0x12, 1, 16, 8, 3)                                               # vfcvt.f.x.v v16, v8
        a.vse(32, 16, PY)
    strip(32, 8, cvt32, "bi", "xs"); strip(32, 8, cvt32, "bb", "ys")
    def cvt64(src):
        def f():
            a.vle(32, 8, PX); a.vop(0x12, 1, 16, 8, 0x0B)                      # vfwcvt.f.x.v v16, v8 (e32 -> e64)
            a.vsetvli(0, 0, Asm.vtype(64, 8, ta=True, ma=True))                # same vl at e64, LMUL 8
            a.vop(0x24, 5, 16, 16, 4)                                          # vfmul.vf v16, v16, f4
            a.vse(64, 16, PY)
            a.slli(T0, VL, 2); a.add(PY, PY, T0)                               # py advances by 8 bytes per element: the other half here
        return f
    strip(32, 4, cvt64("bi"), "bi", "xd"); strip(32, 4, cvt64("bb"), "bb", "yd")
    a.li(ITER, iters)
    a.label("outer")
    # ---- saxpy: ys += f1 * xs
    def saxpy():
        a.vle(32, 8, PX); a.vle(32, 16, PY); a.vop(_V_FP["vfmacc"], 5, 16, 8, 1); a.vse(32, 16, PY)
    strip(32, 8, saxpy, "xs", "ys")
    # ---- daxpy with a splat: yd = f3 * xd + yd
    def daxpy():
        a.vle(64, 8, PX); a.vle(64, 16, PY); a.vop(0x17, 5, 24, 0, 3)          # vfmv.v.f v24, f3
        a.vop(_V_FP["vfmadd"], 1, 8, 16, 24)                                   # vfmadd.vv v8, v24, v16: v8 = v8 * v24 + v16
        a.vse(64, 8, PY)
    strip(64, 8, daxpy, "xd", "yd")
    # ---- integer pipeline, e32
    def ipipe():
        a.vle(32, 8, PX); a.vle(32, 16, PY)
        a.vop(_V_MUL["vmul"], 6, 24, 8, SEED)                                  # vmul.vx v24, v8, seed
        a.vop(0x00, 0, 24, 24, 16)                                             # vadd.vv
        a.vop(0x28, 3, 8, 24, 7)                                               # vsrl.vi v8, v24, 7
        a.vop(0x0B, 0, 24, 24, 8)                                              # vxor.vv
        a.vop(0x07, 0, 16, 24, 16)                                             # vmax.vv v16, v24, v16
        a.vse(32, 16, PY)
    strip(32, 8, ipipe, "bi", "bb")
    # ---- bytes at LMUL 8
    def bytes8():
        a.vle(8, 8, PX); a.vle(8, 16, PY); a.vop(0x00, 0, 24, 8, 16); a.vop(0x09, 4, 24, 24, SEED); a.vop(0x02, 0, 16, 24, 8); a.vse(8, 16, PY)
    a.li(N, 4 * n); a.li(PX, S["bi"]); a.li(PY, S["bb"])
    top = a.here()
    a.vsetvli(VL, N, Asm.vtype(8, 8, ta=True, ma=False)); bytes8(); a.add(PX, PX, VL); a.add(PY, PY, VL); a.sub(N, N, VL); a.bnez(N, top)
    # ---- masked update: where bi < 0: bb = bb + bi (masked, mask undisturbed), then merge with a scalar
    def masked():
        a.vle(32, 8, PX); a.vle(32, 16, PY)
        a.vop(0x1B, 4, 0, 8, 0)                                                # vmslt.vx v0, v8, x0
        a.vop(0x00, 0, 16, 16, 8, 0)                                           # vadd.vv v16, v16, v8, v0.t
        a.vop(0x17, 4, 16, 16, ITER, 0)                                        # vmerge.vxm v16, v16, iter, v0
        a.vse(32, 16, PY)
    strip(32, 8, masked, "bi", "bb", ta=False, ma=False)
    # ---- dot product xs . ys (vfmul + unordered sum per strip, accumulated in f5) and integer checksum of bb
    a.fmv_w_x(5, 0)
    def dot():
        a.vle(32, 8, PX); a.vle(32, 16, PY); a.vop(_V_FP["vfmul"], 1, 8, 8, 16)
        a.vop(0x10, 5, 24, 0, 5)                                               # vfmv.s.f v24, f5
        a.vop(0x01, 1, 24, 8, 24)                                              # vfredusum.vs v24, v8, v24
        a.vop(0x10, 1, 5, 24, 0)                                               # vfmv.f.s f5, v24
    strip(32, 8, dot, "xs", "ys")
    a.fsw(5, 0, RES)
    a.li(T2, 0)
    def isum():
        a.vle(32, 8, PX); a.vop(0x10, 6, 24, 0, T2)                            # vmv.s.x v24, t2
        a.vop(0x00, 2, 24, 8, 24)                                              # vredsum.vs v24, v8, v24
        a.vop(0x10, 2, T2, 24, 0)                                              # vmv.x.s t2, v24
    strip(32, 8, isum, "bb")
    a.sd(T2, 8, RES)
    a.addi(ITER, ITER, -1)
    a.beqz(ITER, "done")
    a.j("outer")
    a.label("done")
    a.csrr(T0, CSR["fcsr"]); a.sd(T0, 16, RES)
    a.csrr(T0, CSR["minstret"]); a.sd(T0, 24, RES)
    _exit_tohost(a, S["tohost"])
    assert a.here() < data_base
    size = 1 << 16
    while data_base + 0x100 + 32 * n > BASE + size:
        size <<= 1
    return _finish(a, f"rvvk-{n}-{vlenb}", "rv64imafdcv_zicsr", d, mem_size=size, max_inst=1 << 26, meta=dict(n=n, iters=iters, vlenb=vlenb))


# ------------------------------------------------------------------------------------------ RV32 (BASELINE config 0)
def readme_rv32_program() -> Program:
    """BASELINE config 0: the README.md:167-185 test program (x = 1; y = 2; z = x + y; tohost = 1) as riscv32 gcc -nostdlib
    would emit it for rv32imc: an ELF32 file of c.* forms, a call into main (c.jal, the RV32-only code point) and a store of 1
    to tohost."""
    data_base = BASE + 0x1000
    a = Asm(BASE, xlen=32)
    a.label("_start")
    a.li(2, BASE + 0x3000)                  # sp
    a.half(0x2000 | (((8 >> 1) & 7) << 3) | (((8 >> 4) & 1) << 11) | (((8 >> 5) & 1) << 2) | 1)   # c.jal +8 (main): quadrant 1, funct3 001 on RV32
    a.pc  # noqa: B018
    a.label("after_call")
    a.j("exit")
    a.align(4)
    a.label("main")
    a.c_li(10, 1)
    a.c_li(11, 2)
    a.c_add(10, 11)
    a.c_jr(1)
    a.align(4)
    a.label("exit")
    a.li(5, data_base)
    a.c_li(6, 1)
    a.sw(6, 0, 5)
    a.jal(0, a.pc)
    d = Asm(data_base, xlen=32)
    d.label("tohost")
    d.dword(0)
    d.label("fromhost")
    d.dword(0)
    return _finish(a, "readme-rv32", "rv32imc_zicsr", d, mem_size=1 << 14, max_inst=1000)


def rv32_program(seed: int, n: int = 300, fp: bool = False) -> Program:
    """Random RV32 code (M mode, bare): the integer ops whose results differ between the two widths (add / sub / shifts / mul*
    / div* / slt* on values with bit 31 set), loads / stores with wrapping addresses, branches, jal / jalr / auipc links,
    A (lr / sc / amo*.w), the compressed forms including the RV32-only code points (c.jal, c.flw / c.fsw and friends when F is
    on), CSR traffic over the 32-bit views (mstatus / mstatush, mcycle / mcycleh, minstret / minstreth, menvcfg / menvcfgh,
    odd pmpcfg registers, mcause after traps), RV64-only encodings that must trap (addiw, ld, sd, slli by 32, amoadd.d,
    fcvt.l.s).  A handler skips the trapping instruction."""
    rng = random.Random(0x32 * 1000003 + seed)
    data_base = BASE + 0x2000
    isa = "rv32imafdc_zicsr_zicntr" if fp else "rv32imac_zicsr_zicntr"
    a = Asm(BASE, xlen=32)
    d = Asm(data_base, xlen=32)
    d.label("tohost"); d.dword(0)
    d.label("fromhost"); d.dword(0)
    d.label("data")
    d.data(bytes(rng.getrandbits(8) for _ in range(1024)))
    S = d.symbols
    H1, H2 = 30, 31
    a.label("_start")
    a.j("init")
    a.align(4)
    a.label("handler")
    a.csrr(H1, CSR["mepc"]); a.lhu(H2, 0, H1); a.andi(H2, H2, 3); a.addi(H2, H2, -3)
    a.bnez(H2, "h2"); a.addi(H1, H1, 2)
    a.label("h2"); a.addi(H1, H1, 2); a.csrw(CSR["mepc"], H1); a.mret()
    a.label("init")
    a.la(5, "handler"); a.csrw(CSR["mtvec"], 5)
    if fp:
        a.li(5, 0x2000); a.csrs(CSR["mstatus"], 5)
        a.li(5, S["data"])
        for i in range(8):
            a.fld(i, 8 * i, 5)
    vals = [0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0x80000001, 0x55555555, 0xAAAAAAAA, 31, 32, 0xFFFF8000]
    for r in range(1, 30):
        a.li(r, rng.choice(vals + [rng.getrandbits(32), rng.getrandbits(12), S["data"] + 4 * rng.randrange(200)]))
    ptrs = [8, 9, 18, 19]
    for r in ptrs:
        a.li(r, S["data"] + 128 + 4 * rng.randrange(64))
    regs = list(range(0, 30))
    rr = ["add", "sub", "sll", "slt", "sltu", "xor", "srl", "sra", "or_", "and_", "mul", "mulh", "mulhsu", "mulhu", "div", "divu", "rem", "remu"]
    ri = ["addi", "slti", "sltiu", "xori", "ori", "andi"]
    csrs_rw = [0x340, 0x300, 0x304, 0x305, 0x341, 0x342, 0x343, 0x30A, 0x31A, 0x310, 0xB00, 0xB80, 0xB02, 0xB82, 0x3A0, 0x3A1, 0x3B0, 0x3B1, 0x306, 0x320, 0xB03, 0xB83, 0x323, 0x723, 0x747, 0x757]
    csrs_ro = [0x301, 0xF14, 0xC00, 0xC80, 0xC01, 0xC81, 0xC02, 0xC82, 0xF11, 0x344]
    labels = 0
    for i in range(n):
        k = rng.random()
        rd, r1, r2 = rng.choice(regs), rng.choice(regs), rng.choice(regs)
        if rd in ptrs:
            rd = 5
        if k < 0.30:
            getattr(a, rng.choice(rr))(rd, r1, r2)
        elif k < 0.42:
            getattr(a, rng.choice(ri))(rd, r1, rng.randrange(-2048, 2048))
        elif k < 0.50:
            getattr(a, rng.choice(["slli", "srli", "srai"]))(rd, r1, rng.randrange(32))
        elif k < 0.62:
            p = rng.choice(ptrs)
            name, sz = rng.choice([("lb", 1), ("lh", 2), ("lw", 4), ("lbu", 1), ("lhu", 2), ("sb", 1), ("sh", 2), ("sw", 4)])
            off = rng.randrange(-64, 64)
            if rng.random() < 0.7:
                off -= off % sz
            if name[0] == "l":
                getattr(a, name)(rd, off, p)
            else:
                getattr(a, name)(r2, off, p)
        elif k < 0.66:                                       # addresses that wrap at 32 bits / lie outside the arena
            a.li(5, rng.choice([0xFFFFFFFC, 0, S["data"] - 0x80000000 + 0x80000000, BASE + (1 << 14) - 2]))
            (a.lw if rng.random() < 0.5 else a.sw)(6, rng.choice([0, 4, 8, -4]), 5)
        elif k < 0.74:
            lab = f"L{labels}"; labels += 1
            getattr(a, rng.choice(_BR))(r1, r2, lab)
            for _ in range(rng.randrange(1, 4)):
                getattr(a, rng.choice(rr))(rng.choice([5, 6, 7]), rng.choice(regs), rng.choice(regs))
            a.label(lab)
        elif k < 0.78:
            lab = f"L{labels}"; labels += 1
            if rng.random() < 0.5:
                a.jal(rng.choice([0, 1, 5]), lab)
            else:
                a.la(6, lab); a.jalr(rng.choice([0, 1, 7]), 6, 0)
            a.addi(7, 7, 1)
            a.label(lab)
            a.auipc(rng.choice([5, 6, 7]), rng.randrange(1 << 20))
        elif k < 0.86:
            q = rng.random()
            if q < 0.2:
                a.c_li(rng.choice([5, 6, 7, 10, 11]), rng.randrange(-32, 32))
            elif q < 0.35:
                a.c_addi(rng.choice([5, 6, 7, 10, 11]), rng.randrange(1, 32))
            elif q < 0.45:
                a.c_slli(rng.choice([5, 6, 7]), rng.randrange(1, 32))
            elif q < 0.55:
                getattr(a, rng.choice(["c_srli", "c_srai"]))(rng.choice([10, 11, 12]), rng.randrange(1, 32))
            elif q < 0.65:
                getattr(a, rng.choice(["c_sub", "c_xor", "c_or", "c_and"]))(rng.choice([10, 11, 12]), rng.choice([13, 14, 15]))
            elif q < 0.75:
                a.c_lw(rng.choice([10, 11, 12]), 8, 4 * rng.randrange(16)) if rng.random() < 0.5 else a.c_sw(rng.choice([10, 11, 12]), 9, 4 * rng.randrange(16))
            elif q < 0.82:
                a.c_add(rng.choice([5, 6, 7]), rng.choice([10, 11, 12])) if rng.random() < 0.5 else a.c_mv(rng.choice([5, 6, 7]), rng.choice([10, 11, 12]))
            elif q < 0.90:                                   # c.jal: quadrant 1, funct3 001 (c.addiw on RV64)
                a.half(0x2000 | (((4 >> 1) & 7) << 3) | 1)  # c.jal +4
                a.c_addi(7, 1)
                a.c_addi(7, 2)
            else:                                            # the RV32 code points of c.ld / c.sd / c.ldsp / c.sdsp, shamt[5], c.subw / c.addw
                a.half(rng.choice([0x6000 | (1 << 10) | (1 << 7), 0xE000 | (1 << 10) | (1 << 7) | (2 << 2), 0x6002 | (5 << 7) | (1 << 4),
                                   0xE002 | (6 << 2) | (1 << 9), 0x1002 | (5 << 7) | (3 << 2), 0x9C01 | (2 << 7) | (3 << 2), 0x9C21 | (2 << 7) | (3 << 2),
                                   0x9001 | (2 << 7) | (3 << 2)]))
        elif k < 0.93:
            q = rng.random()
            if q < 0.45:
                n_ = rng.choice(csrs_rw)
                op = rng.choice([a.csrrw, a.csrrs, a.csrrc])
                src = rng.choice(regs)
                if n_ == 0x300:
                    a.li(5, rng.choice([0x1800, 0x1888, 0x0088, 0x80001800, 0x6000 if fp else 0x1800, 0x1808])); src = 5
                    op = a.csrrw
                if n_ == 0x305:
                    src = 0; op = a.csrrs                    # mtvec stays the handler
                if n_ in (0x3A0, 0x3A1):
                    a.li(5, rng.choice([0, 0x0F, 0x1F00, 0x00180F00, 0x1F1F1F1F & ~0x80808080])); src = 5
                op(rng.choice(regs[:29]), n_, src)
            elif q < 0.8:
                a.csrrs(rng.choice(regs[:29]), rng.choice(csrs_ro + csrs_rw), 0)
            else:
                a.csrrw(rng.choice(regs[:29]), rng.choice(csrs_ro), rng.choice(regs))          # read-only: illegal
        elif k < 0.97:
            q = rng.random()
            p = rng.choice(ptrs)
            if q < 0.5:
                getattr(a, rng.choice(["amoadd_w", "amoswap_w", "amoxor_w", "amoand_w", "amoor_w", "amomin_w", "amomax_w", "amominu_w", "amomaxu_w"]))(rd, r2, p)
            elif q < 0.8:
                a.lr_w(rd, p); a.sc_w(rng.choice([5, 6, 7]), r2, p)
            else:
                a.word(rng.choice([0x0000302F | (p << 15) | (6 << 20) | (5 << 7), 0x1000302F | (p << 15) | (5 << 7)]))   # amoadd.d / lr.d: RV64 only
        else:
            a.word(rng.choice([0x0010029B, 0x0002B283, 0x0052B023, 0x02029293, 0x4202D293, 0x005302BB, 0x0252C2BB, 0x00000073, 0x00100073,
                               0xC0201053 | (1 << 20) * 2, 0xE2000053, 0xF2000053, 0x0062E283, rng.getrandbits(32) | 3]))
    _exit_tohost32(a, S["tohost"])
    assert a.here() < data_base
    return _finish(a, f"rv32-{seed}", isa, d, mem_size=1 << 14, max_inst=40000, meta=dict(seed=seed))


def _exit_tohost32(a: Asm, tohost_addr, value=1, tmp1=5, tmp2=6):
    a.li(tmp1, tohost_addr)
    a.li(tmp2, value)
    a.sw(tmp2, 0, tmp1)
    a.jal(0, a.pc)


# ------------------------------------------------------------------------------------------ Zba / Zbb / Zbs / Zbc / Zicond
# name -> (major opcode, funct3, funct7 or 12-bit immediate pattern, form): "r" three registers, "u" unary (imm12 fixed),
# "s" shift-immediate (imm12 = pattern | shamt), "w5" 5-bit shift in the rs2 field
ZB_ENC = {
    "sh1add": (0x33, 2, 0x10, "r"), "sh2add": (0x33, 4, 0x10, "r"), "sh3add": (0x33, 6, 0x10, "r"),
    "add.uw": (0x3B, 0, 0x04, "r"), "sh1add.uw": (0x3B, 2, 0x10, "r"), "sh2add.uw": (0x3B, 4, 0x10, "r"), "sh3add.uw": (0x3B, 6, 0x10, "r"),
    "slli.uw": (0x1B, 1, 0x080, "s"),
    "andn": (0x33, 7, 0x20, "r"), "orn": (0x33, 6, 0x20, "r"), "xnor": (0x33, 4, 0x20, "r"),
    "clz": (0x13, 1, 0x600, "u"), "ctz": (0x13, 1, 0x601, "u"), "cpop": (0x13, 1, 0x602, "u"),
    "clzw": (0x1B, 1, 0x600, "u"), "ctzw": (0x1B, 1, 0x601, "u"), "cpopw": (0x1B, 1, 0x602, "u"),
    "max": (0x33, 6, 0x05, "r"), "maxu": (0x33, 7, 0x05, "r"), "min": (0x33, 4, 0x05, "r"), "minu": (0x33, 5, 0x05, "r"),
    "sext.b": (0x13, 1, 0x604, "u"), "sext.h": (0x13, 1, 0x605, "u"), "zext.h": (0x3B, 4, 0x080, "u"), "zext.h32": (0x33, 4, 0x080, "u"),
    "rol": (0x33, 1, 0x30, "r"), "ror": (0x33, 5, 0x30, "r"), "rori": (0x13, 5, 0x600, "s"),
    "rolw": (0x3B, 1, 0x30, "r"), "rorw": (0x3B, 5, 0x30, "r"), "roriw": (0x1B, 5, 0x600, "w5"),
    "orc.b": (0x13, 5, 0x287, "u"), "rev8": (0x13, 5, 0x6B8, "u"), "rev8_32": (0x13, 5, 0x698, "u"),
    "bclr": (0x33, 1, 0x24, "r"), "bclri": (0x13, 1, 0x480, "s"), "bext": (0x33, 5, 0x24, "r"), "bexti": (0x13, 5, 0x480, "s"),
    "binv": (0x33, 1, 0x34, "r"), "binvi": (0x13, 1, 0x680, "s"), "bset": (0x33, 1, 0x14, "r"), "bseti": (0x13, 1, 0x280, "s"),
    "czero.eqz": (0x33, 5, 0x07, "r"), "czero.nez": (0x33, 7, 0x07, "r"),
    "clmul": (0x33, 1, 0x05, "r"), "clmulh": (0x33, 3, 0x05, "r"), "clmulr": (0x33, 2, 0x05, "r"),
}


def zb_word(name, rd, rs1, rs2_or_shamt=0):
    op, f3, pat, form = ZB_ENC[name]
    if form == "r":
        return Asm.enc_r(op, f3, pat, rd, rs1, rs2_or_shamt)
    imm = pat if form == "u" else pat | rs2_or_shamt
    return (imm << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op


def bitmanip_program(seed: int, n: int = 400, xlen: int = 64, ext: str = "zba_zbb_zbs_zbc_zicond") -> Program:
    """Random Zba / Zbb / Zbs / Zbc / Zicond code (bitmanip.cpp, Hart.cpp:14728-14760) on corner-heavy operands, both widths: every
    mnemonic, shift amounts beyond the register width (must trap), the RV64-only forms on RV32 and the other width's rev8 / zext.h
    code points, neighbouring encodings of Zbkb / Zbkx that stay illegal, and - with `ext` naming fewer extensions - the
    instructions of the ones that are off.  A handler skips whatever traps; results are folded into memory as they are made."""
    rng = random.Random(0xB17 * 1000003 + seed * 7 + xlen)
    data_base = BASE + 0x3000
    rv32 = xlen == 32
    isa = ("rv32imac" if rv32 else "rv64imac") + "_zicsr_" + ext
    a = Asm(BASE, xlen=xlen)
    d = Asm(data_base, xlen=xlen)
    d.label("tohost"); d.dword(0)
    d.label("fromhost"); d.dword(0)
    d.label("data")
    d.zeros(2048)
    S = d.symbols
    H1, H2, PTR = 30, 31, 29
    a.label("_start")
    a.j("init")
    a.align(4)
    a.label("handler")
    a.csrr(H1, CSR["mepc"]); a.addi(H1, H1, 4); a.csrw(CSR["mepc"], H1); a.mret()
    a.label("init")
    a.la(5, "handler"); a.csrw(CSR["mtvec"], 5)
    m = (1 << xlen) - 1
    corner = [0, 1, 2, m, m >> 1, 1 << (xlen - 1), 0x80, 0xFF, 0x8000, 0xFFFF, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0x100000000 & m,
              0x0001000100010001 & m, 0x8040201008040201 & m, 0x00FF00FF00FF00FF & m, xlen - 1, xlen, xlen + 1, 31, 32, 33, 63, 64, 65]
    regs = list(range(0, 29))
    for r in range(1, 29):
        a.li(r, rng.choice(corner) if rng.random() < 0.6 else rng.getrandbits(xlen))
    a.li(PTR, S["data"])
    names = sorted(ZB_ENC)
    slot = 0
    for i in range(n):
        rd, r1, r2 = rng.choice(regs), rng.choice(regs), rng.choice(regs)
        k = rng.random()
        if k < 0.78:
            name = rng.choice(names)
            form = ZB_ENC[name][3]
            x = r2 if form == "r" else rng.randrange(32) if form == "w5" else rng.choice([rng.randrange(xlen), rng.randrange(64), rng.randrange(128)])
            w = zb_word(name, rd, r1, x)
            if form == "u" and rng.random() < 0.05:
                w |= rng.randrange(1, 32) << 20 if name.startswith("zext") else 0           # pack / packw with rs2 != 0: Zbkb, illegal
            a.word(w)
        elif k < 0.84:                                       # neighbours: Zbkb (brev8, zip, unzip, packh), Zbkx (xperm), unknown funct7
            a.word(rng.choice([0x68705013 | (rd << 7) | (r1 << 15), 0x08F01013 | (rd << 7) | (r1 << 15), 0x08F05013 | (rd << 7) | (r1 << 15),
                               Asm.enc_r(0x33, 7, 0x04, rd, r1, r2), Asm.enc_r(0x33, 2, 0x14, rd, r1, r2), Asm.enc_r(0x33, 4, 0x14, rd, r1, r2),
                               Asm.enc_r(0x33, rng.randrange(8), rng.choice([0x02, 0x06, 0x08, 0x11, 0x21, 0x25, 0x31, 0x35, 0x3F]), rd, r1, r2),
                               (rng.choice([0x603, 0x606, 0x607, 0x284, 0x288, 0x6B9, 0x699, 0x4C0 if rv32 else 0x4FF]) << 20) | (r1 << 15) | (rng.choice([1, 5]) << 12) | (rd << 7) | 0x13]))
        elif k < 0.92:
            getattr(a, rng.choice(["add", "sub", "xor", "sll", "srl", "sra"]))(rd, r1, r2)
        else:
            src = rng.choice(regs)
            (a.sw if rv32 else a.sd)(src, (slot % 256) * 8, PTR); slot += 1
    for r in range(1, 29):
        (a.sw if rv32 else a.sd)(r, 0, PTR); a.addi(PTR, PTR, 8)
    if rv32:
        _exit_tohost32(a, S["tohost"])
    else:
        _exit_tohost(a, S["tohost"])
    assert a.here() < data_base
    return _finish(a, f"zb{xlen}-{seed}", isa, d, mem_size=1 << 15, max_inst=40000, meta=dict(seed=seed))
