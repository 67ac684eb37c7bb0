"""Tiny RISC-V assembler used to build the synthetic workloads and parity-test programs.

There is no RISC-V toolchain in the build image and the reference ships no binaries
(SURVEY.md section 8c), so every program the tests and bench.py run is assembled here.
Encodings follow the RISC-V unprivileged/privileged specs; tests/test_rvasm.py checks every
mnemonic against the reference's own Decoder/Disassembler (oracle/_ref) when it is present.

Usage:
    a = Asm(base=0x80000000)
    a.li(5, 0x1234); a.label("loop"); a.addi(5, 5, -1); a.bne(5, 0, "loop")
    image = a.assemble()      # bytes
    a.symbols["loop"]         # absolute address
"""
from __future__ import annotations

import struct

MASK64 = (1 << 64) - 1


def _sx(v: int, bits: int) -> int:
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def _chk_simm(v: int, bits: int, what: str) -> int:
    if not -(1 << (bits - 1)) <= v < (1 << (bits - 1)):
        raise ValueError(f"{what}: immediate {v} does not fit {bits} signed bits")
    return v & ((1 << bits) - 1)


class Asm:
    def __init__(self, base: int = 0x80000000, xlen: int = 64):
        self.base = base
        self.xlen = xlen        # 32: li builds 32-bit constants with lui / addi (no addiw, no 64-bit shifts)
        self.items: list = []   # (kind, size, payload)
        self.pc = base
        self.symbols: dict[str, int] = {}

    # ------------------------------------------------------------------ plumbing
    def here(self) -> int:
        return self.pc

    def label(self, name: str) -> int:
        if name in self.symbols:
            raise ValueError(f"duplicate label {name}")
        self.symbols[name] = self.pc
        return self.pc

    def _emit32(self, w: int):
        self.items.append(("w", 4, w & 0xFFFFFFFF))
        self.pc += 4

    def _emit16(self, h: int):
        self.items.append(("h", 2, h & 0xFFFF))
        self.pc += 2

    def _emit_fix(self, size: int, fn):
        """Deferred encoding: fn(symbols, pc) -> int once all labels are known."""
        self.items.append(("f", size, (fn, self.pc)))
        self.pc += size

    def word(self, w: int):
        self._emit32(w)

    def half(self, h: int):
        self._emit16(h)

    def dword(self, d: int):
        self._emit32(d & 0xFFFFFFFF)
        self._emit32((d >> 32) & 0xFFFFFFFF)

    def data(self, b: bytes):
        self.items.append(("b", len(b), bytes(b)))
        self.pc += len(b)

    def zeros(self, n: int):
        self.data(bytes(n))

    def align(self, n: int):
        while self.pc % n:
            if self.pc % 2:
                self.data(b"\0")
            elif (self.pc % 4) and n >= 4:
                self._emit16(0x0001)  # c.nop
            else:
                self._emit32(0x00000013)  # nop

    def align_data(self, n: int):
        pad = (-self.pc) % n
        if pad:
            self.zeros(pad)

    def org(self, addr: int):
        if addr < self.pc:
            raise ValueError("org goes backwards")
        self.zeros(addr - self.pc)

    def _target(self, t) -> int:
        return self.symbols[t] if isinstance(t, str) else int(t)

    def assemble(self) -> bytes:
        out = bytearray()
        for kind, size, payload in self.items:
            if kind == "w":
                out += struct.pack("<I", payload)
            elif kind == "h":
                out += struct.pack("<H", payload)
            elif kind == "b":
                out += payload
            else:
                fn, pc = payload
                v = fn(pc)
                out += struct.pack("<I" if size == 4 else "<H", v & (0xFFFFFFFF if size == 4 else 0xFFFF))
        return bytes(out)

    # ------------------------------------------------------------------ raw formats
    @staticmethod
    def enc_r(op, f3, f7, rd, rs1, rs2):
        return (f7 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op

    @staticmethod
    def enc_i(op, f3, rd, rs1, imm):
        return (_chk_simm(imm, 12, "I") << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op

    @staticmethod
    def enc_s(op, f3, rs1, rs2, imm):
        i = _chk_simm(imm, 12, "S")
        return ((i >> 5) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | ((i & 31) << 7) | op

    @staticmethod
    def enc_b(f3, rs1, rs2, off):
        if off & 1:
            raise ValueError("branch offset odd")
        i = _chk_simm(off, 13, "B")
        return (((i >> 12) & 1) << 31) | (((i >> 5) & 0x3F) << 25) | (rs2 << 20) | (rs1 << 15) | \
               (f3 << 12) | (((i >> 1) & 0xF) << 8) | (((i >> 11) & 1) << 7) | 0x63

    @staticmethod
    def enc_u(op, rd, imm20):
        return ((imm20 & 0xFFFFF) << 12) | (rd << 7) | op

    @staticmethod
    def enc_j(rd, off):
        if off & 1:
            raise ValueError("jal offset odd")
        i = _chk_simm(off, 21, "J")
        return (((i >> 20) & 1) << 31) | (((i >> 1) & 0x3FF) << 21) | (((i >> 11) & 1) << 20) | \
               (((i >> 12) & 0xFF) << 12) | (rd << 7) | 0x6F

    # ------------------------------------------------------------------ pseudo
    def nop(self):
        self.addi(0, 0, 0)

    def mv(self, rd, rs):
        self.addi(rd, rs, 0)

    def li(self, rd, value: int):
        """Load a 64-bit constant (sign-agnostic) with lui/addi(w)/slli sequences."""
        value &= MASK64
        sv = _sx(value, 64)
        if self.xlen == 32:
            sv = _sx(value & 0xFFFFFFFF, 32)
        if -2048 <= sv < 2048:
            self.addi(rd, 0, sv)
            return
        if -(1 << 31) <= sv < (1 << 31):
            lo = _sx(sv & 0xFFF, 12)
            hi = ((sv - lo) >> 12) & 0xFFFFF
            self.lui(rd, hi)
            if lo:
                (self.addi if self.xlen == 32 else self.addiw)(rd, rd, lo)
            return
        # general: build upper part recursively then shift in 12-bit chunks
        lo = _sx(sv & 0xFFF, 12)
        rest = (sv - lo) >> 12
        shift = 12
        while rest & 1 == 0:
            rest >>= 1
            shift += 1
        self.li(rd, rest)
        self.slli(rd, rd, shift)
        if lo:
            self.addi(rd, rd, lo)

    def la(self, rd, target):
        """auipc+addi pc-relative address of a label (resolved at assemble time)."""
        pc0 = self.pc

        def hi(pc, t=target):
            off = self._target(t) - pc0
            lo = _sx(off & 0xFFF, 12)
            return self.enc_u(0x17, rd, ((off - lo) >> 12) & 0xFFFFF)

        def lo_(pc, t=target):
            off = self._target(t) - pc0
            return self.enc_i(0x13, 0, rd, rd, _sx(off & 0xFFF, 12))

        self._emit_fix(4, hi)
        self._emit_fix(4, lo_)

    def j(self, target):
        self.jal(0, target)

    def ret(self):
        self.jalr(0, 1, 0)

    def call(self, target):
        self.jal(1, target)

    def beqz(self, rs, t):
        self.beq(rs, 0, t)

    def bnez(self, rs, t):
        self.bne(rs, 0, t)

    def bltz(self, rs, t):
        self.blt(rs, 0, t)

    def bgez(self, rs, t):
        self.bge(rs, 0, t)

    def csrr(self, rd, csr):
        self.csrrs(rd, csr, 0)

    def csrw(self, csr, rs):
        self.csrrw(0, csr, rs)

    def csrs(self, csr, rs):
        self.csrrs(0, csr, rs)

    def csrc(self, csr, rs):
        self.csrrc(0, csr, rs)

    # ------------------------------------------------------------------ control flow
    def jal(self, rd, target):
        self._emit_fix(4, lambda pc, t=target: self.enc_j(rd, self._target(t) - pc))

    def jalr(self, rd, rs1, imm=0):
        self._emit32(self.enc_i(0x67, 0, rd, rs1, imm))

    def _branch(self, f3, rs1, rs2, target):
        self._emit_fix(4, lambda pc, t=target: self.enc_b(f3, rs1, rs2, self._target(t) - pc))

    def lui(self, rd, imm20):
        self._emit32(self.enc_u(0x37, rd, imm20))

    def auipc(self, rd, imm20):
        self._emit32(self.enc_u(0x17, rd, imm20))

    # ------------------------------------------------------------------ system
    def ecall(self):
        self._emit32(0x00000073)

    def ebreak(self):
        self._emit32(0x00100073)

    def mret(self):
        self._emit32(0x30200073)

    def sret(self):
        self._emit32(0x10200073)

    def wfi(self):
        self._emit32(0x10500073)

    def fence(self, pred=0xF, succ=0xF):
        self._emit32((pred << 24) | (succ << 20) | 0x0F)

    def fence_i(self):
        self._emit32(0x0000100F)

    def sfence_vma(self, rs1=0, rs2=0):
        self._emit32(self.enc_r(0x73, 0, 0x09, 0, rs1, rs2))

    # ------------------------------------------------------------------ RVV (the slice the simulator models)
    @staticmethod
    def vtype(sew=32, lmul=1, ta=False, ma=False):
        """vtype immediate: sew in bits, lmul 1/2/4/8 or a fraction given as 'mf2' / 'mf4' / 'mf8'."""
        l = {1: 0, 2: 1, 4: 2, 8: 3, "mf8": 5, "mf4": 6, "mf2": 7}[lmul]
        return l | ({8: 0, 16: 1, 32: 2, 64: 3}[sew] << 3) | (int(ta) << 6) | (int(ma) << 7)

    def vsetvli(self, rd, rs1, vtypei):
        self._emit32(((vtypei & 0x7FF) << 20) | (rs1 << 15) | (7 << 12) | (rd << 7) | 0x57)

    def vsetivli(self, rd, uimm, vtypei):
        self._emit32((3 << 30) | ((vtypei & 0x3FF) << 20) | ((uimm & 31) << 15) | (7 << 12) | (rd << 7) | 0x57)

    def vsetvl(self, rd, rs1, rs2):
        self._emit32((0x40 << 25) | (rs2 << 20) | (rs1 << 15) | (7 << 12) | (rd << 7) | 0x57)

    def vop(self, f6, f3, vd, vs2, s1, vm=1):
        """Generic OP-V: funct6, funct3 (0 OPIVV 1 OPFVV 2 OPMVV 3 OPIVI 4 OPIVX 5 OPFVF 6 OPMVX), vd, vs2, vs1 / rs1 / imm5, vm (0 = masked by v0)."""
        self._emit32((f6 << 26) | ((vm & 1) << 25) | ((vs2 & 31) << 20) | ((s1 & 31) << 15) | (f3 << 12) | ((vd & 31) << 7) | 0x57)

    def vlse(self, eew, vd, rs1, rs2, vm=1):
        self._emit32((2 << 26) | ((vm & 1) << 25) | (rs2 << 20) | (rs1 << 15) | ({8: 0, 16: 5, 32: 6, 64: 7}[eew] << 12) | (vd << 7) | 0x07)

    def vsse(self, eew, vs3, rs1, rs2, vm=1):
        self._emit32((2 << 26) | ((vm & 1) << 25) | (rs2 << 20) | (rs1 << 15) | ({8: 0, 16: 5, 32: 6, 64: 7}[eew] << 12) | (vs3 << 7) | 0x27)

    def vle(self, eew, vd, rs1, vm=1):
        self._emit32(((vm & 1) << 25) | (rs1 << 15) | ({8: 0, 16: 5, 32: 6, 64: 7}[eew] << 12) | (vd << 7) | 0x07)

    def vse(self, eew, vs3, rs1, vm=1):
        self._emit32(((vm & 1) << 25) | (rs1 << 15) | ({8: 0, 16: 5, 32: 6, 64: 7}[eew] << 12) | (vs3 << 7) | 0x27)

    def vls(self, store, eew, v, rs1, lumop=0, vm=1, nf=0, mew=0, mop=0):
        """Generic vector load / store word: lumop 8 = whole registers (nf = count - 1), 0xb = mask (vlm.v / vsm.v)."""
        self._emit32((nf << 29) | (mew << 28) | (mop << 26) | ((vm & 1) << 25) | ((lumop & 31) << 20) | (rs1 << 15) |
                     ({8: 0, 16: 5, 32: 6, 64: 7}[eew] << 12) | ((v & 31) << 7) | (0x27 if store else 0x07))

    # ------------------------------------------------------------------ compressed (explicit)
    def c_nop(self):
        self._emit16(0x0001)

    def c_addi(self, rd, imm):
        i = _chk_simm(imm, 6, "c.addi")
        self._emit16((0 << 13) | ((i >> 5) << 12) | (rd << 7) | ((i & 31) << 2) | 1)

    def c_addiw(self, rd, imm):
        i = _chk_simm(imm, 6, "c.addiw")
        self._emit16((1 << 13) | ((i >> 5) << 12) | (rd << 7) | ((i & 31) << 2) | 1)

    def c_li(self, rd, imm):
        i = _chk_simm(imm, 6, "c.li")
        self._emit16((2 << 13) | ((i >> 5) << 12) | (rd << 7) | ((i & 31) << 2) | 1)

    def c_lui(self, rd, imm6):
        i = _chk_simm(imm6, 6, "c.lui")
        self._emit16((3 << 13) | ((i >> 5) << 12) | (rd << 7) | ((i & 31) << 2) | 1)

    def c_addi16sp(self, imm):
        i = _chk_simm(imm, 10, "c.addi16sp")
        self._emit16((3 << 13) | (((i >> 9) & 1) << 12) | (2 << 7) | (((i >> 4) & 1) << 6) |
                     (((i >> 6) & 1) << 5) | (((i >> 7) & 3) << 3) | (((i >> 5) & 1) << 2) | 1)

    def c_addi4spn(self, rdp, uimm):
        u = uimm
        self._emit16((0 << 13) | (((u >> 4) & 3) << 11) | (((u >> 6) & 0xF) << 7) | (((u >> 2) & 1) << 6) |
                     (((u >> 3) & 1) << 5) | ((rdp - 8) << 2) | 0)

    def c_slli(self, rd, sh):
        self._emit16((0 << 13) | ((sh >> 5) << 12) | (rd << 7) | ((sh & 31) << 2) | 2)

    def c_srli(self, rdp, sh):
        self._emit16((4 << 13) | ((sh >> 5) << 12) | (0 << 10) | ((rdp - 8) << 7) | ((sh & 31) << 2) | 1)

    def c_srai(self, rdp, sh):
        self._emit16((4 << 13) | ((sh >> 5) << 12) | (1 << 10) | ((rdp - 8) << 7) | ((sh & 31) << 2) | 1)

    def c_andi(self, rdp, imm):
        i = _chk_simm(imm, 6, "c.andi")
        self._emit16((4 << 13) | ((i >> 5) << 12) | (2 << 10) | ((rdp - 8) << 7) | ((i & 31) << 2) | 1)

    def _c_arith(self, f6bit, f2, rdp, rs2p):
        self._emit16((4 << 13) | (f6bit << 12) | (3 << 10) | ((rdp - 8) << 7) | (f2 << 5) | ((rs2p - 8) << 2) | 1)

    def c_sub(self, rdp, rs2p):
        self._c_arith(0, 0, rdp, rs2p)

    def c_xor(self, rdp, rs2p):
        self._c_arith(0, 1, rdp, rs2p)

    def c_or(self, rdp, rs2p):
        self._c_arith(0, 2, rdp, rs2p)

    def c_and(self, rdp, rs2p):
        self._c_arith(0, 3, rdp, rs2p)

    def c_subw(self, rdp, rs2p):
        self._c_arith(1, 0, rdp, rs2p)

    def c_addw(self, rdp, rs2p):
        self._c_arith(1, 1, rdp, rs2p)

    def c_mv(self, rd, rs2):
        self._emit16((4 << 13) | (0 << 12) | (rd << 7) | (rs2 << 2) | 2)

    def c_add(self, rd, rs2):
        self._emit16((4 << 13) | (1 << 12) | (rd << 7) | (rs2 << 2) | 2)

    def c_jr(self, rs1):
        self._emit16((4 << 13) | (0 << 12) | (rs1 << 7) | 2)

    def c_jalr(self, rs1):
        self._emit16((4 << 13) | (1 << 12) | (rs1 << 7) | 2)

    def c_ebreak(self):
        self._emit16(0x9002)

    @staticmethod
    def _cj_off(off):
        i = _chk_simm(off, 12, "c.j")
        return (((i >> 11) & 1) << 12) | (((i >> 4) & 1) << 11) | (((i >> 8) & 3) << 9) | (((i >> 10) & 1) << 8) | \
               (((i >> 6) & 1) << 7) | (((i >> 7) & 1) << 6) | (((i >> 1) & 7) << 3) | (((i >> 5) & 1) << 2)

    def c_j(self, target):
        self._emit_fix(2, lambda pc, t=target: (5 << 13) | self._cj_off(self._target(t) - pc) | 1)

    @staticmethod
    def _cb_off(off):
        i = _chk_simm(off, 9, "c.b")
        return (((i >> 8) & 1) << 12) | (((i >> 3) & 3) << 10) | (((i >> 6) & 3) << 5) | (((i >> 1) & 3) << 3) | \
               (((i >> 5) & 1) << 2)

    def c_beqz(self, rs1p, target):
        self._emit_fix(2, lambda pc, t=target: (6 << 13) | self._cb_off(self._target(t) - pc) | ((rs1p - 8) << 7) | 1)

    def c_bnez(self, rs1p, target):
        self._emit_fix(2, lambda pc, t=target: (7 << 13) | self._cb_off(self._target(t) - pc) | ((rs1p - 8) << 7) | 1)

    def c_lw(self, rdp, rs1p, uimm):
        u = uimm
        self._emit16((2 << 13) | (((u >> 3) & 7) << 10) | ((rs1p - 8) << 7) | (((u >> 2) & 1) << 6) |
                     (((u >> 6) & 1) << 5) | ((rdp - 8) << 2) | 0)

    def c_sw(self, rs2p, rs1p, uimm):
        u = uimm
        self._emit16((6 << 13) | (((u >> 3) & 7) << 10) | ((rs1p - 8) << 7) | (((u >> 2) & 1) << 6) |
                     (((u >> 6) & 1) << 5) | ((rs2p - 8) << 2) | 0)

    def c_ld(self, rdp, rs1p, uimm):
        u = uimm
        self._emit16((3 << 13) | (((u >> 3) & 7) << 10) | ((rs1p - 8) << 7) | (((u >> 6) & 3) << 5) | ((rdp - 8) << 2) | 0)

    def c_sd(self, rs2p, rs1p, uimm):
        u = uimm
        self._emit16((7 << 13) | (((u >> 3) & 7) << 10) | ((rs1p - 8) << 7) | (((u >> 6) & 3) << 5) | ((rs2p - 8) << 2) | 0)

    def c_fld(self, rdp, rs1p, uimm):
        u = uimm
        self._emit16((1 << 13) | (((u >> 3) & 7) << 10) | ((rs1p - 8) << 7) | (((u >> 6) & 3) << 5) | ((rdp - 8) << 2) | 0)

    def c_fsd(self, rs2p, rs1p, uimm):
        u = uimm
        self._emit16((5 << 13) | (((u >> 3) & 7) << 10) | ((rs1p - 8) << 7) | (((u >> 6) & 3) << 5) | ((rs2p - 8) << 2) | 0)

    def c_lwsp(self, rd, uimm):
        u = uimm
        self._emit16((2 << 13) | (((u >> 5) & 1) << 12) | (rd << 7) | (((u >> 2) & 7) << 4) | (((u >> 6) & 3) << 2) | 2)

    def c_ldsp(self, rd, uimm):
        u = uimm
        self._emit16((3 << 13) | (((u >> 5) & 1) << 12) | (rd << 7) | (((u >> 3) & 3) << 5) | (((u >> 6) & 7) << 2) | 2)

    def c_fldsp(self, rd, uimm):
        u = uimm
        self._emit16((1 << 13) | (((u >> 5) & 1) << 12) | (rd << 7) | (((u >> 3) & 3) << 5) | (((u >> 6) & 7) << 2) | 2)

    def c_swsp(self, rs2, uimm):
        u = uimm
        self._emit16((6 << 13) | (((u >> 2) & 0xF) << 9) | (((u >> 6) & 3) << 7) | (rs2 << 2) | 2)

    def c_sdsp(self, rs2, uimm):
        u = uimm
        self._emit16((7 << 13) | (((u >> 3) & 7) << 10) | (((u >> 6) & 7) << 7) | (rs2 << 2) | 2)

    def c_fsdsp(self, rs2, uimm):
        u = uimm
        self._emit16((5 << 13) | (((u >> 3) & 7) << 10) | (((u >> 6) & 7) << 7) | (rs2 << 2) | 2)


# ---------------------------------------------------------------------- table-generated mnemonics
def _def_r(name, op, f3, f7):
    def f(self, rd, rs1, rs2):
        self._emit32(self.enc_r(op, f3, f7, rd, rs1, rs2))
    setattr(Asm, name, f)


def _def_i(name, op, f3):
    def f(self, rd, rs1, imm):
        self._emit32(self.enc_i(op, f3, rd, rs1, imm))
    setattr(Asm, name, f)


def _def_load(name, op, f3):
    def f(self, rd, imm, rs1):
        self._emit32(self.enc_i(op, f3, rd, rs1, imm))
    setattr(Asm, name, f)


def _def_store(name, op, f3):
    def f(self, rs2, imm, rs1):
        self._emit32(self.enc_s(op, f3, rs1, rs2, imm))
    setattr(Asm, name, f)


def _def_shift(name, op, f3, f6, bits):
    def f(self, rd, rs1, sh):
        if not 0 <= sh < (1 << bits):
            raise ValueError(f"{name}: shift {sh}")
        self._emit32((f6 << 26) | (sh << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op)
    setattr(Asm, name, f)


def _def_branch(name, f3):
    def f(self, rs1, rs2, target):
        self._branch(f3, rs1, rs2, target)
    setattr(Asm, name, f)


def _def_csr(name, f3, immform):
    def f(self, rd, csr, src):
        self._emit32(((csr & 0xFFF) << 20) | ((src & 31) << 15) | (f3 << 12) | (rd << 7) | 0x73)
    setattr(Asm, name, f)


for _n, _f3, _f7 in [("add", 0, 0), ("sub", 0, 0x20), ("sll", 1, 0), ("slt", 2, 0), ("sltu", 3, 0), ("xor", 4, 0),
                     ("srl", 5, 0), ("sra", 5, 0x20), ("or_", 6, 0), ("and_", 7, 0),
                     ("mul", 0, 1), ("mulh", 1, 1), ("mulhsu", 2, 1), ("mulhu", 3, 1), ("div", 4, 1), ("divu", 5, 1),
                     ("rem", 6, 1), ("remu", 7, 1)]:
    _def_r(_n, 0x33, _f3, _f7)
for _n, _f3, _f7 in [("addw", 0, 0), ("subw", 0, 0x20), ("sllw", 1, 0), ("srlw", 5, 0), ("sraw", 5, 0x20),
                     ("mulw", 0, 1), ("divw", 4, 1), ("divuw", 5, 1), ("remw", 6, 1), ("remuw", 7, 1)]:
    _def_r(_n, 0x3B, _f3, _f7)
for _n, _f3 in [("addi", 0), ("slti", 2), ("sltiu", 3), ("xori", 4), ("ori", 6), ("andi", 7)]:
    _def_i(_n, 0x13, _f3)
_def_i("addiw", 0x1B, 0)
_def_shift("slli", 0x13, 1, 0x00, 6)
_def_shift("srli", 0x13, 5, 0x00, 6)
_def_shift("srai", 0x13, 5, 0x10, 6)
_def_shift("slliw", 0x1B, 1, 0x00, 5)
_def_shift("srliw", 0x1B, 5, 0x00, 5)
_def_shift("sraiw", 0x1B, 5, 0x10, 5)
for _n, _f3 in [("lb", 0), ("lh", 1), ("lw", 2), ("ld", 3), ("lbu", 4), ("lhu", 5), ("lwu", 6)]:
    _def_load(_n, 0x03, _f3)
for _n, _f3 in [("sb", 0), ("sh", 1), ("sw", 2), ("sd", 3)]:
    _def_store(_n, 0x23, _f3)
_def_load("flw", 0x07, 2)
_def_load("fld", 0x07, 3)
_def_store("fsw", 0x27, 2)
_def_store("fsd", 0x27, 3)
for _n, _f3 in [("beq", 0), ("bne", 1), ("blt", 4), ("bge", 5), ("bltu", 6), ("bgeu", 7)]:
    _def_branch(_n, _f3)
for _n, _f3 in [("csrrw", 1), ("csrrs", 2), ("csrrc", 3), ("csrrwi", 5), ("csrrsi", 6), ("csrrci", 7)]:
    _def_csr(_n, _f3, _f3 >= 5)


# A extension: name(rd, rs2, rs1, aq=0, rl=0); lr has no rs2
def _def_amo(name, f5, f3):
    if name.startswith("lr"):
        def f(self, rd, rs1, aq=0, rl=0):
            self._emit32((f5 << 27) | (aq << 26) | (rl << 25) | (rs1 << 15) | (f3 << 12) | (rd << 7) | 0x2F)
    else:
        def f(self, rd, rs2, rs1, aq=0, rl=0):
            self._emit32((f5 << 27) | (aq << 26) | (rl << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | 0x2F)
    setattr(Asm, name, f)


for _sfx, _f3 in [("w", 2), ("d", 3)]:
    for _n, _f5 in [("lr", 2), ("sc", 3), ("amoswap", 1), ("amoadd", 0), ("amoxor", 4), ("amoand", 12), ("amoor", 8),
                    ("amomin", 16), ("amomax", 20), ("amominu", 24), ("amomaxu", 28)]:
        _def_amo(f"{_n}_{_sfx}", _f5, _f3)

RM_DYN = 7


# F/D: arithmetic name(rd, rs1, rs2, rm=7)
def _def_fp_r(name, f7, rm_fixed=None, rs2_fixed=None):
    if rs2_fixed is not None and rm_fixed is None:
        def f(self, rd, rs1, rm=RM_DYN):
            self._emit32(self.enc_r(0x53, rm, f7, rd, rs1, rs2_fixed))
    elif rs2_fixed is not None:
        def f(self, rd, rs1):
            self._emit32(self.enc_r(0x53, rm_fixed, f7, rd, rs1, rs2_fixed))
    elif rm_fixed is None:
        def f(self, rd, rs1, rs2, rm=RM_DYN):
            self._emit32(self.enc_r(0x53, rm, f7, rd, rs1, rs2))
    else:
        def f(self, rd, rs1, rs2):
            self._emit32(self.enc_r(0x53, rm_fixed, f7, rd, rs1, rs2))
    setattr(Asm, name, f)


def _def_fma(name, op, fmt):
    def f(self, rd, rs1, rs2, rs3, rm=RM_DYN):
        self._emit32((rs3 << 27) | (fmt << 25) | (rs2 << 20) | (rs1 << 15) | (rm << 12) | (rd << 7) | op)
    setattr(Asm, name, f)


for _sfx, _fmt in [("s", 0), ("d", 1)]:
    _def_fp_r(f"fadd_{_sfx}", 0x00 | _fmt)
    _def_fp_r(f"fsub_{_sfx}", 0x04 | _fmt)
    _def_fp_r(f"fmul_{_sfx}", 0x08 | _fmt)
    _def_fp_r(f"fdiv_{_sfx}", 0x0C | _fmt)
    _def_fp_r(f"fsqrt_{_sfx}", 0x2C | _fmt, rs2_fixed=0)
    _def_fp_r(f"fsgnj_{_sfx}", 0x10 | _fmt, rm_fixed=0)
    _def_fp_r(f"fsgnjn_{_sfx}", 0x10 | _fmt, rm_fixed=1)
    _def_fp_r(f"fsgnjx_{_sfx}", 0x10 | _fmt, rm_fixed=2)
    _def_fp_r(f"fmin_{_sfx}", 0x14 | _fmt, rm_fixed=0)
    _def_fp_r(f"fmax_{_sfx}", 0x14 | _fmt, rm_fixed=1)
    _def_fp_r(f"fle_{_sfx}", 0x50 | _fmt, rm_fixed=0)
    _def_fp_r(f"flt_{_sfx}", 0x50 | _fmt, rm_fixed=1)
    _def_fp_r(f"feq_{_sfx}", 0x50 | _fmt, rm_fixed=2)
    _def_fp_r(f"fclass_{_sfx}", 0x70 | _fmt, rm_fixed=1, rs2_fixed=0)
    _def_fp_r(f"fcvt_w_{_sfx}", 0x60 | _fmt, rs2_fixed=0)
    _def_fp_r(f"fcvt_wu_{_sfx}", 0x60 | _fmt, rs2_fixed=1)
    _def_fp_r(f"fcvt_l_{_sfx}", 0x60 | _fmt, rs2_fixed=2)
    _def_fp_r(f"fcvt_lu_{_sfx}", 0x60 | _fmt, rs2_fixed=3)
    _def_fp_r(f"fcvt_{_sfx}_w", 0x68 | _fmt, rs2_fixed=0)
    _def_fp_r(f"fcvt_{_sfx}_wu", 0x68 | _fmt, rs2_fixed=1)
    _def_fp_r(f"fcvt_{_sfx}_l", 0x68 | _fmt, rs2_fixed=2)
    _def_fp_r(f"fcvt_{_sfx}_lu", 0x68 | _fmt, rs2_fixed=3)
    _def_fma(f"fmadd_{_sfx}", 0x43, _fmt)
    _def_fma(f"fmsub_{_sfx}", 0x47, _fmt)
    _def_fma(f"fnmsub_{_sfx}", 0x4B, _fmt)
    _def_fma(f"fnmadd_{_sfx}", 0x4F, _fmt)
_def_fp_r("fmv_x_w", 0x70, rm_fixed=0, rs2_fixed=0)
_def_fp_r("fmv_w_x", 0x78, rm_fixed=0, rs2_fixed=0)
_def_fp_r("fmv_x_d", 0x71, rm_fixed=0, rs2_fixed=0)
_def_fp_r("fmv_d_x", 0x79, rm_fixed=0, rs2_fixed=0)
_def_fp_r("fcvt_s_d", 0x20, rs2_fixed=1)
_def_fp_r("fcvt_d_s", 0x21, rs2_fixed=0)

# CSR numbers used by the workloads
CSR = dict(fflags=0x001, frm=0x002, fcsr=0x003, cycle=0xC00, time=0xC01, instret=0xC02,
           sstatus=0x100, sie=0x104, stvec=0x105, scounteren=0x106, sscratch=0x140, sepc=0x141, scause=0x142,
           stval=0x143, sip=0x144, satp=0x180,
           mstatus=0x300, misa=0x301, medeleg=0x302, mideleg=0x303, mie=0x304, mtvec=0x305, mcounteren=0x306,
           mscratch=0x340, mepc=0x341, mcause=0x342, mtval=0x343, mip=0x344, pmpcfg0=0x3A0, pmpaddr0=0x3B0,
           mcycle=0xB00, minstret=0xB02, mvendorid=0xF11, marchid=0xF12, mimpid=0xF13, mhartid=0xF14,
           mcountinhibit=0x320, menvcfg=0x30A, senvcfg=0x10A)
