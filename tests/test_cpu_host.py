"""CPU-only tests (no GPU): host logic, the C-ABI surface, the device semantics compiled for the host
(tests/host_shim) against the oracle, the device decoder/softfloat against the reference's own Decoder
and SoftFloat (oracle/_ref), and the N>1 sharding path on gloo."""
import ctypes as C
import os
import random
import re
import subprocess
import sys

import pytest

import simutil as S
from simutil import rvoracle
from whisper_amd import elfio, workloads as WL
from whisper_amd.rvasm import Asm

ROOT = S.ROOT
HAVE_REF = rvoracle.ref() is not None
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built")

# Mnemonics (reference InstEntry names) of the modelled subset: every one is retired without a trap by the
# generated workloads (checked in test_supported_names_are_exercised) or is a privileged/system form.
SUPPORTED = set("""add addi addiw addw amoadd.d amoadd.w amoand.d amoand.w amomax.d amomax.w amomaxu.d amomaxu.w amomin.d amomin.w
amominu.d amominu.w amoor.d amoor.w amoswap.d amoswap.w amoxor.d amoxor.w and andi auipc beq bge bgeu blt bltu bne c.add c.addi
c.addi4spn c.addiw c.addw c.and c.andi c.beqz c.bnez c.fld c.fsd c.j c.jalr c.jr c.ld c.ldsp c.li c.lui c.lw c.lwsp c.mv c.or c.sd
c.sdsp c.slli c.srai c.srli c.sub c.subw c.sw c.swsp c.xor csrrc csrrci csrrs csrrsi csrrw csrrwi div divu divuw divw fadd.d fadd.s
fclass.d fclass.s fcvt.d.l fcvt.d.lu fcvt.d.s fcvt.d.w fcvt.d.wu fcvt.l.d fcvt.l.s fcvt.lu.d fcvt.lu.s fcvt.s.d fcvt.s.lu fcvt.s.w
fcvt.s.wu fcvt.w.d fcvt.w.s fcvt.wu.d fcvt.wu.s fdiv.d fdiv.s fence fence.i feq.d feq.s fld fle.d fle.s flt.d flt.s flw fmadd.d
fmadd.s fmax.d fmax.s fmin.d fmin.s fmsub.d fmsub.s fmul.d fmul.s fmv.d.x fmv.w.x fmv.x.d fmv.x.w fnmadd.d fnmadd.s fnmsub.d
fnmsub.s fsd fsgnj.d fsgnj.s fsgnjn.d fsgnjn.s fsgnjx.d fsgnjx.s fsqrt.d fsqrt.s fsub.d fsub.s fsw jal jalr lb lbu ld lh lhu lr.d
lr.w lui lw lwu mret mul mulh mulhsu mulhu mulw or ori rem remu remuw remw sb sc.d sc.w sd sh sll slli slliw sllw slt slti sltiu
sltu sra srai sraiw sraw srl srli srliw srlw sub subw sw wfi xor xori
fcvt.s.l c.nop c.ebreak c.fldsp c.fsdsp c.addi16sp ecall ebreak sret sfence.vma fence.tso""".split())


class Dec(C.Structure):
    _fields_ = [("w0", C.c_uint32), ("imm", C.c_int32), ("pad", C.c_uint32), ("raw", C.c_uint32)]


def _host_decode(isa, words):
    L = S.shim()
    L.hs_decode.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(Dec)]
    n = len(words)
    arr = (C.c_uint32 * n)(*words)
    out = (Dec * n)()
    L.hs_decode(rvoracle.isa_flags(isa), arr, n, out)
    return out


# ------------------------------------------------------------------------------------------ decoder
@needs_ref
def test_decoder_legality_matches_reference_decoder():
    """Device decoder (host build) vs the reference's Decoder.cpp on 125k words: whatever the reference calls illegal is
    illegal here; whatever decodes here is a SUPPORTED mnemonic there; a SUPPORTED mnemonic rejected here must be one the
    reference rejects at execute time (reserved rounding mode, Hart-level checks) - listed explicitly."""
    rng = random.Random(5)
    words = [rng.getrandbits(32) | 3 for _ in range(60000)] + list(range(0, 65536))
    words = [w for w in words if w < 65536 and (w & 3) != 3 or (w & 3) == 3 and (w & 0x1c) != 0x1c]
    out = _host_decode("rv64imafdc_zicsr", words)
    exec_time_rejects = 0
    for i, w in enumerate(words):
        _, name, _ = rvoracle.ref_decode(w)
        mine_ill = (out[i].w0 & 0xff) == 0           # OP_ILLEGAL == 0
        if name == "illegal":
            assert mine_ill, f"{w:#x}: reference illegal, decoded here"
        elif not mine_ill:
            assert name in SUPPORTED, f"{w:#x}: decoded here, reference says {name}"
        elif name in SUPPORTED:
            exec_time_rejects += 1
            fp_rm = name.startswith("f") and ((w >> 12) & 7) in (5, 6)                        # float.cpp:203-210
            c_hint_or_reserved = name.startswith("c.")                                        # Decoder.cpp:1455 reserved encodings -> execute-time illegal
            shamt = name in ("slli", "srli", "srai", "slliw", "srliw", "sraiw")
            csr_or_sys = name in ("csrrw", "csrrs", "csrrc", "csrrwi", "csrrsi", "csrrci", "sret", "sfence.vma", "wfi", "mret", "ecall", "ebreak", "fence", "fence.i")
            amo = name.startswith(("amo", "lr.", "sc."))
            assert fp_rm or c_hint_or_reserved or shamt or csr_or_sys or amo, f"{w:#x}: {name} rejected at decode"
    assert exec_time_rejects < len(words) // 10


def test_supported_names_are_exercised():
    """Every non-system mnemonic in SUPPORTED is retired at least once by the generated workloads (so execution parity
    tests cover it)."""
    if not HAVE_REF:
        pytest.skip("oracle/_ref not built")
    names = set()
    progs = [WL.compute_kernel(iters=1)] + WL.archtest_suite() + [WL.fuzz_program(s, 300) for s in range(20)]
    for p in progs:
        _, _, log = S.run_oracle(p, log_cap=200000)
        for c in log:
            if c.trap == 0:
                names.add(rvoracle.ref_decode(c.inst if c.size == 4 else c.inst & 0xffff)[1])
    missing = SUPPORTED - names - set("fence.tso fcvt.s.l c.nop c.ebreak c.fldsp c.fsdsp c.addi16sp ecall ebreak sret sfence.vma".split())
    assert not missing, missing


@needs_ref
def test_assembler_matches_reference_disassembler():
    a = Asm(0x80000000)
    a.addi(1, 2, -3); a.sd(5, 16, 6); a.lw(7, -4, 8); a.beq(1, 2, a.pc + 16); a.jal(1, a.pc + 8); a.lui(3, 0x12345)
    a.mulhsu(4, 5, 6); a.amoadd_w(7, 8, 9); a.lr_d(1, 2); a.fmadd_d(1, 2, 3, 4, 7); a.fcvt_w_s(5, 6, 1); a.csrrw(1, 0x300, 2)
    a.c_addi(9, -5); a.c_li(10, 1); a.c_mv(3, 4); a.c_add(10, 11); a.mret(); a.ecall()
    want = ["addi     x1, x2, -3", "sd       x5, 0x10(x6)", "lw       x7, -0x4(x8)", "beq      x1, x2, . + 0x10", "jal      x1, . + 0x8",
            "lui      x3, 0x12345", "mulhsu   x4, x5, x6", "amoadd.w x7, x8, (x9)", "lr.d     x1, (x2)", "fmadd.d  f1, f2, f3, f4, dyn",
            "fcvt.w.s x5, f6, rtz", "csrrw    x1, mstatus, x2", "c.addi   x9, -0x5", "c.li     x10, 0x1", "c.mv     x3, x4", "c.add    x10, x11",
            "mret", "ecall"]
    text = a.assemble()
    got, off = [], 0
    while off < len(text):
        lo = int.from_bytes(text[off:off + 2], "little")
        if (lo & 3) == 3:
            w = int.from_bytes(text[off:off + 4], "little"); off += 4
        else:
            w = lo; off += 2
        got.append(rvoracle.ref_disassemble(w).strip())
    norm = lambda s: re.sub(r"\s+", " ", s)
    assert [norm(g) for g in got] == [norm(w) for w in want]


# ------------------------------------------------------------------------------------------ softfloat
@needs_ref
def test_device_softfloat_bit_exact_vs_reference_library():
    """whisper_amd/csrc/wsp_softfloat.h (host build) against the reference's vendored SoftFloat 3e: every op x 5 rounding
    modes, directed corners + random operands; results and exception flags."""
    exe = os.path.join(ROOT, "oracle", "_ref", "sf_check")
    if not os.path.exists(exe):
        pytest.skip("sf_check not built")
    out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    m = re.search(r"checked \d+ cases \((\d+) on the FP-unit paths\), 0 mismatches", out.stdout)
    assert m and int(m.group(1)) > 100000, out.stdout[-500:]     # the hw_* paths (host FP unit here) took their share


# ------------------------------------------------------------------------------------------ device semantics on the host vs oracle
def _same(p, **kw):
    ro, h, _ = S.run_oracle(p, **kw)
    rh, xs, fs = S.run_hostsim(p, **kw)
    assert ro == rh, f"{p.name}: {ro} != {rh}: {S.explain_diff(p, ro, h, xs, fs)}"
    return ro


def test_hostsim_readme_program():
    r = _same(WL.readme_program())
    assert r["status"] == rvoracle.STOP_TOHOST and r["value"] == 1 and r["exec"] == r["retired"] <= 9


def test_hostsim_archtest_suite_signatures():
    for p in WL.archtest_suite(4):
        r = _same(p)
        assert r["status"] == rvoracle.STOP_TOHOST, p.name


def test_hostsim_compute_kernel_seeds():
    p = WL.compute_kernel(iters=1)
    for i in (0, 1, 77, 4095):
        r = _same(p, seed_a0=WL.compute_seed(i), seed_mem=i, hart_id=i)
        assert r["status"] == rvoracle.STOP_TOHOST and r["value"] == 1


def test_hostsim_fuzz():
    for s in range(150):
        _same(WL.fuzz_program(s, 300))
    for s in range(1000, 1040):
        _same(WL.fuzz_program(s, 300, fp=False, compressed=False))


def test_hostsim_fuzz_interrupts():
    taken = 0
    for s in range(120):
        p = WL.fuzz_program(s, 300, interrupts=True)
        _same(p)
        _, _, log = S.run_oracle(p, log_cap=30000)
        taken += sum(1 for c in log if c.trap == 2)
    assert taken > 20


def test_hostsim_pmp_programs():
    """Row a7p: TOR/NA4/NAPOT entries, locks, MPRV, fetch / load / store / AMO access faults from U, S and M, the
    4-byte fetch that straddles a region end, pmpcfg / pmpaddr rewrites -- device core (host build) == oracle, with
    pre-decoded fetch and with decode from RAM."""
    faults = stale = 0
    for s in range(120):
        p = WL.pmp_program(s)
        _same(p)
        if s < 30:
            a, _, _ = S.run_hostsim(p, predecode=False)
            b, _, log = S.run_oracle(p, log_cap=40000)
            assert a == b
            faults += sum(1 for c in log if c.trap == 1)
            stale += b["status"] == 4 and b["value"] == 0x181
    assert faults > 1000 and stale >= 1


def test_hostsim_sv39_programs():
    """Row a11: Sv39 walks (4K / 2M / 1G leaves, malformed PTEs), the direct-mapped TLB with its replacement counter and
    stale entries after page-table edits, sfence.vma forms, SUM / MXR, ASID switches, MPRV accesses, page-crossing
    misaligned data and fetches -- device core (host build) == oracle, records and TLB contents included."""
    pf = ok_su = 0
    for s in range(96):
        p = WL.sv39_program(s)
        if s < 24:
            b, _, log = S.run_oracle(p, log_cap=60000)
            a, _, _ = S.run_hostsim(p)
            assert a == b, (s, a, b)
            for c in log:
                if c.trap == 1:
                    pf += any(k == 5 and key in (0x342, 0x142) and val in (12, 13, 15) for k, key, val in c.changes())
                elif c.priv != 3:
                    ok_su += 1
        else:
            _same(p)
        if s < 48:
            _same(WL.sv39_program(s, high=True, exec_body=bool(s & 1)))      # the upper half of the address space, non-canonical addresses
    assert pf > 1500 and ok_su > 10000


def test_hostsim_rvv_programs():
    """Rows a14-a16: the random vector programs (configuration, every memory-op form, integer / fixed-point / mask / permutation /
    FP instructions, LMUL mf8..8, ta / ma, illegal configurations, faults inside vector loads) -- device core (host build) ==
    oracle for VLEN 512, 256 and 128."""
    vw = traps = uns = 0
    for s in range(40):
        for vlenb in (64, 32, 16):
            p = WL.rvv_program(s, vlenb=vlenb)
            if s < 12:
                b, _, log = S.run_oracle(p, log_cap=40000)
                a, _, _ = S.run_hostsim(p)
                assert a == b, (s, vlenb, a, b)
                vw += sum(1 for c in log if any(k == rvoracle.CH_V for k, _, _ in c.changes()))
                traps += sum(1 for c in log if c.trap == 1)
                uns += b["status"] == 4
            else:
                _same(p)
    assert vw > 1500 and traps > 500 and uns >= 1


def test_pmp_semantics_directed():
    """PmpManager.hpp:155-177 + Hart.cpp:3079/3158/1988: no entry -> U store faults (cause 7, tval = address); a NAPOT rw-
    entry admits it; a locked r-- entry makes the same store fault in M mode; a misaligned access that runs into a
    denied word reports the address of the second part."""
    base = 0x80000000
    a = Asm(base)
    a.la(5, "h"); a.csrw(0x305, 5)
    a.li(6, 0x80002000); a.li(7, 0x1234)
    a.li(5, ((0x80002000 >> 2) | 7)); a.csrw(0x3B0, 5)          # entry 0: NAPOT 64 bytes at 0x80002000
    a.li(5, 0x003FFFFFFFFFFFFF); a.csrw(0x3B1, 5)               # entry 1: background
    a.li(5, (0x1F << 8) | 0x1B); a.csrw(0x3A0, 5)               # entry 0 NAPOT rw-, entry 1 NAPOT rwx
    a.li(5, 3 << 11); a.csrc(0x300, 5); a.la(5, "u"); a.csrw(0x341, 5); a.mret()
    a.label("u")
    a.sd(7, 0, 6)                                               # allowed (rw-)
    a.ld(8, 0, 6)
    a.li(9, 0x80002000 + 62); a.sw(7, 0, 9)                     # bytes 62..65: second part (0x80002040) is background rwx: allowed
    a.ecall()                                                   # -> M: lock entry 0 as r--
    a.sd(7, 8, 6)                                               # U store now faults
    a.ecall()                                                   # -> M tries the same store: faults too (locked)
    a.li(9, 0x80002000 - 2); a.sw(7, 0, 9)                      # first part allowed (background), second part denied: tval = 0x80002000
    a.li(17, 1); a.ecall()
    a.align(4)
    a.label("h")
    a.csrr(28, 0x342); a.csrr(29, 0x343)
    a.addi(30, 28, -8); a.bnez(30, "skip")
    a.bnez(17, "exit")
    a.addi(20, 20, 1); a.addi(30, 20, -1); a.bnez(30, "second")
    a.li(5, 0x99); a.csrrw(0, 0x3A0, 5)                         # first ecall: entry 0 := L | NAPOT | r--  (entry 1 byte -> Off!)
    a.li(5, (0x1F << 8) | 0x99); a.csrw(0x3A0, 5)               # locked byte stays, entry 1 back on
    a.j("skip")
    a.label("second")
    a.csrr(24, 0x341); a.csrr(25, 0x300)
    a.sd(7, 16, 6); a.csrr(22, 0x342); a.csrr(23, 0x343)                 # M store into the locked r-- region: nested trap
    a.csrw(0x341, 24); a.csrw(0x300, 25)
    a.label("skip")
    a.csrr(30, 0x341); a.addi(30, 30, 4); a.csrw(0x341, 30); a.mret()
    a.label("exit")
    a.li(30, 0x80001000); a.li(31, 1); a.sd(31, 0, 30)
    a.label("hang"); a.j("hang")
    img = a.assemble()
    h = rvoracle.Hart(isa="rv64imacsu_zicsr", mem_base=base, mem_size=1 << 16, tohost=0x80001000, reset_pc=base)
    h.write_mem(base, img)
    st, log = h.run(5000, log_cap=5000)
    assert st == 1
    traps = [(c.pc, [(key, val) for kind, key, val in c.changes() if kind == 5]) for c in log if c.trap == 1]
    causes = [dict(t[1]) for t in traps]
    seq = [(c[0x342], c[0x343]) for c in causes]
    # ecall(8), U store fault, ecall(8), nested M store fault, U misaligned store fault at the second part, final ecall
    assert seq[0][0] == 8
    assert seq[1] == (7, 0x80002008)
    assert seq[2][0] == 8
    assert seq[3] == (7, 0x80002010)
    assert seq[4] == (7, 0x80002000)
    assert seq[5][0] == 8 and len(seq) == 6
    assert (h.x(22), h.x(23)) == (7, 0x80002010)
    assert h.x(8) == 0x1234 and h.read_mem(0x80002000 + 62, 4) == (0x1234).to_bytes(4, "little")
    assert h.csr(0x3A0) == (0x1F << 8) | 0x99


def test_hostsim_decode_from_ram_equals_predecoded():
    p = WL.fuzz_program(3, 300)
    a, _, _ = S.run_hostsim(p, predecode=True)
    b, _, _ = S.run_hostsim(p, predecode=False)
    assert a == b


def test_interrupt_semantics_direct_and_vectored():
    """Hart.cpp:7028-7083 + 3565-3905: SSIP raised and enabled in M mode, not delegated -> M trap, mcause = 1<<63|1,
    mepc = interrupted pc, mtval = 0, MPIE<-MIE, MIE<-0, vectored pc = base + 4*cause; delegated -> not taken in M."""
    for vectored in (0, 1):
        base = 0x80000000
        a = Asm(base)
        a.la(5, "vec"); a.addi(5, 5, vectored); a.csrw(0x305, 5)
        a.li(6, 2); a.csrw(0x304, 6); a.csrw(0x344, 6)          # mie.SSIE, mip.SSIP
        a.csrrsi(0, 0x300, 8)                                    # mstatus.MIE -> interrupt before the next instruction
        a.label("resume"); a.li(10, 0x111); a.jal(0, "done")
        a.align(4)
        a.label("vec"); a.jal(0, "direct")                       # base: exceptions / direct mode
        a.jal(0, "cause1")                                       # base + 4: interrupt cause 1 in vectored mode
        a.label("direct"); a.li(11, 0xE); a.jal(0, "done")
        a.label("cause1"); a.li(11, 0x1); a.jal(0, "done")
        a.label("done"); a.li(5, 1); a.li(6, base + 0x1000); a.sd(5, 0, 6); a.jal(0, a.pc)
        text = a.assemble()
        h = rvoracle.Hart(isa="rv64imacsu_zicsr", mem_size=1 << 14, tohost=base + 0x1000)
        h.write_mem(base, text)
        st, log = h.run(1000, log_cap=64)
        assert st == rvoracle.STOP_TOHOST
        irq = [c for c in log if c.trap == 2]
        assert len(irq) == 1 and irq[0].pc == a.symbols["resume"]
        assert h.csr(0x342) == (1 << 63) | 1 and h.csr(0x341) == a.symbols["resume"] and h.csr(0x343) == 0
        assert h.x(11) == (0x1 if vectored else 0xE) and h.x(10) == 0
        ms = h.csr(0x300)
        assert (ms >> 7) & 1 == 1 and (ms >> 3) & 1 == 0 and (ms >> 11) & 3 == 3
    # delegated to S: never taken while in M mode
    a = Asm(base)
    a.li(6, 2); a.csrw(0x303, 6); a.csrw(0x304, 6); a.csrw(0x344, 6); a.csrrsi(0, 0x300, 8)
    a.li(10, 0x111); a.li(5, 1); a.li(6, base + 0x1000); a.sd(5, 0, 6); a.jal(0, a.pc)
    h = rvoracle.Hart(isa="rv64imacsu_zicsr", mem_size=1 << 14, tohost=base + 0x1000)
    h.write_mem(base, a.assemble())
    st, log = h.run(1000, log_cap=64)
    assert st == rvoracle.STOP_TOHOST and not any(c.trap == 2 for c in log) and h.x(10) == 0x111


# ------------------------------------------------------------------------------------------ host logic
def test_elf_writer_reader_roundtrip():
    segs = [(0x80000000, b"\x13\x00\x00\x00" * 5, elfio.PF_R | elfio.PF_X), (0x80002000, bytes(range(64)), elfio.PF_R | elfio.PF_W)]
    syms = {"_start": 0x80000000, "tohost": 0x80002000, "begin_signature": 0x80002010, "end_signature": 0x80002040}
    img = elfio.write_elf(segs, 0x80000000, syms)
    entry, rsegs, rsyms = elfio.read_elf(img)
    assert entry == 0x80000000
    assert [(s[0], bytes(s[1])) for s in rsegs] == [(p, d) for p, d, _ in segs]
    for k, v in syms.items():
        assert rsyms[k] == v


def test_isa_string_parsing_matches_oracle_flags():
    import whisper_amd as W
    for isa in ("rv64imafdc_zicsr", "rv64imac", "rv64gc", "rv64imafdcsu_zicsr_zicntr", "rv64i"):
        assert W.lib().wsp_isa_flags(isa.encode()) == rvoracle.isa_flags(isa), isa


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly the entry points include/whisper_amd.h declares."""
    import whisper_amd as W
    hdr = open(os.path.join(ROOT, "include", "whisper_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = re.findall(r"\b(wsp_[a-z0-9_]+)\s*\(", hdr)
    assert len(names) >= 22
    L = W.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/whisper_amd.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", W.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (wsp_[a-z0-9_]+)", out))
    assert exported == set(names), exported ^ set(names)


def test_no_cpu_fallback_without_device():
    """HIP only: with no usable device the batch cannot be created (and nothing silently runs on the CPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import whisper_amd as W
    with pytest.raises(W.WspError) as ei:
        W.Batch(4, isa="rv64imac_zicsr", mem_base=0x80000000, mem_size=1 << 14)
    assert ei.value.code in (W.ENODEV, W.EHIP)


def test_host_decoder_in_library_equals_shim_build():
    import whisper_amd as W
    rng = random.Random(11)
    words = [rng.getrandbits(32) for _ in range(5000)] + [rng.getrandbits(16) for _ in range(5000)]
    a = W.decode_words("rv64imafdc_zicsr", words, host=True)
    b = _host_decode("rv64imafdc_zicsr", words)
    for i in range(len(words)):
        assert (a[i].w0, a[i].imm, a[i].raw) == (b[i].w0, b[i].imm, b[i].raw), hex(words[i])


# ------------------------------------------------------------------------------------------ N > 1 (gloo, world size 2)
_WORKER = r"""
import os, sys, ctypes as C
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
import simutil as S
import whisper_amd as W
from whisper_amd import shard, workloads as WL
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
per = 3
lo, hi = shard.weak_range(per, rank)
p = WL.compute_kernel(iters=1)
recs = (W.Result * per)()
insts = 0
for k, i in enumerate(range(lo, hi)):                       # the oracle stands in for the device on this CPU-only box
    r, _, _ = S.run_oracle(p, seed_a0=WL.compute_seed(i), seed_mem=i, hart_id=i, max_inst=3000)
    recs[k].status = r["status"]; recs[k].stop_value = r["value"]; recs[k].pc = r["pc"]; recs[k].exec_count = r["exec"]
    recs[k].retired_count = r["retired"]; recs[k].state_hash = r["state_hash"]; recs[k].mem_hash = r["mem_hash"]
    insts += r["exec"]
parts = shard.gather_results(bytes(recs), dist)
tmax, tot = shard.reduce_totals(1.0 + rank, float(insts), dist)
allrec = (W.Result * (per * world)).from_buffer_copy(b"".join(parts))
ok = len(parts) == world and tmax == float(world) and tot == sum(r.exec_count for r in allrec)
mine = [(r.state_hash, r.exec_count) for r in allrec[lo:hi]] == [(r.state_hash, r.exec_count) for r in recs]
distinct = len(set(r.state_hash for r in allrec)) == per * world
# strong split of ONE list of 7 systems over the ranks (uneven: 3 + 4), gathered, against the whole list run by one rank
total = 7
slo, shi = shard.partition(total, world)[rank]
def one(i):
    r, _, _ = S.run_oracle(p, seed_a0=WL.compute_seed(i), seed_mem=i, hart_id=i, max_inst=2000)
    return (r["status"], r["exec"], r["state_hash"], r["mem_hash"], r["trace_hash"])
srecs = (W.Result * (shi - slo))()
for k, i in enumerate(range(slo, shi)):
    st, ex, sh, mh, th = one(i)
    srecs[k].status = st; srecs[k].exec_count = ex; srecs[k].state_hash = sh; srecs[k].mem_hash = mh; srecs[k].trace_hash = th
sparts = shard.gather_results(bytes(srecs), dist)
sall = (W.Result * total).from_buffer_copy(b"".join(sparts))
strong_ok = [len(x) // C.sizeof(W.Result) for x in sparts] == [hi_ - lo_ for lo_, hi_ in shard.partition(total, world)]
strong_ok = strong_ok and [(r.status, r.exec_count, r.state_hash, r.mem_hash, r.trace_hash) for r in sall] == [one(i) for i in range(total)]
ok = ok and strong_ok
# one write per rank: print() with several arguments writes them piece by piece, and the two ranks share one pipe
# (that interleaving - "RANK RANK1  0OK" - was the intermittent failure of this test)
sys.stdout.write("RANK %d %s\n" % (rank, "OK" if ok and mine and distinct else "BAD")); sys.stdout.flush()
dist.destroy_process_group()
"""


def test_sharding_gather_world_size_2_gloo(tmp_path):
    from whisper_amd import shard
    assert shard.partition(10, 4) == [(0, 2), (2, 5), (5, 7), (7, 10)]
    assert shard.weak_range(4096, 3) == (12288, 16384)
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    # Causes of the failures seen in rounds 1 and 2, all removed: (a) both ranks printed their verdict with a multi-argument print()
    # into one pipe and the pieces interleaved (found from the captured output: "RANK RANK1  0OK"); (b) both ranks could rebuild a
    # stale oracle library at the same time: builds run under a file lock and once, here, before the ranks start; (c) the launcher
    # picks its own rendezvous port (--standalone).
    S.rvoracle.lib()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
                          "--nproc-per-node", "2", str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    assert "RANK 0 OK" in out.stdout and "RANK 1 OK" in out.stdout, out.stdout + out.stderr[-2000:]
