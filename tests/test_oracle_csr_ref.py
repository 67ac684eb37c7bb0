"""CSR restatement (oracle) vs the reference's own CsRegs class compiled into oracle/_ref:
presence, reset values, and write->read-back of csrrw through the architectural masks
(CsRegs.cpp:1252 read, :3908 write).  CPU only."""
import ctypes as C
import random

import pytest

from simutil import rvoracle
from whisper_amd.rvasm import Asm

R = rvoracle.ref()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref not built")

if R is not None:
    R.ref_csr_new.restype = C.c_void_p
    R.ref_csr_new.argtypes = [C.c_int] * 5
    R.ref_csr_read.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    R.ref_csr_write.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint64]
    R.ref_csr_peek.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    R.ref_csr_free.argtypes = [C.c_void_p]
    R.ref_csr_implemented.argtypes = [C.c_void_p, C.c_uint]

# every CSR the oracle/device model but the PMP file (own test below); satp!=bare/vector are "unsupported" stops
MODELLED = [0x300, 0x301, 0x302, 0x303, 0x304, 0x305, 0x306, 0x30A, 0x320, 0x340, 0x341, 0x342, 0x343, 0x344,
            0x100, 0x104, 0x105, 0x106, 0x10A, 0x140, 0x141, 0x142, 0x143, 0x144, 0x180,
            0x001, 0x002, 0x003, 0xB00, 0xB02, 0xF11, 0xF12, 0xF13, 0xF14, 0xC00, 0xC01, 0xC02,
            # hardwired zero in the default configuration: mhpmevent3..31, mhpmcounter3..31, mseccfg; debug-mode only: dcsr, dpc
            0x323, 0x330, 0x33F, 0xB03, 0xB10, 0xB1F, 0x747]
# written through csrrw in the read-back test (counters are compensated at Hart level, Hart.cpp:13251-13360; satp: below)
WRITABLE = [0x300, 0x302, 0x303, 0x304, 0x305, 0x306, 0x30A, 0x320, 0x340, 0x341, 0x342, 0x343, 0x344,
            0x100, 0x104, 0x105, 0x106, 0x10A, 0x140, 0x141, 0x142, 0x143, 0x144, 0x001, 0x002, 0x003,
            0x323, 0x33F, 0xB03, 0xB1F, 0x747]
CONFIGS = {"rv64imafdc_zicsr": (0, 0, 1, 1, 0), "rv64imafdcsu_zicsr": (1, 1, 1, 1, 0), "rv64imacu_zicsr": (0, 1, 0, 1, 0),
           "rv64imafdsu_zicsr_zicntr": (1, 1, 1, 0, 1)}
M64 = (1 << 64) - 1
FP_CSRS = (1, 2, 3)


@pytest.mark.parametrize("isa", list(CONFIGS))
def test_presence_and_reset_values(isa):
    b = R.ref_csr_new(*CONFIGS[isa])
    h = rvoracle.Hart(isa=isa)
    v = C.c_uint64()
    try:
        for n in MODELLED:
            impl = bool(R.ref_csr_implemented(b, n))
            try:
                o = h.csr(n)
            except KeyError:
                o = None
            assert (o is not None) == impl, f"{isa} csr {n:#x}: oracle models={o is not None} reference implemented={impl}"
            if impl:
                assert R.ref_csr_peek(b, n, C.byref(v))
                assert o == v.value, f"{isa} csr {n:#x} reset value"
    finally:
        R.ref_csr_free(b)


def _values(rng):
    vals = [M64, 0, 0x5555555555555555, 0xAAAAAAAAAAAAAAAA]
    vals += [1 << i for i in range(0, 64, 3)]
    vals += [rng.getrandbits(64) for _ in range(12)]
    return vals


@pytest.mark.parametrize("isa", list(CONFIGS))
def test_csrrw_readback_matches_reference_masks(isa):
    """M-mode csrrw of many patterns into every writable modelled CSR, then csrrs read-back of it and of its
    aliases (mstatus/sstatus, mie/sie, mip/sip, fcsr/fflags/frm): identical in the oracle and the reference."""
    b = R.ref_csr_new(*CONFIGS[isa])
    rng = random.Random(0xC5)
    plan = []
    for n in WRITABLE:
        if not R.ref_csr_implemented(b, n):
            continue
        for val in _values(rng):
            plan.append((n, val))
    rng.shuffle(plan)
    # keep interrupts from being taken while mie/mip hold random bits: xstatus.MIE/SIE stay clear during the shuffled
    # part, then mie is zeroed and mstatus/sstatus get the full patterns
    plan = [(n, val & ~0xA if n in (0x300, 0x100) else val) for n, val in plan]
    plan.append((0x304, 0))
    for n in (0x300, 0x100):
        if R.ref_csr_implemented(b, n):
            plan += [(n, val) for val in _values(rng)]
    # MPRV would redirect the program's own stores to U/S privilege (PMP denies them): never set it
    plan = [(n, val & ~(1 << 17) if n == 0x300 else val) for n, val in plan]
    alias = {0x300: [0x100], 0x100: [0x300], 0x304: [0x104], 0x104: [0x304], 0x344: [0x144], 0x144: [0x344],
             0x001: [0x003], 0x002: [0x003], 0x003: [0x001, 0x002], 0x303: [0x104, 0x144], 0x306: [], 0x30A: [0x10A]}
    base, data = 0x80000000, 0x80000000 + (1 << 20)
    a = Asm(base)
    a.li(7, data)
    reads = []
    for n, val in plan:
        if n in FP_CSRS:       # Hart.cpp:12945-12958: FP CSRs are illegal while mstatus.FS is Off; FS=Dirty also makes
            a.li(5, 0x6000)    # the Hart-level markFsDirty after the write a no-op (CsRegs alone does not do it)
            a.csrrs(0, 0x300, 5)
        a.li(5, val)
        a.csrrw(0, n, 5)
        for r in [n] + [x for x in alias.get(n, []) if R.ref_csr_implemented(b, x)]:
            a.csrrs(6, r, 0)
            a.sd(6, 0, 7)
            a.addi(7, 7, 8)
            reads.append((n, val, r))
    a.li(5, 1); a.li(6, data - 16); a.sd(5, 0, 6)
    text = a.assemble()
    h = rvoracle.Hart(isa=isa, mem_base=base, mem_size=1 << 22, tohost=data - 16)
    h.write_mem(base, text)
    st, _ = h.run(1 << 24)
    assert st == rvoracle.STOP_TOHOST and h.retired == h.exec_count, f"no trap expected: status {st} value {h.stop_value:#x} exec {h.exec_count} retired {h.retired} pc {h.pc:#x}"
    got = h.read_mem(data, 8 * len(reads))
    v = C.c_uint64()
    i = 0
    try:
        for n, val in plan:
            if n in FP_CSRS:
                assert R.ref_csr_read(b, 0x300, 3, C.byref(v)) and R.ref_csr_write(b, 0x300, 3, v.value | 0x6000)
            assert R.ref_csr_write(b, n, 3, val)
            for r in [n] + [x for x in alias.get(n, []) if R.ref_csr_implemented(b, x)]:
                assert R.ref_csr_read(b, r, 3, C.byref(v))
                o = int.from_bytes(got[8 * i:8 * i + 8], "little")
                assert o == v.value, f"{isa}: after csrrw {n:#x} <- {val:#x}, csr {r:#x}: oracle {o:#x} reference {v.value:#x}"
                i += 1
    finally:
        R.ref_csr_free(b)


def test_pmp_csrs_presence_and_write_sequences_match_reference():
    """pmpcfg0,2,..14 / pmpaddr0..63 (CsRegs.cpp:4777-4800): presence (odd pmpcfg absent on RV64), write masks, per-byte
    legalisation (PmpManager.hpp:480-551) and pmpaddr locks (CsRegs.cpp:6773-6793) under random interleaved writes."""
    isa = "rv64imafdcsu_zicsr"
    b = R.ref_csr_new(*CONFIGS[isa])
    h = rvoracle.Hart(isa=isa)
    v = C.c_uint64()
    rng = random.Random(0x9A9)
    try:
        for n in range(0x3A0, 0x3F0):
            impl = bool(R.ref_csr_implemented(b, n))
            try:
                o = h.csr(n)
            except KeyError:
                o = None
            assert (o is not None) == impl, f"csr {n:#x}: oracle models={o is not None} reference implemented={impl}"
        for step in range(3000):
            if rng.random() < 0.6:
                n = 0x3B0 + rng.choice([0, 1, 2, 3, 7, 8, 9, 15, 16, 62, 63, rng.randrange(64)])
                val = rng.choice([rng.getrandbits(64), rng.getrandbits(54), rng.getrandbits(20), 0, M64])
            else:
                n = 0x3A0 + 2 * rng.choice([0, 0, 1, 2, 7, rng.randrange(8)])
                val = rng.getrandbits(64)
                if rng.random() < 0.8:           # mostly unlocked so that the file stays writable for a while
                    val &= 0x7F7F7F7F7F7F7F7F | (0x80 << (8 * rng.randrange(8)) if rng.random() < 0.15 else 0)
            assert R.ref_csr_write(b, n, 3, val)
            h.set_csr(n, val)
            if step % 16 == 15 or step > 2900:
                for r in list(range(0x3A0, 0x3B0, 2)) + list(range(0x3B0, 0x3F0)):
                    assert R.ref_csr_read(b, r, 3, C.byref(v))
                    assert h.csr(r) == v.value, f"step {step} after {n:#x} <- {val:#x}: csr {r:#x} oracle {h.csr(r):#x} reference {v.value:#x}"
    finally:
        R.ref_csr_free(b)
