"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY - see oracle/rv_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "librvoracle.so")
SF_LIB = os.path.join(HERE, "_ref", "libsoftfloat_ref.so")
WH_LIB = os.path.join(HERE, "_ref", "libwhisper_ref.so")

ISA_BITS = dict(m=1, a=2, f=4, d=8, c=16, s=32, u=64, v=128, zicntr=256, zba=1 << 10, zbb=1 << 11, zbs=1 << 12, zicond=1 << 13, zbc=1 << 14)

RUNNING, STOP_TOHOST, STOP_ILLEGAL, STOP_LIMIT, STOP_UNSUPPORTED, STOP_ADDR = range(6)
CH_R, CH_F, CH_V, CH_M, CH_C = 1, 2, 3, 4, 5
MAX_CHANGES = 16


def isa_flags(isa: str) -> int:
    """'rv64imafdc_zicsr_zicntr' -> RVO_ISA_* bit set."""
    s = isa.lower()
    if s.startswith("rv64") or s.startswith("rv32"):
        s = s[4:]
    parts = s.split("_")
    bits = 0
    for ch in parts[0]:
        if ch == "g":
            bits |= 1 | 2 | 4 | 8
        elif ch in ISA_BITS:
            bits |= ISA_BITS[ch]
    for p in parts[1:]:
        if p in ISA_BITS:
            bits |= ISA_BITS[p]
    return bits


class Change(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("pad", C.c_uint32), ("key", C.c_uint64), ("val", C.c_uint64)]


class Commit(C.Structure):
    _fields_ = [("tag", C.c_uint64), ("pc", C.c_uint64), ("inst", C.c_uint32), ("size", C.c_uint8),
                ("priv", C.c_uint8), ("trap", C.c_uint8), ("nchanges", C.c_uint8), ("ch", Change * MAX_CHANGES),
                ("ld_va", C.c_uint64), ("ld_pa", C.c_uint64), ("has_ld", C.c_uint32), ("pad2", C.c_uint32)]

    def changes(self):
        return [(self.ch[i].kind, self.ch[i].key, self.ch[i].val) for i in range(self.nchanges)]


class Config(C.Structure):
    _fields_ = [("xlen", C.c_uint32), ("isa", C.c_uint32), ("mem_base", C.c_uint64), ("mem_size", C.c_uint64),
                ("tohost", C.c_uint64), ("tohost_valid", C.c_uint32), ("hart_id", C.c_uint32), ("reset_pc", C.c_uint64),
                ("tlb_entries", C.c_uint32), ("vlenb", C.c_uint32)]


_lib = None


def build():
    """Build under an exclusive file lock: several processes (ranks of a multi-process test, pytest-xdist workers) may find the
    library stale at the same moment, and a library must never be loaded while another process is still writing it."""
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(HERE, n) for n in ("rv_oracle.c", "rv_oracle.h", "rv_oracle_vec.inc")]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
            build()                                   # a stale oracle is never used silently
        L = C.CDLL(LIB)
        L.rvo_create.restype = C.c_void_p
        L.rvo_create.argtypes = [C.POINTER(Config)]
        L.rvo_destroy.argtypes = [C.c_void_p]
        L.rvo_bind_softfloat.argtypes = [C.c_char_p]
        L.rvo_write_mem.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.rvo_read_mem.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.rvo_set_pc.argtypes = [C.c_void_p, C.c_uint64]
        L.rvo_get_pc.restype = C.c_uint64
        L.rvo_get_pc.argtypes = [C.c_void_p]
        L.rvo_set_x.argtypes = [C.c_void_p, C.c_uint, C.c_uint64]
        L.rvo_get_x.restype = C.c_uint64
        L.rvo_get_x.argtypes = [C.c_void_p, C.c_uint]
        L.rvo_set_f.argtypes = [C.c_void_p, C.c_uint, C.c_uint64]
        L.rvo_get_f.restype = C.c_uint64
        L.rvo_get_f.argtypes = [C.c_void_p, C.c_uint]
        L.rvo_peek_csr.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint64)]
        L.rvo_poke_csr.argtypes = [C.c_void_p, C.c_uint, C.c_uint64]
        L.rvo_priv.argtypes = [C.c_void_p]
        L.rvo_run.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Commit), C.c_size_t, C.POINTER(C.c_size_t)]
        for n in ("rvo_stop_value", "rvo_exec_count", "rvo_retired_count", "rvo_trace_hash", "rvo_state_hash",
                  "rvo_mem_hash"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_void_p]
        L.rvo_status.argtypes = [C.c_void_p]
        L.rvo_set_stop.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64]
        L.rvo_translate.restype = C.c_uint
        L.rvo_translate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_uint, C.POINTER(C.c_uint64)]
        L.rvo_sfence.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_uint64]
        L.rvo_tlb_hash.restype = C.c_uint64
        L.rvo_tlb_hash.argtypes = [C.c_void_p]
        L.rvo_fold.restype = C.c_uint64
        L.rvo_fold.argtypes = [C.c_uint64, C.c_uint64]
        L.rvo_mix.restype = C.c_uint64
        L.rvo_mix.argtypes = [C.c_uint64]
        if os.path.exists(SF_LIB):
            if L.rvo_bind_softfloat(SF_LIB.encode()) != 0:
                raise RuntimeError("cannot bind reference softfloat")
        _lib = L
    return _lib


def has_softfloat() -> bool:
    return os.path.exists(SF_LIB)


class Hart:
    def __init__(self, isa="rv64imafdc_zicsr", mem_base=0x80000000, mem_size=1 << 16, tohost=None,
                 hart_id=0, reset_pc=None, tlb_entries=0, vlenb=0, features=0):
        """features: RVO_CFG_* configuration switches (same bit positions as wsp_config_info.features)."""
        self.L = lib()
        self.xlen = 32 if isa.lower().startswith("rv32") else 64              # Session.cpp:1373-1424: the ISA string decides
        cfg = Config(self.xlen, isa_flags(isa) | features, mem_base, mem_size, tohost or 0, 1 if tohost is not None else 0, hart_id,
                     mem_base if reset_pc is None else reset_pc, tlb_entries, vlenb)
        self.h = self.L.rvo_create(C.byref(cfg))
        if not self.h:
            raise ValueError("rvo_create failed")
        self.cfg = cfg

    def __del__(self):
        if getattr(self, "h", None):
            self.L.rvo_destroy(self.h)
            self.h = None

    def set_stop(self, max_ret=2**64 - 1, stop_pc=None):
        self.L.rvo_set_stop(self.h, max_ret, 0 if stop_pc is None else 1, stop_pc or 0)

    def write_mem(self, addr, data: bytes):
        if self.L.rvo_write_mem(self.h, addr, data, len(data)) != 0:
            raise ValueError(f"write_mem out of range: {addr:#x}+{len(data)}")

    def read_mem(self, addr, n) -> bytes:
        buf = C.create_string_buffer(n)
        if self.L.rvo_read_mem(self.h, addr, buf, n) != 0:
            raise ValueError("read_mem out of range")
        return buf.raw

    @property
    def pc(self):
        return self.L.rvo_get_pc(self.h)

    @pc.setter
    def pc(self, v):
        self.L.rvo_set_pc(self.h, v)

    def x(self, i):
        return self.L.rvo_get_x(self.h, i)

    def set_x(self, i, v):
        self.L.rvo_set_x(self.h, i, v & (2**64 - 1))

    def f(self, i):
        return self.L.rvo_get_f(self.h, i)

    def set_f(self, i, v):
        self.L.rvo_set_f(self.h, i, v)

    def csr(self, num):
        v = C.c_uint64()
        if self.L.rvo_peek_csr(self.h, num, C.byref(v)) != 0:
            raise KeyError(f"csr {num:#x} not modelled")
        return v.value

    def set_csr(self, num, v):
        if self.L.rvo_poke_csr(self.h, num, v) != 0:
            raise KeyError(f"csr {num:#x} not modelled")

    @property
    def priv(self):
        return self.L.rvo_priv(self.h)

    def vreg(self, i) -> bytes:
        buf = C.create_string_buffer(64)
        self.L.rvo_get_vreg.argtypes = [C.c_void_p, C.c_uint, C.c_char_p]
        n = self.L.rvo_get_vreg(self.h, i, buf)
        return buf.raw[:n]

    def set_vreg(self, i, data: bytes):
        self.L.rvo_set_vreg.argtypes = [C.c_void_p, C.c_uint, C.c_char_p]
        self.L.rvo_set_vreg(self.h, i, bytes(data))

    def translate(self, va, pm, acc):
        """(cause, pa) of VirtMem::translate for privilege pm and access acc (1 r, 2 w, 4 x)."""
        pa = C.c_uint64()
        c = self.L.rvo_translate(self.h, va, pm, acc, C.byref(pa))
        return c, pa.value

    def sfence(self, addr=None, asid=None):
        self.L.rvo_sfence(self.h, addr is not None, addr or 0, asid is not None, asid or 0)

    tlb_hash = property(lambda s: s.L.rvo_tlb_hash(s.h))

    def resume(self):
        self.L.rvo_resume.argtypes = [C.c_void_p]
        self.L.rvo_resume.restype = None
        self.L.rvo_resume(self.h)

    def run(self, max_exec=1 << 62, log_cap=0):
        """Returns (status, [Commit...])."""
        if log_cap:
            arr = (Commit * log_cap)()
            n = C.c_size_t(0)
            st = self.L.rvo_run(self.h, max_exec, arr, log_cap, C.byref(n))
            return st, [arr[i] for i in range(n.value)]
        st = self.L.rvo_run(self.h, max_exec, None, 0, None)
        return st, []

    status = property(lambda s: s.L.rvo_status(s.h))
    stop_value = property(lambda s: s.L.rvo_stop_value(s.h))
    exec_count = property(lambda s: s.L.rvo_exec_count(s.h))
    retired = property(lambda s: s.L.rvo_retired_count(s.h))
    trace_hash = property(lambda s: s.L.rvo_trace_hash(s.h))
    state_hash = property(lambda s: s.L.rvo_state_hash(s.h))
    mem_hash = property(lambda s: s.L.rvo_mem_hash(s.h))


# --------------------------------------------------------------------------- reference glue (oracle/_ref)
_ref = None


def ref():
    """The reference's own Decoder/Disassembler + SoftFloat (None when oracle/_ref is absent)."""
    global _ref
    if _ref is None:
        if not (os.path.exists(SF_LIB) and os.path.exists(WH_LIB)):
            return None
        C.CDLL(SF_LIB, mode=C.RTLD_GLOBAL)
        R = C.CDLL(WH_LIB)
        R.ref_decode.restype = C.c_uint
        R.ref_decode.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.c_uint32), C.c_char_p, C.c_int]
        R.ref_disassemble.argtypes = [C.c_int, C.c_uint32, C.c_char_p, C.c_int]
        R.ref_expand16.restype = C.c_uint32
        R.ref_expand16.argtypes = [C.c_int, C.c_uint16]
        _ref = R
    return _ref


def ref_disassemble(inst: int, rv64=True) -> str:
    buf = C.create_string_buffer(256)
    ref().ref_disassemble(1 if rv64 else 0, inst, buf, 256)
    return buf.value.decode()


def ref_decode(inst: int, rv64=True):
    out = (C.c_uint32 * 5)()
    name = C.create_string_buffer(64)
    iid = ref().ref_decode(1 if rv64 else 0, inst, out, name, 64)
    return iid, name.value.decode(), list(out)


# --------------------------------------------------------------------------- commit-log text (RV64)
PRIV_STR = {0: "U", 1: "S", 3: "M"}


def format_commit(c, hart_ix=0, disasm=None, xlen=64, has_d=True, vbytes=None) -> str:
    """One logical commit record in the reference's --log text format (printTrace.cpp:56-209 line layouts for the two widths,
    302-517 record assembly): continuation lines joined by '  +\n' (' +\n' before a vector register line).  disasm: the
    disassembly column; vbytes: {reg: bytes} for the vector register lines (the record itself carries a digest only)."""
    op = f"{c.inst:08x}" if c.size == 4 else f"{c.inst & 0xffff:04x}"
    txt = disasm if disasm is not None else ""
    if c.has_ld:                                             # printTrace.cpp:338-346
        txt += f" [0x{c.ld_va:x}]" if c.ld_va == c.ld_pa else f" [0x{c.ld_va:x}:0x{c.ld_pa:x}]"
    if c.trap == 1:
        txt += " (exception)"
    elif c.trap == 2:
        txt += " (interrupted)"
    out = ""
    pending = False
    for kind, key, val in c.changes():
        if pending:
            out += " +\n" if kind == CH_V else "  +\n"
        head = f"#{c.tag} {hart_ix} {PRIV_STR[c.priv]:>2s} " + (f"{c.pc:08x}" if xlen == 32 else f"{c.pc:016x}") + f" {op:>8s} "
        if kind == CH_V:
            data = vbytes[key] if vbytes else b""
            out += head + f"v {key:02x} " + "".join(f"{x:02x}" for x in reversed(data)) + " " + txt
        elif kind == CH_F:
            if xlen == 32:
                out += head + (f"f {key:02x} {val:016x}  " if has_d else f"f {key:02x}         {val & 0xffffffff:08x}  ") + txt
            else:
                out += head + (f"f {key:016x} {val:016x} " if has_d else f"f {key:016x}         {val & 0xffffffff:08x} ") + txt
        else:
            ch = {CH_R: "r", CH_M: "m", CH_C: "c"}[kind]
            if xlen == 64:
                out += head + f"{ch} {key:016x} {val:016x} " + txt
            elif ch == "r":
                out += head + f"r {key:02x}         {val:08x}  " + txt
            elif ch == "c" and (key >> 16) == 0:
                out += head + f"c {key:04x}       {val:08x}  " + txt
            else:
                out += head + f"{ch} {key:08x}   {val:08x}  " + txt
        pending = True
    return out


def pmp_perm(hart: "Hart", priv: int, addr: int) -> int:
    """PMP permission bits (r=1 w=2 x=4) the oracle grants privilege `priv` at `addr`."""
    L = lib()
    L.rvo_pmp_perm.restype = C.c_uint
    L.rvo_pmp_perm.argtypes = [C.c_void_p, C.c_uint, C.c_uint64]
    return L.rvo_pmp_perm(hart.h, priv, addr)
