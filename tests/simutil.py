"""Shared helpers for the parity tests: run one Program through the CPU oracle, through the
host-compiled build of the device semantics (tests/host_shim), and through the GPU C ABI."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import rvoracle  # noqa: E402

SHIM_SRC = os.path.join(ROOT, "tests", "host_shim", "wsp_hostsim.cpp")
SHIM_LIB = os.path.join(ROOT, "tests", "host_shim", "libwsp_hostsim.so")


def build_shim():
    deps = [SHIM_SRC, os.path.join(ROOT, "whisper_amd", "csrc", "wsp_core.h"), os.path.join(ROOT, "whisper_amd", "csrc", "wsp_softfloat.h")]
    def fresh():
        return os.path.exists(SHIM_LIB) and all(os.path.getmtime(SHIM_LIB) >= os.path.getmtime(d) for d in deps)
    if fresh():
        return
    import fcntl
    with open(SHIM_LIB + ".lock", "w") as lk:          # one builder at a time; the library appears under its name only when complete
        fcntl.flock(lk, fcntl.LOCK_EX)
        if fresh():
            return
        tmp = SHIM_LIB + f".{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", tmp, SHIM_SRC])
        os.replace(tmp, SHIM_LIB)


_shim = None


def shim():
    global _shim
    if _shim is None:
        build_shim()
        L = C.CDLL(SHIM_LIB)
        L.hs_create.restype = C.c_void_p
        L.hs_create.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        L.hs_create3.restype = C.c_void_p
        L.hs_create3.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.hs_create2.restype = C.c_void_p
        L.hs_create2.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        L.hs_destroy.argtypes = [C.c_void_p]
        L.hs_write_mem.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
        L.hs_read_mem.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
        L.hs_predecode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.hs_set_pc.argtypes = [C.c_void_p, C.c_uint64]
        L.hs_set_x.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.hs_set_f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.hs_poke_csr.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.hs_run.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.hs_set_stop.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        for n in ("hs_get_pc", "hs_stop_value", "hs_exec", "hs_retired", "hs_thash", "hs_state_hash", "hs_mem_hash"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_void_p]
        L.hs_priv.argtypes = [C.c_void_p]
        for n in ("hs_get_x", "hs_get_f", "hs_csr"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_void_p, C.c_uint32]
        _shim = L
    return _shim


def result_tuple(status, value, pc, execn, retired, state_hash, mem_hash, trace_hash):
    return dict(status=status, value=value, pc=pc, exec=execn, retired=retired, state_hash=state_hash, mem_hash=mem_hash,
                trace_hash=trace_hash)


def run_oracle(prog, seed_a0=None, seed_mem=None, max_inst=None, log_cap=0, hart_id=0, features=0, max_ret=None, stop_pc=None):
    h = rvoracle.Hart(isa=prog.isa, mem_base=prog.mem_base, mem_size=prog.mem_size, tohost=prog.tohost, reset_pc=prog.entry,
                      hart_id=hart_id, tlb_entries=prog.meta.get("tlb_entries", 0), vlenb=prog.meta.get("vlenb", 0),
                      features=features or prog.meta.get("features", 0))
    if max_ret is not None or stop_pc is not None:
        h.set_stop(2**64 - 1 if max_ret is None else max_ret, stop_pc)
    for paddr, data, _ in prog.segments:
        h.write_mem(paddr, data)
    if seed_a0 is not None:
        h.set_x(10, seed_a0)
    if seed_mem is not None:
        h.write_mem(prog.symbols["seed"], seed_mem.to_bytes(8, "little"))
    st, log = h.run(prog.max_inst if max_inst is None else max_inst, log_cap=log_cap)
    r = result_tuple(st, h.stop_value, h.pc, h.exec_count, h.retired, h.state_hash, h.mem_hash, h.trace_hash)
    return r, h, log


last_fast_tr_steps = 0


def run_hostsim(prog, seed_a0=None, seed_mem=None, max_inst=None, predecode=True, hart_id=0, features=0, max_ret=None, stop_pc=None):
    L = shim()
    s = L.hs_create3(rvoracle.isa_flags(prog.isa) | (512 if prog.isa.lower().startswith("rv32") else 0) | (features or prog.meta.get("features", 0)), prog.mem_base, prog.mem_size, prog.tohost if prog.tohost is not None else 2**64 - 1, hart_id,
                     prog.meta.get("tlb_entries", 0), prog.meta.get("vlenb", 0))
    try:
        for paddr, data, _ in prog.segments:
            assert L.hs_write_mem(s, paddr, data, len(data)) == 0
        if predecode:
            x = [(p, len(d)) for p, d, f in prog.segments if f & 1]
            lo = min(p for p, _ in x); hi = max(p + n for p, n in x)
            L.hs_predecode(s, lo, hi - lo)
        L.hs_set_pc(s, prog.entry)
        if max_ret is not None or stop_pc is not None:
            L.hs_set_stop(s, 2**64 - 1 if max_ret is None else max_ret, 2**64 - 1 if stop_pc is None else stop_pc)
        if seed_a0 is not None:
            L.hs_set_x(s, 10, seed_a0)
        if seed_mem is not None:
            L.hs_write_mem(s, prog.symbols["seed"], seed_mem.to_bytes(8, "little"), 8)
        st = L.hs_run(s, prog.max_inst if max_inst is None else max_inst, 1)
        r = result_tuple(st, L.hs_stop_value(s), L.hs_get_pc(s), L.hs_exec(s), L.hs_retired(s), L.hs_state_hash(s), L.hs_mem_hash(s),
                         L.hs_thash(s))
        xs = [L.hs_get_x(s, i) for i in range(32)]
        fs = [L.hs_get_f(s, i) for i in range(32)]
        global last_fast_tr_steps
        L.hs_fast_tr_steps.restype = C.c_uint64
        L.hs_fast_tr_steps.argtypes = [C.c_void_p]
        last_fast_tr_steps = L.hs_fast_tr_steps(s)          # steps the TLB-hit fast step took in this run
        return r, xs, fs
    finally:
        L.hs_destroy(s)


def explain_diff(prog, ro, h, xs, fs):
    out = []
    for i in range(32):
        if h.x(i) != xs[i]:
            out.append(f"x{i}: oracle {h.x(i):#x} dev {xs[i]:#x}")
        if h.f(i) != fs[i]:
            out.append(f"f{i}: oracle {h.f(i):#x} dev {fs[i]:#x}")
    return "; ".join(out[:12])
