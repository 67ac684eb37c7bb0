"""whisper_amd: MI355X-native batched RISC-V hart execution (the Hart::run hot path of
tenstorrent/whisper for many independent harts).  This module is a thin ctypes mirror of the C ABI
in include/whisper_amd.h; the simulator itself is HIP (whisper_amd/csrc).  There is no CPU
execution path: creating a Batch without a HIP device raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("WSP_AMD_LIB") or os.path.join(HERE, "lib", "libwhisper_amd.so")   # WSP_AMD_LIB: A/B builds while tuning
SRC = os.path.join(HERE, "csrc", "wsp_lib.hip")

ST_RUNNING, ST_STOP_TOHOST, ST_STOP_ILLEGAL, ST_STOP_LIMIT, ST_STOP_UNSUPPORTED, ST_STOP_ADDR = range(6)
RES_PC, RES_X, RES_F, RES_CSR, RES_MEM, RES_V = range(6)
RUN_TRACE_HASH = 1
LOAD_EXEC = 1
OK, EINVAL, ENODEV, ENOMEM, EIO, EFORMAT, ERANGE, EUNSUPPORTED, EHIP = 0, -1, -2, -3, -4, -5, -6, -7, -8
NO_LIMIT = 2**64 - 1


class WspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"whisper_amd error {code}: {msg}")
        self.code = code


class BatchDesc(C.Structure):
    _fields_ = [("xlen", C.c_uint32), ("n_harts", C.c_uint32), ("isa", C.c_char_p), ("mem_base", C.c_uint64),
                ("mem_size", C.c_uint64), ("lanes_per_wave", C.c_uint32), ("device", C.c_int32), ("hart_id_base", C.c_uint32),
                ("tlb_entries", C.c_uint32), ("vlen_bytes", C.c_uint32), ("reserved", C.c_uint32), ("config_json", C.c_char_p)]


class ConfigInfo(C.Structure):
    _fields_ = [("xlen", C.c_uint32), ("features", C.c_uint32), ("mem_size", C.c_uint64), ("reset_pc", C.c_uint64),
                ("has_reset_pc", C.c_uint32), ("tlb_entries", C.c_uint32), ("vlen_bytes", C.c_uint32), ("abi_names", C.c_uint32),
                ("n_csr_resets", C.c_uint32), ("reserved", C.c_uint32), ("csr_num", C.c_uint32 * 32), ("csr_value", C.c_uint64 * 32),
                ("isa", C.c_char * 256)]


class Result(C.Structure):
    _fields_ = [("status", C.c_uint32), ("priv", C.c_uint32), ("stop_value", C.c_uint64), ("pc", C.c_uint64),
                ("exec_count", C.c_uint64), ("retired_count", C.c_uint64), ("state_hash", C.c_uint64), ("mem_hash", C.c_uint64),
                ("trace_hash", C.c_uint64)]

    def as_dict(self):
        return dict(status=self.status, value=self.stop_value, pc=self.pc, exec=self.exec_count, retired=self.retired_count,
                    state_hash=self.state_hash, mem_hash=self.mem_hash, trace_hash=self.trace_hash)


class Decoded(C.Structure):
    _fields_ = [("w0", C.c_uint32), ("imm", C.c_int32), ("pad", C.c_uint32), ("raw", C.c_uint32)]


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    deps = [SRC, os.path.join(HERE, "csrc", "wsp_core.h"), os.path.join(HERE, "csrc", "wsp_conv_core.h"), os.path.join(HERE, "csrc", "wsp_conv_handlers.inc"), os.path.join(HERE, "csrc", "wsp_softfloat.h"), os.path.join(HERE, "csrc", "wsp_config.h"),
            os.path.join(HERE, "csrc", "wsp_disasm.h"), os.path.join(HERE, "csrc", "wsp_disasm_tables.inc"), os.path.join(ROOT, "include", "whisper_amd.h")]
    # fresh = built from exactly these sources: a digest of them is kept beside the library (file times do not survive copies of the
    # tree, and the device compile takes a quarter of an hour); a library without a digest file falls back to the file times
    import hashlib
    h = hashlib.sha256()
    for d in deps:
        with open(d, "rb") as fh:
            h.update(fh.read())
    digest = h.hexdigest()
    dpath = LIB_PATH + ".digest"
    if not force and os.path.exists(LIB_PATH):
        if os.path.exists(dpath):
            if open(dpath).read().strip() == digest:
                return LIB_PATH
        elif all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
            return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    # -structurizecfg-skip-uniform-regions: the interpreter's dispatch is a tree of wave-uniform branches; left to the
    # structurizer every level of it costs three extra scalar instructions (measured: 10.6 -> 11.9 GIPS, profiles/README.md)
    # one translation unit: the out-of-line general path dominates the device compile (~6 min) and every step kernel calls it, so
    # compiling the four kernels side by side repeats it four times (tried: no gain)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           "-mllvm", "-structurizecfg-skip-uniform-regions", "-o", LIB_PATH, SRC])
    with open(dpath, "w") as fh:
        fh.write(digest + "\n")
    return LIB_PATH


_lib = None


def lib():
    """Load the HIP extension; fails loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.wsp_batch_create.argtypes = [C.POINTER(BatchDesc), C.POINTER(vp)]
        L.wsp_batch_destroy.argtypes = [vp]
        L.wsp_batch_load_elf.argtypes = [vp, u32, u32, C.c_char_p]
        L.wsp_batch_load_elf_mem.argtypes = [vp, u32, u32, C.c_char_p, C.c_size_t]
        L.wsp_batch_load_elf_many.argtypes = [vp, u32, u32, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), u32]
        L.wsp_batch_load_image.argtypes = [vp, u32, u32, u64, C.c_char_p, C.c_size_t, u32]
        L.wsp_batch_symbol.argtypes = [vp, u32, C.c_char_p, C.POINTER(u64)]
        L.wsp_batch_poke.argtypes = [vp, u32, u32, u64, C.c_void_p, C.c_size_t]
        L.wsp_batch_peek.argtypes = [vp, u32, u32, u64, C.c_void_p, C.c_size_t]
        L.wsp_batch_poke_each.argtypes = [vp, u32, u32, u32, u64, C.POINTER(u64)]
        L.wsp_batch_set_stop.argtypes = [vp, u32, u32, u64, u64, u64, u64]
        L.wsp_config_parse.argtypes = [C.c_char_p, C.POINTER(ConfigInfo)]
        L.wsp_batch_run.restype = C.c_int64
        L.wsp_batch_run.argtypes = [vp, u64, u32]
        L.wsp_batch_last_run_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u32)]
        L.wsp_batch_results.argtypes = [vp, C.POINTER(Result)]
        L.wsp_batch_results_noalloc.argtypes = [vp, C.c_void_p]
        L.wsp_sf_eval.argtypes = [u32, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.c_size_t, C.POINTER(u64), C.POINTER(u32)]
        L.wsp_batch_last_hash_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64)]
        L.wsp_batch_signature.restype = C.c_int64
        L.wsp_batch_signature.argtypes = [vp, u32, C.POINTER(u32), C.c_size_t]
        L.wsp_batch_trace_ring.argtypes = [vp, u32]
        L.wsp_batch_trace_read.restype = C.c_int64
        L.wsp_batch_trace_read.argtypes = [vp, u32, C.POINTER(u64), C.c_size_t]
        L.wsp_batch_trace_text.restype = C.c_int64
        L.wsp_batch_trace_text.argtypes = [vp, u32, u64, C.c_char_p, C.c_size_t]
        L.wsp_disassemble.argtypes = [u32, u32, C.c_char_p, C.c_size_t]
        L.wsp_batch_resume.argtypes = [vp, u32, u32]
        L.wsp_batch_snapshot.argtypes = [vp]
        L.wsp_batch_restore.argtypes = [vp]
        L.wsp_decode_words.argtypes = [C.c_char_p, C.POINTER(u32), C.c_size_t, C.POINTER(Decoded)]
        L.wsp_decode_words_host.argtypes = [C.c_char_p, C.POINTER(u32), C.c_size_t, C.POINTER(Decoded)]
        L.wsp_isa_flags.restype = u32
        L.wsp_isa_flags.argtypes = [C.c_char_p]
        L.wsp_last_error.restype = C.c_char_p
        L.wsp_version.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise WspError(rc, lib().wsp_last_error().decode())
    return rc


class Batch:
    """N independent single-hart systems on one MI355X (mirrors the shape of pywhisper's
    System64/Hart64, py-bindings.cpp:330-469, for a batch)."""

    def __init__(self, n_harts, isa="rv64imafdc_zicsr", mem_base=0x80000000, mem_size=1 << 16, lanes_per_wave=0, device=-1,
                 hart_id_base=0, tlb_entries=0, vlen_bytes=0, config=None, xlen=0):
        """config: text of a Whisper JSON configuration file (or a dict, dumped to JSON); isa / mem_size = None / 0 take the file's."""
        self.L = lib()
        if isinstance(config, dict):
            import json
            config = json.dumps(config)
        self._cfg = config.encode() if isinstance(config, str) else config
        info = parse_config(self._cfg) if self._cfg else None
        self._isa = isa.encode() if isa else None
        desc = BatchDesc(xlen, n_harts, self._isa, mem_base, mem_size or 0, lanes_per_wave, device, hart_id_base, tlb_entries, vlen_bytes, 0, self._cfg)
        h = C.c_void_p()
        _check(self.L.wsp_batch_create(C.byref(desc), C.byref(h)))
        self.h = h
        self.n = n_harts
        eff_isa = isa or (info.isa.decode() if info else "")
        vb = vlen_bytes or (info.vlen_bytes if info else 0) or 64                              # Hart.cpp:1081 default
        self.vlen_bytes = vb if "v" in eff_isa.lower().split("_")[0][4:] else 0

    def close(self):
        if getattr(self, "h", None):
            self.L.wsp_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def _rng(self, lo, hi):
        return (0 if lo is None else lo, self.n if hi is None else hi)

    def load_elf(self, path, lo=None, hi=None):
        lo, hi = self._rng(lo, hi)
        _check(self.L.wsp_batch_load_elf(self.h, lo, hi, os.fsencode(path)))

    def load_elf_bytes(self, elf: bytes, lo=None, hi=None):
        lo, hi = self._rng(lo, hi)
        _check(self.L.wsp_batch_load_elf_mem(self.h, lo, hi, elf, len(elf)))

    def load_elfs(self, elfs, lo=0, harts_per_elf=1):
        """wsp_batch_load_elf_many: ELF image i -> harts [lo + i * harts_per_elf, lo + (i + 1) * harts_per_elf), one call."""
        n = len(elfs)
        ptrs = (C.c_char_p * n)(*elfs)
        sizes = (C.c_size_t * n)(*[len(e) for e in elfs])
        _check(self.L.wsp_batch_load_elf_many(self.h, lo, n, ptrs, sizes, harts_per_elf))

    def load_image(self, addr, data: bytes, lo=None, hi=None, executable=False):
        lo, hi = self._rng(lo, hi)
        _check(self.L.wsp_batch_load_image(self.h, lo, hi, addr, data, len(data), LOAD_EXEC if executable else 0))

    def symbol(self, hart, name):
        v = C.c_uint64()
        _check(self.L.wsp_batch_symbol(self.h, hart, name.encode(), C.byref(v)))
        return v.value

    def poke(self, hart, res, idx, value):
        if res in (RES_MEM, RES_V):
            buf = bytes(value)
            _check(self.L.wsp_batch_poke(self.h, hart, res, idx, buf, len(buf)))
        else:
            v = C.c_uint64(value & NO_LIMIT)
            _check(self.L.wsp_batch_poke(self.h, hart, res, idx, C.byref(v), 8))

    def peek(self, hart, res, idx, size=8):
        if res == RES_V:
            size = self.vlen_bytes
        if res in (RES_MEM, RES_V):
            buf = C.create_string_buffer(size)
            _check(self.L.wsp_batch_peek(self.h, hart, res, idx, buf, size))
            return buf.raw
        v = C.c_uint64()
        _check(self.L.wsp_batch_peek(self.h, hart, res, idx, C.byref(v), 8))
        return v.value

    def poke_each(self, res, idx, values, lo=None, hi=None):
        lo, hi = self._rng(lo, hi)
        arr = (C.c_uint64 * (hi - lo))(*[v & NO_LIMIT for v in values])
        _check(self.L.wsp_batch_poke_each(self.h, lo, hi, res, idx, arr))

    def set_stop(self, tohost=None, max_inst=NO_LIMIT, lo=None, hi=None, max_ret=NO_LIMIT, stop_pc=None):
        lo, hi = self._rng(lo, hi)
        _check(self.L.wsp_batch_set_stop(self.h, lo, hi, NO_LIMIT if tohost is None else tohost, max_inst, max_ret,
                                         NO_LIMIT if stop_pc is None else stop_pc))

    def resume(self, lo=None, hi=None):
        """wsp_batch_resume: harts stopped by a count limit or their stop address run again."""
        lo, hi = self._rng(lo, hi)
        _check(self.L.wsp_batch_resume(self.h, lo, hi))

    def run(self, max_steps=NO_LIMIT, trace_hash=False):
        return _check(self.L.wsp_batch_run(self.h, max_steps, RUN_TRACE_HASH if trace_hash else 0))

    def run_stats(self):
        s = C.c_double(); n = C.c_uint32()
        _check(self.L.wsp_batch_last_run_stats(self.h, C.byref(s), C.byref(n)))
        return s.value, n.value

    def results(self):
        arr = (Result * self.n)()
        _check(self.L.wsp_batch_results(self.h, arr))
        return arr

    def results_noalloc(self, device_ptr):
        """wsp_batch_results_noalloc: records written to caller-owned DEVICE memory (n * sizeof(Result) bytes at device_ptr)."""
        _check(self.L.wsp_batch_results_noalloc(self.h, C.c_void_p(device_ptr)))

    def hash_stats(self):
        s = C.c_double(); n = C.c_uint64()
        _check(self.L.wsp_batch_last_hash_stats(self.h, C.byref(s), C.byref(n)))
        return s.value, n.value

    def trace_ring(self, words_per_hart):
        """Switch the commit ring on (words per hart) or off (0); see include/whisper_amd.h."""
        _check(self.L.wsp_batch_trace_ring(self.h, words_per_hart))
        self._ring_cap = words_per_hart

    def trace(self, hart):
        """Commit records of one hart: [(pc, inst, size, priv, trap, [(kind, id, value), ...]), ...]; a record cut off by the
        end of the ring is dropped.  (The ring also holds each record's load / store address pair and vector register bytes:
        trace_text prints them.)"""
        buf = (C.c_uint64 * self._ring_cap)()
        n = _check(self.L.wsp_batch_trace_read(self.h, hart, buf, self._ring_cap))
        out, i = [], 0
        while i + 3 <= n:
            pc, meta, kinds = buf[i], buf[i + 1], buf[i + 2]
            npairs = meta >> 56
            pairs = [((kinds >> (4 * k)) & 15, buf[i + 3 + 2 * k], buf[i + 4 + 2 * k]) for k in range(npairs) if i + 4 + 2 * k < n]
            length = 3 + 2 * npairs + (2 if (meta >> 55) & 1 else 0) + sum(1 for p in pairs if p[0] == 3) * (self.vlen_bytes // 8)
            if i + length > n:
                break
            out.append((pc, meta & 0xFFFFFFFF, (meta >> 48) & 0x7F, (meta >> 32) & 0xFF, (meta >> 40) & 0xFF, pairs))
            i += length
        return out

    def trace_text(self, hart, first_tag=1):
        """wsp_batch_trace_text: the recorded commits of one hart in the reference's --logfile text."""
        n = _check(self.L.wsp_batch_trace_text(self.h, hart, first_tag, None, 0))
        buf = C.create_string_buffer(n + 1)
        _check(self.L.wsp_batch_trace_text(self.h, hart, first_tag, buf, n + 1))
        return buf.value.decode()

    def snapshot(self):
        _check(self.L.wsp_batch_snapshot(self.h))

    def restore(self):
        _check(self.L.wsp_batch_restore(self.h))

    def signature(self, hart, cap=1 << 16):
        arr = (C.c_uint32 * cap)()
        n = _check(self.L.wsp_batch_signature(self.h, hart, arr, cap))
        return list(arr[:n])


def parse_config(text):
    """wsp_config_parse: what a Whisper JSON configuration asks of a batch (raises WspError on tags the device does not model)."""
    if isinstance(text, str):
        text = text.encode()
    info = ConfigInfo()
    _check(lib().wsp_config_parse(text, C.byref(info)))
    return info


def decode_words(isa, words, host=False):
    L = lib()
    n = len(words)
    arr = (C.c_uint32 * n)(*words)
    out = (Decoded * n)()
    fn = L.wsp_decode_words_host if host else L.wsp_decode_words
    _check(fn(isa.encode(), arr, n, out))
    return out


def disassemble(inst, xlen=64):
    """wsp_disassemble: the log's disassembly column for one instruction word."""
    buf = C.create_string_buffer(256)
    _check(lib().wsp_disassemble(inst, xlen, buf, 256))
    return buf.value.decode()


def sf_eval(op, rm, a, b, c):
    """Device softfloat over operand lists (wsp_sf_eval): returns (results, flags)."""
    L = lib()
    n = len(a)
    A = (C.c_uint64 * n)(*a); B = (C.c_uint64 * n)(*b); Cc = (C.c_uint64 * n)(*c)
    out = (C.c_uint64 * n)(); fl = (C.c_uint32 * n)()
    _check(L.wsp_sf_eval(op, rm, A, B, Cc, n, out, fl))
    return list(out), list(fl)
