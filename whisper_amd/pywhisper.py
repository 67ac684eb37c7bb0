"""`pywhisper`-shaped Python surface over the batch C ABI (the reference's pybind11 module, /root/reference/py-bindings.cpp:74-486):

    from whisper_amd import pywhisper
    sys_ = pywhisper.system.System64("cfg.json", memory_size=1 << 20, memory_base=0x80000000)
    sys_.load_elf_files(["prog.elf"])
    hart = sys_.harts()[0]
    finished, di, changes = hart.step()          # ['x5', 'mstatus', ('m2147491840', 8)] as py-bindings.cpp:338-386 builds it
    hart.x10 = 5; print(hart.pc, hart.mstatus, hart.int_regs); hart.run()
    sys_.memory().peek32(0x80002000)

Same class, method and attribute names, argument meaning and error behaviour (RuntimeError / AttributeError / TypeError /
ValueError where the reference throws std::runtime_error / py::attribute_error / py::type_error / std::invalid_argument).  What is
different, because this library runs batches of independent single-hart systems on a GPU:
  * a System is ONE hart with its private memory; core_count * harts_per_core > 1 (harts sharing a memory) is refused.  The
    extension argument `instances=N` makes N independent replicas of the system in one batch: harts() then returns the N harts and
    memory(i) the memory of instance i;
  * `memory_base` (extension, default 0) places the memory window; memory_size defaults to the reference's 4 GiB only on paper -
    pass what the program needs (each instance owns that much HBM);
  * step() is a one-instruction run of the batch's step kernel with the commit ring on; the change list comes from the ring record.
"""
from __future__ import annotations

import enum
import sys as _sys
import types

import whisper_amd as W


class PrivilegeMode(enum.IntEnum):            # trapEnums.hpp (py-bindings.cpp:78-83)
    User = 0
    Supervisor = 1
    Reserved = 2
    Machine = 3


class File:
    """pywhisper.file.File: an open FILE* the reference hands to Hart::singleStep / run for the log."""

    def __init__(self, f):
        self.file_ = f


def _file_open(path):
    try:
        return File(open(path, "w"))
    except OSError:
        print("Error: file::open: failed to open file", file=_sys.stderr)
        return File(None)


def _file_close(f):
    if f.file_ is None:
        raise RuntimeError("file::close: failed to close file")
    f.file_.close()


class DecodedInst:
    """decoded_inst.DecodedInst (py-bindings.cpp:124-137): what step() returns about the instruction it executed."""

    def __init__(self, inst=0, address=0, phys=0, rec=None):
        self._inst, self._addr, self._phys, self._rec = inst, address, phys, rec

    def inst(self):
        return self._inst

    def address(self):
        return self._addr

    def phys_address(self):
        return self._phys

    def _field(self, i):
        if self._rec is None:
            raise RuntimeError("DecodedInst: no decoded record")
        w0 = self._rec.w0
        return [(w0 >> 8) & 31, (w0 >> 13) & 31, (w0 >> 18) & 31, (w0 >> 23) & 31][i]

    def ith_operand(self, i):
        """Operand i as the device's pre-decoded record holds it: 0 rd, 1 rs1, 2 rs2 or the immediate, 3 rs3."""
        if i == 2 and self._rec is not None and (self._rec.w0 >> 29) & 1:
            return self._rec.imm
        return self._field(i)

    def ith_operand_type(self, i):
        return "imm" if (i == 2 and self._rec is not None and (self._rec.w0 >> 29) & 1) else "reg"

    def ith_operand_mode(self, i):
        return "write" if i == 0 else "read"


_INT_ABI = ["zero", "ra", "sp", "gp", "tp", "t0", "t1", "t2", "s0", "s1", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "s2", "s3", "s4",
            "s5", "s6", "s7", "s8", "s9", "s10", "s11", "t3", "t4", "t5", "t6"]
_FP_ABI = ["ft0", "ft1", "ft2", "ft3", "ft4", "ft5", "ft6", "ft7", "fs0", "fs1", "fa0", "fa1", "fa2", "fa3", "fa4", "fa5", "fa6", "fa7", "fs2",
           "fs3", "fs4", "fs5", "fs6", "fs7", "fs8", "fs9", "fs10", "fs11", "ft8", "ft9", "ft10", "ft11"]
_csr_by_name = None


def _csr_names():
    """CSR name -> number, names as the reference's disassembler prints them (wsp_disassemble of `csrrs x0, <csr>, x0`)."""
    global _csr_by_name
    if _csr_by_name is None:
        d = {}
        for num in range(4096):
            text = W.disassemble(0x00002073 | (num << 20))
            parts = [t.strip() for t in text.split(None, 1)[1].split(",")] if " " in text else []
            if len(parts) == 3 and not parts[1].startswith("0x"):
                d.setdefault(parts[1], num)
        _csr_by_name = d
    return _csr_by_name


def _find_reg(name, prefix, abi):
    if name.startswith(prefix) and name[len(prefix):].isdigit() and int(name[len(prefix):]) < 32:
        return int(name[len(prefix):])
    if abi and name in abi:
        return abi.index(name)
    if prefix == "x" and name == "fp":
        return 8
    return None


class _Hart:
    _own = ("_sys", "_ix", "_b")

    def __init__(self, system, ix):
        object.__setattr__(self, "_sys", system)
        object.__setattr__(self, "_ix", ix)
        object.__setattr__(self, "_b", system._batch)

    # ---- Hart::singleStep / run (py-bindings.cpp:338-389)
    def step(self, verbose=False, file=None):
        """Step a single instruction.  Returns (target program finished, DecodedInst, list of modified resources)."""
        b, h = self._b, self._ix
        self._sys._ensure_ring()
        st = self._sys._status(h)
        if st in (W.ST_STOP_TOHOST, W.ST_STOP_ILLEGAL, W.ST_STOP_UNSUPPORTED):
            return (st == W.ST_STOP_TOHOST, DecodedInst(), [])
        execd = b.peek(h, W.RES_PC, 1)
        b.set_stop(max_inst=execd + 1, lo=h, hi=h + 1, stop_pc=None)
        b.resume(h, h + 1)
        self._sys._run_only(h)
        recs = b.trace(h)
        changes, di = [], DecodedInst()
        if recs:
            pc, inst, size, priv, trap, pairs = recs[-1]
            names = {v: k for k, v in _csr_names().items()}
            dec = W.decode_words(self._sys._isa, [inst], host=True)[0]
            di = DecodedInst(inst, pc, pc, dec)
            ireg = [p for p in pairs if p[0] == 1]; freg = [p for p in pairs if p[0] == 2]; vreg = [p for p in pairs if p[0] == 3]
            csrs = [p for p in pairs if p[0] == 5]; mem = [p for p in pairs if p[0] == 4]
            changes += ["x%d" % p[1] for p in ireg if p[1] > 0]
            changes += ["f%d" % p[1] for p in freg]
            changes += ["v%d" % p[1] for p in vreg]
            changes += [names.get(p[1], hex(p[1])) for p in csrs]
            for p in mem:
                changes.append(("m%d" % p[1], self._sys._store_size(dec)))
            if verbose:
                out = (file.file_ if isinstance(file, File) else file) or _sys.stdout
                text = b.trace_text(h, first_tag=execd + 1)
                out.write(text)
        self._sys._clear_ring()
        return (self._sys._status(h) == W.ST_STOP_TOHOST, di, changes)

    def run(self, verbose=False, file=None):
        """Run until the hart reaches a stopping point."""
        b, h = self._b, self._ix
        if self._sys._status(h) in (W.ST_STOP_LIMIT, W.ST_STOP_ADDR):
            b.set_stop(max_inst=W.NO_LIMIT, lo=h, hi=h + 1)
            b.resume(h, h + 1)
        else:
            b.set_stop(max_inst=W.NO_LIMIT, lo=h, hi=h + 1)
        if verbose:
            self._sys._ensure_ring(big=True)
            first = b.peek(h, W.RES_PC, 1) + 1
        elif self._sys._ring:                                      # recording costs the fast paths: off for a plain run
            b.trace_ring(0); self._sys._ring = 0
        self._sys._run_only(h)
        if verbose:
            out = (file.file_ if isinstance(file, File) else file) or _sys.stdout
            out.write(b.trace_text(h, first_tag=first))
            self._sys._clear_ring()

    def disass_inst(self, inst):
        return W.disassemble(inst, self._sys._xlen)

    # ---- attribute access (py-bindings.cpp:140-327)
    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        b, h = self._b, self._ix
        r = _find_reg(name, "x", _INT_ABI)
        if r is not None:
            return b.peek(h, W.RES_X, r)
        r = _find_reg(name, "f", _FP_ABI)
        if r is not None:
            return b.peek(h, W.RES_F, r)
        r = _find_reg(name, "v", None)
        if r is not None:
            if not b.vlen_bytes:
                raise AttributeError("Hart::__getattr__: invalid vec register access")
            return list(b.peek(h, W.RES_V, r)[::-1])          # Hart::peekVecReg: most significant byte first (Hart.cpp:4884-4900)
        num = _csr_names().get(name)
        if num is not None:
            try:
                return b.peek(h, W.RES_CSR, num)
            except W.WspError:
                raise AttributeError("Hart::__getattr__: invalid csr access") from None
        if name == "pc":
            return b.peek(h, W.RES_PC, 0)
        if name == "int_regs":
            return [b.peek(h, W.RES_X, i) for i in range(32)]
        if name == "fp_regs":
            return [b.peek(h, W.RES_F, i) for i in range(32)] if self._sys._has("f") else []
        if name == "vec_regs":
            return [list(b.peek(h, W.RES_V, i)[::-1]) for i in range(32)] if b.vlen_bytes else []
        if name == "cs_regs":
            out = {}
            for nm, n in _csr_names().items():
                try:
                    out[nm] = b.peek(h, W.RES_CSR, n)
                except W.WspError:
                    pass
            return out
        if name == "priv_mode":
            return PrivilegeMode(b.peek(h, W.RES_PC, 3))
        if name == "virt_mode":
            return False
        raise AttributeError("Hart::__getattr__: unknown attribute")

    def __setattr__(self, name, value):
        b, h = self._b, self._ix

        def need_int(what):
            if not isinstance(value, int) or isinstance(value, bool):
                raise TypeError(f"Hart::__setattr__: must assign int to {what}")
        r = _find_reg(name, "x", _INT_ABI)
        if r is not None:
            need_int("int register"); b.poke(h, W.RES_X, r, value); return
        r = _find_reg(name, "f", _FP_ABI)
        if r is not None:
            need_int("fp register"); b.poke(h, W.RES_F, r, value); return
        r = _find_reg(name, "v", None)
        if r is not None:
            if not isinstance(value, list):
                raise TypeError("Hart::__setattr___: must assign list of ints to vec register")
            if not b.vlen_bytes or len(value) != b.vlen_bytes:
                raise AttributeError("Hart::__setattr___: invalid vec register access")
            b.poke(h, W.RES_V, r, bytes(value)[::-1]); return
        num = _csr_names().get(name)
        if num is not None:
            need_int("csr")
            try:
                b.poke(h, W.RES_CSR, num, value)
            except W.WspError:
                raise AttributeError("Hart::__setattr___: invalid csr access") from None
            return
        if name == "pc":
            need_int("pc"); b.poke(h, W.RES_PC, 0, value); return
        if name == "int_regs":
            if not isinstance(value, list):
                raise TypeError("Hart::__setattr___: must assign list of ints to int_regs")
            for i, v in enumerate(value):
                b.poke(h, W.RES_X, i, v)
            return
        if name == "fp_regs":
            if not isinstance(value, list):
                raise TypeError("Hart::__setattr___: must assign list of ints to fp_regs")
            for i, v in enumerate(value):
                b.poke(h, W.RES_F, i, v)
            return
        if name == "vec_regs":
            if not isinstance(value, list):
                raise TypeError("Hart::__setattr___: must assign list of list of ints to vec_regs")
            for i, v in enumerate(value):
                b.poke(h, W.RES_V, i, bytes(v)[::-1])
            return
        if name == "cs_regs":
            if not isinstance(value, dict):
                raise TypeError("Hart::__setattr___: must assign dict of name, value to cs_regs")
            for k, v in value.items():
                n = _csr_names().get(k)
                if n is None:
                    raise AttributeError("Hart::__setattr___: invalid csr access")
                try:
                    b.poke(h, W.RES_CSR, n, v)
                except W.WspError:
                    raise AttributeError("Hart::__setattr___: invalid csr access") from None
            return
        raise AttributeError("Hart::__setattr__: unknown attribute")


class Memory:
    """system.memory.Memory (py-bindings.cpp:105-121): peek / poke of one instance's memory, 1 / 2 / 4 / 8 bytes."""

    def __init__(self, batch, hart):
        self._b, self._h = batch, hart

    def _peek(self, address, size):
        try:
            return int.from_bytes(self._b.peek(self._h, W.RES_MEM, address, size), "little")
        except W.WspError:
            raise RuntimeError("Memory::peek: failed to peek memory") from None

    def _poke(self, address, data, size):
        try:
            self._b.poke(self._h, W.RES_MEM, address, (data & ((1 << (8 * size)) - 1)).to_bytes(size, "little"))
        except W.WspError:
            raise RuntimeError("Memory::poke: failed to poke memory") from None

    def peek8(self, address): return self._peek(address, 1)
    def peek16(self, address): return self._peek(address, 2)
    def peek32(self, address): return self._peek(address, 4)
    def peek64(self, address): return self._peek(address, 8)
    def poke8(self, address, data): self._poke(address, data, 1)
    def poke16(self, address, data): self._poke(address, data, 2)
    def poke32(self, address, data): self._poke(address, data, 4)
    def poke64(self, address, data): self._poke(address, data, 8)


class _System:
    _xlen = 64

    def __init__(self, config_file, core_count=1, harts_per_core=1, memory_size=1 << 32, page_size=4096, memory_base=0, instances=1,
                 device=-1):
        if core_count * harts_per_core != 1:
            raise ValueError("System::System: failed configuration (harts sharing one memory are not modelled: use instances=N "
                             "for N independent single-hart systems)")
        try:
            text = open(config_file).read() if config_file else ""
            info = W.parse_config(text) if text else None
        except (OSError, W.WspError):
            raise ValueError("System::System: failed configuration") from None
        if info is None or not info.isa:
            raise ValueError("System::System: failed configuration")          # config.getIsa(isa) must succeed (py-bindings.cpp:423)
        self._isa = info.isa.decode()
        if info.xlen and info.xlen != self._xlen:
            raise ValueError("System::System: failed configuration (xlen of the file does not match the class)")
        try:
            self._batch = W.Batch(instances, isa=None, mem_base=memory_base, mem_size=memory_size, config=text, xlen=self._xlen, device=device)
        except W.WspError:
            raise ValueError("System::System: failed configuration") from None
        self._n = instances
        self._ring = 0
        self._harts = [self._hart_class(self, i) for i in range(instances)]

    def _has(self, letter):
        return letter in self._isa.lower().split("_")[0][4:] or (letter == "f" and "g" in self._isa.lower().split("_")[0][4:])

    def _status(self, h):
        try:
            return self._batch.peek(h, W.RES_PC, 4)                  # stop state without computing the digests
        except W.WspError:
            return self._batch.results()[h].status

    def _ensure_ring(self, big=False):
        want = (1 << 20) if big else 4096
        if self._ring != want:
            self._batch.trace_ring(want); self._ring = want
        return want

    def _clear_ring(self):
        self._batch.trace_ring(self._ring)                       # calling it again clears the ring

    def _run_only(self, h):
        """Run hart h alone: the other instances are parked on an exhausted instruction limit meanwhile."""
        b = self._batch
        parked = []
        if self._n > 1:
            res = b.results()
            for k in range(self._n):
                if k != h and res[k].status == W.ST_RUNNING:
                    b.set_stop(max_inst=res[k].exec_count, lo=k, hi=k + 1); parked.append(k)
        b.run(trace_hash=self._ring != 0)
        for k in parked:
            b.set_stop(max_inst=W.NO_LIMIT, lo=k, hi=k + 1); b.resume(k, k + 1)

    @staticmethod
    def _store_size(dec):
        op = dec.w0 & 0xff
        return {44: 1, 45: 2, 46: 4, 47: 8}.get(op, 8 if (dec.w0 >> 30) & 1 else 4)

    # ---- loaders (py-bindings.cpp:442-453)
    def load_elf_files(self, files, verbose=False):
        try:
            for f in files:
                self._batch.load_elf(str(f))
            return True
        except W.WspError as e:
            print(f"Error: {e}", file=_sys.stderr)
            return False

    def load_hex_files(self, files, verbose=False):
        """Memory::loadHexFile (Memory.cpp:96-198): '@<hex address>' lines and lines of hex byte values."""
        ok = True
        for f in files:
            try:
                lines = open(f).read().splitlines()
            except OSError:
                print(f"Error: Failed to open hex-file '{f}' for input", file=_sys.stderr)
                return False
            addr, run, start = 0, bytearray(), 0

            def flush():
                if run:
                    self._batch.load_image(start, bytes(run), executable=True)
            for ln, line in enumerate(lines):
                line = line.strip()
                if not line:
                    continue
                if line[0] == "@":
                    try:
                        new = int(line[1:].split()[0], 16)
                    except (ValueError, IndexError):
                        print(f"Error: File {f}, Line {ln}: Invalid hexadecimal address: {line}", file=_sys.stderr); ok = False; continue
                    flush(); run = bytearray(); addr = start = new
                    continue
                for tok in line.split():
                    try:
                        v = int(tok, 16)
                    except ValueError:
                        print(f"Error: File {f}, Line {ln}: Invalid data: {line}", file=_sys.stderr); ok = False; break
                    if v > 0xff:
                        print(f"Error: File {f}, Line {ln}: Invalid value: {v:x}", file=_sys.stderr); ok = False
                    if not run:
                        start = addr
                    run.append(v & 0xff); addr += 1
            try:
                flush()
            except W.WspError as e:
                print(f"Error: {e}", file=_sys.stderr); ok = False
        return ok

    def load_bin_files(self, files, offset, verbose=False):
        try:
            for f in files:
                self._batch.load_image(offset, open(f, "rb").read(), executable=True)
            return True
        except (OSError, W.WspError) as e:
            print(f"Error: {e}", file=_sys.stderr)
            return False

    def harts(self):
        return list(self._harts)

    def memory(self, instance=0):
        return Memory(self._batch, instance)

    def close(self):
        self._batch.close()


class Hart64(_Hart):
    pass


class Hart32(_Hart):
    pass


class System64(_System):
    _xlen = 64
    _hart_class = Hart64


class System32(_System):
    _xlen = 32
    _hart_class = Hart32


# ---- submodule layout of the pybind11 module: pywhisper.file / .enums / .system(.memory, .hart) / .decoded_inst
file = types.ModuleType("pywhisper.file"); file.File = File; file.open = _file_open; file.close = _file_close
enums = types.ModuleType("pywhisper.enums"); enums.PrivilegeMode = PrivilegeMode
for _m in PrivilegeMode:
    setattr(enums, _m.name, _m)                                    # export_values()
decoded_inst = types.ModuleType("pywhisper.decoded_inst"); decoded_inst.DecodedInst = DecodedInst
system = types.ModuleType("pywhisper.system"); system.System64 = System64; system.System32 = System32
system.memory = types.ModuleType("pywhisper.system.memory"); system.memory.Memory = Memory
system.hart = types.ModuleType("pywhisper.system.hart"); system.hart.Hart64 = Hart64; system.hart.Hart32 = Hart32
