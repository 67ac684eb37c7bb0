"""Whisper snapshot directories (System::saveSnapshot / loadSnapshot, System.cpp:562-700, 1959-2100; Hart::saveSnapshotRegs /
loadSnapshotRegs, snapshot.cpp:17-100, 230-420; Memory::saveSnapshot_gzip / loadSnapshot_gzip, Memory.cpp:926-1060) for one hart
of a batch, in the reference's own on-disk format so that a run can move between Whisper and this library either way:

    <dir>/registers   text: pm / vm / po / pr / pb / pc / elp lines, then "x <n> 0x<v>" for the non-zero integer registers,
                      "f <n> 0x<v>", "c 0x<num> 0x<v>" (misa first, then the implemented CSRs in number order that are non-zero or
                      reset to non-zero), "v <n> 0x<bytes>" (most significant byte first)
    <dir>/usedblocks  "<address> <size>" per saved memory block (decimal)
    <dir>/memory      gzip stream: the bytes of the used blocks back to back
    <dir>/time        the time counter (decimal)
    <dir>/mtimecmp    "0x<v>" per hart

The device-resident wsp_batch_snapshot / wsp_batch_restore (a rewind inside one process) is separate.  Host code over the C ABI
(peek / poke): snapshots are taken at run boundaries, not inside the kernel.  Files of subsystems that are out of scope (fd, mmap,
data-lines, instr-lines, branch-trace, cache-trace, imsic, device state) are written empty / ignored on load.
"""
from __future__ import annotations

import gzip
import os

import whisper_amd as W

# CSRs with storage in the modelled file, in number order (the reference walks MIN_CSR_ .. MAX_CSR_); misa goes first
_CSRS = [0x001, 0x002, 0x003, 0x008, 0x009, 0x00A, 0x00F, 0x100, 0x104, 0x105, 0x106, 0x10A, 0x140, 0x141, 0x142, 0x143, 0x144, 0x180,
         0x300, 0x302, 0x303, 0x304, 0x305, 0x306, 0x30A, 0x320, 0x340, 0x341, 0x342, 0x343, 0x344] + \
        list(range(0x3A0, 0x3B0)) + list(range(0x3B0, 0x3F0)) + [0xB00, 0xB02, 0xC20, 0xC21]
_BLOCK = 1 << 20
# on load, views of another register are skipped (the register itself is in the file) and the vector CSRs that csr writes cannot
# change are poked last, as the reference defers them (snapshot.cpp:258, 380-420)
_VIEWS = {0x001, 0x002, 0x100, 0x104, 0x144, 0x009, 0x00A}


_DCSR_RESET = 0x40000003                                     # CsRegs.cpp:5328-5331: prv = M, the rest clear


def _derived_or_fixed(b, n, v):
    """CSR lines every snapshot written by the reference holds (saveSnapshotRegs walks all implemented CSRs that are non-zero or
    reset to non-zero, snapshot.cpp:72-83) but that carry no state of their own here: True = skip the line.  A value that would
    need state this library does not model raises instead of being dropped."""
    if n == 0x7B0:                                           # dcsr: debug mode is not modelled; anything but its reset value is state
        if v != _DCSR_RESET:
            raise ValueError(f"snapshot dcsr {v:#x}: debug-mode state is not modelled")
        return True
    if n in (0x7B1, 0x7B2, 0x7B3):                           # dpc, dscratch0/1: meaningful only in debug mode
        return True
    if n == 0xC22:                                           # vlenb: fixed by the batch
        if b.vlen_bytes and v != b.vlen_bytes:
            raise ValueError(f"snapshot vlenb {v} does not match the batch's vector register size {b.vlen_bytes}")
        return True
    if 0xC00 <= n <= 0xC1F or 0xC80 <= n <= 0xC9F:           # cycle / time / instret / hpmcounter shadows of the M-mode counters
        return True
    if n in (0xF11, 0xF12, 0xF13, 0xF14, 0xF15):             # mvendorid marchid mimpid mhartid mconfigptr: read-only identity
        return True
    return False


def _peek_csr(b, hart, n):
    try:
        return b.peek(hart, W.RES_CSR, n)
    except W.WspError:
        return None


def save(b: "W.Batch", hart: int, path: str, mem_base: int, mem_size: int, isa: str = "") -> None:
    """Hart `hart` of batch `b` -> snapshot directory `path`."""
    os.makedirs(path, exist_ok=True)
    xlen32 = isa.lower().startswith("rv32")
    lines = [f"pm {b.peek(hart, W.RES_PC, 3)}", "vm 0", f"po {b.peek(hart, W.RES_PC, 1)}", f"pr {b.peek(hart, W.RES_PC, 2)}", "pb 0x0",
             f"pc 0x{b.peek(hart, W.RES_PC, 0):x}", "elp 0"]
    for i in range(1, 32):
        v = b.peek(hart, W.RES_X, i)
        if v:
            lines.append(f"x {i} 0x{v:x}")
    letters = isa.lower().split("_")[0][4:]
    if "f" in letters or "g" in letters:
        for i in range(32):
            v = b.peek(hart, W.RES_F, i)
            if v:
                lines.append(f"f {i} 0x{v:x}")
    lines.append(f"c 0x301 0x{b.peek(hart, W.RES_CSR, 0x301):x}")
    for n in _CSRS:
        if xlen32 is False and 0x3A0 <= n < 0x3B0 and (n & 1):
            continue
        v = _peek_csr(b, hart, n)
        if v is None:
            continue
        if v == 0 and n != 0x300:                          # (mstatus resets to a non-zero value: always written)
            continue
        lines.append(f"c 0x{n:x} 0x{v:x}")
    if b.vlen_bytes:
        for i in range(32):
            data = b.peek(hart, W.RES_V, i)
            lines.append(f"v {i} 0x" + "".join(f"{x:02x}" for x in reversed(data)))
    with open(os.path.join(path, "registers"), "w") as f:
        f.write("\n".join(lines) + "\n")
    blocks = []
    with gzip.open(os.path.join(path, "memory"), "wb", compresslevel=1) as gz:
        for off in range(0, mem_size, _BLOCK):
            n = min(_BLOCK, mem_size - off)
            data = b.peek(hart, W.RES_MEM, mem_base + off, n)
            if any(data):
                blocks.append((mem_base + off, n))
                gz.write(data)
    with open(os.path.join(path, "usedblocks"), "w") as f:
        f.write("".join(f"{a} {n}\n" for a, n in blocks))
    with open(os.path.join(path, "time"), "w") as f:
        f.write(f"{b.peek(hart, W.RES_PC, 1)}\n")            # time_ ticks once per executed instruction (Hart.hpp:3101)
    with open(os.path.join(path, "mtimecmp"), "w") as f:
        f.write("0x0\n")
    for name in ("fd", "mmap"):
        open(os.path.join(path, name), "w").close()


def load(b: "W.Batch", hart: int, path: str, mem_base: int, mem_size: int) -> None:
    """Snapshot directory `path` (written by Whisper or by save) -> hart `hart` of batch `b`."""
    blocks = []
    with open(os.path.join(path, "usedblocks")) as f:
        for ln in f:
            if ln.strip():
                a, n = ln.split()
                blocks.append((int(a), int(n)))
    with gzip.open(os.path.join(path, "memory"), "rb") as gz:
        for a, n in blocks:
            data = gz.read(n)
            if len(data) != n:
                raise ValueError("snapshot memory file is shorter than its used blocks")
            if a < mem_base or a + n > mem_base + mem_size:
                raise ValueError(f"snapshot block {a:#x}+{n:#x} lies outside the hart arena")
            b.poke(hart, W.RES_MEM, a, data)
    deferred = []
    priv = 3
    with open(os.path.join(path, "registers")) as f:
        for ln in f:
            t = ln.split()
            if not t:
                continue
            if t[0] == "pc":
                b.poke(hart, W.RES_PC, 0, int(t[1], 0))
            elif t[0] == "po":
                b.poke(hart, W.RES_PC, 1, int(t[1], 0))
            elif t[0] == "pr":
                b.poke(hart, W.RES_PC, 2, int(t[1], 0))
            elif t[0] == "pm":
                priv = int(t[1], 0)
            elif t[0] == "x":
                b.poke(hart, W.RES_X, int(t[1], 0), int(t[2], 0))
            elif t[0] == "f":
                b.poke(hart, W.RES_F, int(t[1], 0), int(t[2], 0))
            elif t[0] == "c":
                n, v = int(t[1], 0), int(t[2], 0)
                if n == 0x301 or n in _VIEWS:
                    continue
                if n in (0xC20, 0xC21, 0x008, 0x00F):
                    deferred.append((n, v))
                    continue
                if _derived_or_fixed(b, n, v):
                    continue
                try:
                    b.poke(hart, W.RES_CSR, n, v)
                except W.WspError:
                    if v:                                   # a CSR outside the modelled file holding state: refuse, do not guess
                        raise
            elif t[0] == "v":
                hexs = t[2][2:]
                hexs = hexs if len(hexs) % 2 == 0 else "0" + hexs
                b.poke(hart, W.RES_V, int(t[1], 0), bytes.fromhex(hexs)[::-1])
            elif t[0] in ("vm", "pb", "elp"):
                if t[0] == "vm" and int(t[1], 0):
                    raise ValueError("snapshot taken in virtual (hypervisor) mode: not modelled")
            else:
                raise ValueError(f"unknown snapshot register tag {t[0]!r}")
    for n, v in deferred:
        b.poke(hart, W.RES_CSR, n, v)
    b.poke(hart, W.RES_PC, 3, priv)
