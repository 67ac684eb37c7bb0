"""Minimal ELF64 little-endian RISC-V writer (and reader for tests).

The reference front end loads programs with ELFIO (Memory.cpp:687); the build image has no RISC-V
toolchain, so the synthetic workloads are assembled by rvasm.Asm and wrapped into real ELF files here
so that the ELF path of the C ABI (wsp_batch_load_elf) is what the tests and bench exercise.
"""
from __future__ import annotations

import struct

PF_X, PF_W, PF_R = 1, 2, 4
EM_RISCV = 243


def write_elf(segments, entry, symbols=None, sections=None) -> bytes:
    """segments: list of (paddr, bytes, flags); symbols: {name: addr}; sections: {name: addr}
    (extra empty named sections, e.g. '.tdata')."""
    symbols = symbols or {}
    sections = sections or {}
    ehsize, phentsize, shentsize = 64, 56, 64
    phnum = len(segments)
    off = ehsize + phnum * phentsize
    seg_offsets = []
    blob = bytearray()
    for paddr, data, flags in segments:
        pad = (-off - len(blob)) % 16
        blob += bytes(pad)
        seg_offsets.append(off + len(blob))
        blob += data
    # string tables
    shstr = bytearray(b"\0")

    def shname(s):
        o = len(shstr)
        shstr.extend(s.encode() + b"\0")
        return o

    strtab = bytearray(b"\0")
    symtab = bytearray(24)  # null symbol
    for name, addr in symbols.items():
        o = len(strtab)
        strtab += name.encode() + b"\0"
        symtab += struct.pack("<IBBHQQ", o, 0x10, 0, 0xFFF1, addr & (2**64 - 1), 0)  # GLOBAL NOTYPE, SHN_ABS
    sec_hdrs = [bytes(shentsize)]
    base = off + len(blob)
    pad = (-base) % 8
    blob += bytes(pad)
    symtab_off = off + len(blob)
    blob += symtab
    strtab_off = off + len(blob)
    blob += strtab
    n_symtab, n_strtab, n_extra = shname(".symtab"), shname(".strtab"), {k: shname(k) for k in sections}
    n_shstr = shname(".shstrtab")
    shstr_off = off + len(blob)
    blob += shstr
    # section headers: null, symtab(link->strtab idx 2), strtab, extras..., shstrtab
    sec_hdrs.append(struct.pack("<IIQQQQIIQQ", n_symtab, 2, 0, 0, symtab_off, len(symtab), 2, 1, 8, 24))
    sec_hdrs.append(struct.pack("<IIQQQQIIQQ", n_strtab, 3, 0, 0, strtab_off, len(strtab), 0, 0, 1, 0))
    for k, addr in sections.items():
        sec_hdrs.append(struct.pack("<IIQQQQIIQQ", n_extra[k], 1, 2, addr, 0, 0, 0, 0, 8, 0))
    sec_hdrs.append(struct.pack("<IIQQQQIIQQ", n_shstr, 3, 0, 0, shstr_off, len(shstr), 0, 0, 1, 0))
    pad = (-(off + len(blob))) % 8
    blob += bytes(pad)
    shoff = off + len(blob)
    shnum = len(sec_hdrs)
    eh = b"\x7fELF" + bytes([2, 1, 1, 0]) + bytes(8)
    eh += struct.pack("<HHIQQQIHHHHHH", 2, EM_RISCV, 1, entry, ehsize, shoff, 0x5, ehsize, phentsize, phnum, shentsize,
                      shnum, shnum - 1)
    ph = b""
    for (paddr, data, flags), so in zip(segments, seg_offsets):
        ph += struct.pack("<IIQQQQQQ", 1, flags, so, paddr, paddr, len(data), len(data), 16)
    return bytes(eh + ph + blob + b"".join(sec_hdrs))


def write_elf32(segments, entry, symbols=None) -> bytes:
    """ELF32 little-endian (an RV32 program, as riscv32 gcc writes for README.md:167-185): segments + one symbol table."""
    symbols = symbols or {}
    ehsize, phentsize, shentsize = 52, 32, 40
    phnum = len(segments)
    off = ehsize + phnum * phentsize
    blob = bytearray()
    seg_offsets = []
    for paddr, data, flags in segments:
        blob += bytes((-off - len(blob)) % 16)
        seg_offsets.append(off + len(blob))
        blob += data
    shstr = bytearray(b"\0.symtab\0.strtab\0.shstrtab\0")
    strtab = bytearray(b"\0")
    symtab = bytearray(16)
    for name, addr in symbols.items():
        o = len(strtab)
        strtab += name.encode() + b"\0"
        symtab += struct.pack("<IIIBBH", o, addr & 0xFFFFFFFF, 0, 0x10, 0, 0xFFF1)
    blob += bytes((-(off + len(blob))) % 8)
    symtab_off = off + len(blob); blob += symtab
    strtab_off = off + len(blob); blob += strtab
    shstr_off = off + len(blob); blob += shstr
    blob += bytes((-(off + len(blob))) % 8)
    shoff = off + len(blob)
    sec = [bytes(shentsize),
           struct.pack("<IIIIIIIIII", 1, 2, 0, 0, symtab_off, len(symtab), 2, 1, 4, 16),
           struct.pack("<IIIIIIIIII", 9, 3, 0, 0, strtab_off, len(strtab), 0, 0, 1, 0),
           struct.pack("<IIIIIIIIII", 17, 3, 0, 0, shstr_off, len(shstr), 0, 0, 1, 0)]
    eh = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8)
    eh += struct.pack("<HHIIIIIHHHHHH", 2, EM_RISCV, 1, entry & 0xFFFFFFFF, ehsize, shoff, 0x1, ehsize, phentsize, phnum, shentsize, len(sec), len(sec) - 1)
    ph = b""
    for (paddr, data, flags), so in zip(segments, seg_offsets):
        ph += struct.pack("<IIIIIIII", 1, so, paddr, paddr, len(data), len(data), flags, 16)
    return bytes(eh + ph + blob + b"".join(sec))


def read_elf(elf: bytes):
    """Returns (entry, [(paddr, bytes, flags)], {symbol: addr}) of a little-endian ELF64 or ELF32 image, as the C++ loader reads it."""
    if elf[:4] != b"\x7fELF" or elf[4] not in (1, 2) or elf[5] != 1:
        raise ValueError("not a little-endian ELF32 / ELF64 image")
    wide = elf[4] == 2
    if wide:
        (_, _, _, entry, phoff, shoff, _, _, phentsize, phnum, shentsize, shnum, shstrndx) = struct.unpack_from("<HHIQQQIHHHHHH", elf, 16)
    else:
        (_, _, _, entry, phoff, shoff, _, _, phentsize, phnum, shentsize, shnum, shstrndx) = struct.unpack_from("<HHIIIIIHHHHHH", elf, 16)
    segs = []
    for i in range(phnum):
        if wide:
            t, fl, o, va, pa, fsz, msz, al = struct.unpack_from("<IIQQQQQQ", elf, phoff + i * phentsize)
        else:
            t, o, va, pa, fsz, msz, fl, al = struct.unpack_from("<IIIIIIII", elf, phoff + i * phentsize)
        if t == 1 and fsz:
            segs.append((pa, elf[o:o + fsz], fl))
    syms = {}
    if wide:
        shs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    else:
        shs = [struct.unpack_from("<IIIIIIIIII", elf, shoff + i * shentsize) for i in range(shnum)]
    for sh in shs:
        if sh[1] == 2:                                       # SHT_SYMTAB: name type flags addr offset size link info addralign entsize
            st = shs[sh[6]]
            esz = 24 if wide else 16
            for k in range(sh[5] // esz):
                if wide:
                    n, info, other, shndx, val, size = struct.unpack_from("<IBBHQQ", elf, sh[4] + k * esz)
                else:
                    n, val, size, info, other, shndx = struct.unpack_from("<IIIBBH", elf, sh[4] + k * esz)
                end = elf.index(b"\0", st[4] + n)
                name = elf[st[4] + n:end].decode()
                if name:
                    syms[name] = val
    return entry, segs, syms
