"""The commit log as text (row f2: Hart::printDecodedInstTrace, printTrace.cpp:302-517; Disassembler.cpp).

CPU: wsp_disassemble word for word against the reference's own Disassembler class (oracle/_ref) over everything the device
decoder accepts, both widths, and against the disassembly column of the reference's captured logs.
GPU: wsp_batch_trace_text of whole runs against the text built from the oracle's commit records."""
import ctypes as C
import json
import os
import random
import re

import pytest

import simutil as S
import whisper_amd as W
from simutil import rvoracle
from whisper_amd import workloads as WL

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "appendix_d.json")))
needs_ref = pytest.mark.skipif(rvoracle.ref() is None, reason="oracle/_ref not built")


class Dec(C.Structure):
    _fields_ = [("w0", C.c_uint32), ("imm", C.c_int32), ("pad", C.c_uint32), ("raw", C.c_uint32)]


def _decodes(flags, words):
    L = S.shim()
    L.hs_decode.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(Dec)]
    n = len(words)
    out = (Dec * n)()
    L.hs_decode(flags, (C.c_uint32 * n)(*words), n, out)
    return [(out[i].w0 & 0xff) != 0 for i in range(n)]


@needs_ref
@pytest.mark.parametrize("rv64", [True, False])
def test_disassembler_matches_reference_disassembler(rv64):
    """Every 16-bit word, 150000 random 32-bit words and 40000 words drawn from the vector / system / atomic / FP major opcodes:
    for each one the device decoder accepts, wsp_disassemble == the reference Disassembler's text, character for character."""
    rng = random.Random(7 if rv64 else 8)
    flags = rvoracle.isa_flags("rv64imafdcvsu_zicsr")
    if not rv64:
        flags = (flags & ~(32 | 128)) | 512
    words = [rng.getrandbits(32) | 3 for _ in range(150000)] + list(range(65536))
    words += [(rng.getrandbits(25) << 7) | rng.choice([0x57, 0x07, 0x27, 0x73, 0x2f, 0x53, 0x0f]) for _ in range(40000)]
    words = [w for w in words if (w & 3) != 3 and w < 65536 or (w & 3) == 3 and (w & 0x1c) != 0x1c]
    ok = _decodes(flags, words)
    n = 0
    for w, legal in zip(words, ok):
        if not legal:
            continue
        n += 1
        assert W.disassemble(w, 64 if rv64 else 32) == rvoracle.ref_disassemble(w, rv64=rv64), hex(w)
    assert n > 80000


def test_disassembly_column_of_the_reference_logs():
    """The disassembly column of every record of the reference's captured logs (FP + traps, Sv39, RVV, RV32)."""
    seen = 0
    for key, xlen in (("fp_trap", 64), ("sv39", 64), ("rvv", 64), ("rv32", 32)):
        for r in GOLD[key]["records"]:
            if r["op"] == "0000":
                continue
            want = re.split(r" \[0x| \((?:exception|interrupted)\)|  \+", r["text"])[0].rstrip()
            # (the captured logs lost the padding operand-less mnemonics end in; the fuzz test above pins it)
            assert W.disassemble(int(r["op"], 16), xlen).rstrip() == want, (key, r["op"])
            seen += 1
    for v in GOLD["rvv"]["vlines"]:
        assert W.disassemble(int(v["op"], 16), 64).rstrip() == v["text"]
    assert seen > 20


def _expected_text(p, log, dev_text, hart_ix=0):
    """The log text the oracle's commit records stand for; vector register bytes are taken from the device's own lines after
    their digest has been checked against the record."""
    xlen = 32 if p.isa.startswith("rv32") else 64
    has_d = "d" in p.isa.split("_")[0][4:]
    dev_lines = dev_text.split("\n")
    vb = {}
    li = 0
    out = []
    L = rvoracle.lib()
    for c in log:
        vbytes = {}
        nlines = c.nchanges
        for k, (kind, key, val) in enumerate(c.changes()):
            if kind == rvoracle.CH_V:
                m = re.match(r"^#\d+ \d +[MSU] [0-9a-f]+ +[0-9a-f]+ v ([0-9a-f]{2}) ([0-9a-f]+) ", dev_lines[li + k])
                assert m and int(m.group(1), 16) == key, dev_lines[li + k]
                data = bytes.fromhex(m.group(2))[::-1]
                d = 0
                for i in range(0, len(data), 8):
                    d = L.rvo_fold(d, int.from_bytes(data[i:i + 8], "little"))
                assert d == val, (c.tag, key)
                vbytes[key] = data
        dis = W.disassemble(c.inst if c.size == 4 else c.inst & 0xffff, xlen)
        out.extend(rvoracle.format_commit(c, hart_ix, dis, xlen=xlen, has_d=has_d, vbytes=vbytes).split("\n"))
        li += nlines
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fuzz", "sv39", "rvv", "rv32", "rv32f"])
def test_trace_text_matches_oracle_log(kind):
    """wsp_batch_trace_text of a whole run == the text of the oracle's commit log, line for line: columns of both widths, r / f / v /
    m / c lines and their continuation marks, load / store address suffixes (virtual : physical under Sv39), trap marks."""
    gen = {"fuzz": lambda s: WL.fuzz_program(s, 300), "sv39": WL.sv39_program, "rvv": lambda s: WL.rvv_program(s, n=120),
           "rv32": WL.rv32_program, "rv32f": lambda s: WL.rv32_program(s, fp=True)}[kind]
    for seed in (1, 2):
        p = gen(seed)
        ro, h, log = S.run_oracle(p, log_cap=60000)
        b = W.Batch(2, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size, tlb_entries=p.meta.get("tlb_entries", 0), vlen_bytes=p.meta.get("vlenb", 0))
        try:
            b.load_elf_bytes(p.elf)
            b.set_stop(max_inst=p.max_inst)
            b.trace_ring(1 << 21)
            assert b.run(trace_hash=True) == 0
            assert b.results()[0].as_dict() == ro
            text = b.trace_text(0)
        finally:
            b.close()
        got = text.rstrip("\n").split("\n")
        want = _expected_text(p, log, text)
        assert len(got) == len(want), (kind, seed, len(got), len(want))
        for i, (g, w) in enumerate(zip(got, want)):
            assert g == w, (kind, seed, i)
        if kind == "sv39":
            assert any(re.search(r"\[0x[0-9a-f]+:0x[0-9a-f]+\]", ln) for ln in got)
        if kind == "rvv":
            assert any(" v " in ln for ln in got)
