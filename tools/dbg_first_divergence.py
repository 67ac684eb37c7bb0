#!/usr/bin/env python3
"""Debug aid: run one program on N harts with instruction limits 1..N and report the first limit at which the device's
trace / state digest differs from the oracle's, with the oracle's commit record there.
    python tools/dbg_first_divergence.py rv32 4        (workload generator name, seed)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simutil as S
import whisper_amd as W
from whisper_amd import workloads as WL
from simutil import rvoracle
kind, seed = sys.argv[1], int(sys.argv[2])
p = {"rv32": WL.rv32_program, "rv32f": lambda s: WL.rv32_program(s, fp=True), "fuzz": WL.fuzz_program, "rvv": WL.rvv_program}[kind](seed)
full, _, log = S.run_oracle(p, log_cap=100000)
n = min(full["exec"], 4096)
b = W.Batch(n, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size, vlen_bytes=p.meta.get("vlenb", 0))
b.load_elf_bytes(p.elf)
for i in range(n):
    b.set_stop(max_inst=i + 1, lo=i, hi=i + 1)
b.run(trace_hash=True)
res = [r.as_dict() for r in b.results()]
for i in range(n):
    ro, _, _ = S.run_oracle(p, max_inst=i + 1, hart_id=i)
    if ro != res[i]:
        c = log[i]
        print("first divergence at instruction", i + 1, {k: (ro[k], res[i][k]) for k in ro if ro[k] != res[i][k]})
        print("oracle record: pc %#x inst %#x size %d trap %d changes %s" % (c.pc, c.inst, c.size, c.trap, [(k, hex(a), hex(v)) for k, a, v in c.changes()]))
        if rvoracle.ref() is not None:
            print("disassembly:", rvoracle.ref_disassemble(c.inst if c.size == 4 else c.inst & 0xffff, rv64=not p.isa.startswith("rv32")) if hasattr(rvoracle, "ref_disassemble") else "")
        break
else:
    print("no divergence in", n, "instructions")
