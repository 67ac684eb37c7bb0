#!/usr/bin/env python3
"""Experiment: does the tail of the distinct-program Sv39 fuzz (harts that run to the instruction limit through trap loops, alone
in their wave) run faster when the wave's other lanes execute copies of the same hart?  Same programs, R replicas each in
adjacent lanes (lanes per wave = R), against R = 1 at the same lanes per wave.
    python tools/exp_replicas.py [programs=1600] [R=16]"""
import json, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
nprog = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nv = 50
progs = []
for g in range(nprog):
    progs.append(WL.variant(WL.sv39_program(g // nv), g % nv))
p0 = progs[0]
for rep, lanes in (((R, R), (1, R), (1, 1)) if len(sys.argv) < 4 else [(1, int(x)) for x in sys.argv[3].split(',')]):
    n = nprog * rep
    b = W.Batch(n, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=lanes, tlb_entries=p0.meta.get("tlb_entries", 0))
    try:
        if rep == 1:
            b.load_elfs([p.elf for p in progs], lo=0)
        else:
            for i, p in enumerate(progs):
                b.load_elf_bytes(p.elf, i * rep, (i + 1) * rep)
        b.set_stop(max_inst=p0.max_inst)
        left = b.run(trace_hash=bool(os.environ.get('EXP_TRACE')))
        ks, _ = b.run_stats()
        res = b.results()
        insts = sum(r.exec_count for r in res)
        print(json.dumps(dict(programs=nprog, replicas=rep, lanes=lanes, harts=n, left=left, launches=b.run_stats()[1], kernel_ms=ks * 1e3, instructions=insts, gips=insts / ks / 1e9,
                              long_harts=sum(1 for r in res if r.exec_count >= p0.max_inst))), flush=True)
    finally:
        b.close()
