#!/usr/bin/env python3
"""Experiment: uniform waves of R replicas of one program; all replicas but the first are stopped after 10 instructions."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
nprog = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nv = 50
progs = [WL.variant(WL.sv39_program(g // nv), g % nv) for g in range(nprog)]
p0 = progs[0]
for R in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,4").split(",")]:
    for stop_others in (False, True):
        n = nprog * R
        b = W.Batch(n, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=R, tlb_entries=p0.meta.get("tlb_entries", 0))
        try:
            for i, p in enumerate(progs):
                b.load_elf_bytes(p.elf, R * i, R * i + R)
            b.set_stop(max_inst=p0.max_inst)
            if stop_others:
                for i in range(nprog):
                    b.set_stop(max_inst=10, lo=R * i + 1, hi=R * i + R)
            b.run()
            ks, launches = b.run_stats()
            print(json.dumps(dict(programs=nprog, replicas=R, others_stopped_at_10=stop_others, kernel_ms=ks * 1e3)), flush=True)
        finally:
            b.close()
