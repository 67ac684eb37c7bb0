#!/usr/bin/env python3
"""Experiment: the same set of different Sv39 fuzz programs in generator order (the 50 code variants of a generated program sit in
adjacent harts: programs that run to the instruction limit come in runs) against the same set dealt out so that variants of one
generated program land in different waves."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
nprog = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
nv = 50
progs = [WL.variant(WL.sv39_program(g // nv), g % nv) for g in range(nprog)]
nb = nprog // nv
spread = [progs[(i % nb) * nv + i // nb] for i in range(nprog)]
p0 = progs[0]
for lanes in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,16,64").split(",")]:
    for tag, lst in (("generator order", progs), ("variants dealt out over the waves", spread)):
        b = W.Batch(nprog, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=lanes, tlb_entries=p0.meta.get("tlb_entries", 0))
        try:
            b.load_elfs([p.elf for p in lst], lo=0)
            b.set_stop(max_inst=p0.max_inst)
            b.run()
            ks, launches = b.run_stats()
            res = b.results()
            longw = len(set(i // lanes for i, r in enumerate(res) if r.exec_count >= p0.max_inst))
            print(json.dumps(dict(programs=nprog, lanes=lanes, order=tag, kernel_ms=ks * 1e3, long_harts=sum(1 for r in res if r.exec_count >= p0.max_inst),
                                  waves_with_long_harts=longw, gips=sum(r.exec_count for r in res) / ks / 1e9)), flush=True)
        finally:
            b.close()
