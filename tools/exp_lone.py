#!/usr/bin/env python3
"""Experiment: a hart that runs to the instruction limit, ALONE in a wave of L lanes from the start (its wave-mates are stopped after
10 instructions), against the same harts one per wave."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
nprog = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
nv = 50
progs = [WL.variant(WL.sv39_program(g // nv), g % nv) for g in range(nprog)]
p0 = progs[0]
long_set = None
for lanes in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,4,64").split(",")]:
    for limit_others in (False, True):
        if limit_others and long_set is None:
            continue
        b = W.Batch(nprog, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=lanes, tlb_entries=p0.meta.get("tlb_entries", 0))
        try:
            b.load_elfs([p.elf for p in progs], lo=0)
            b.set_stop(max_inst=p0.max_inst)
            if limit_others:
                lo = 0
                for h in sorted(long_set) + [nprog]:
                    if h > lo: b.set_stop(max_inst=10, lo=lo, hi=h)
                    lo = h + 1
            b.run()
            ks, launches = b.run_stats()
            res = b.results()
            if long_set is None:
                long_set = set(i for i, r in enumerate(res) if r.exec_count >= p0.max_inst)
            print(json.dumps(dict(lanes=lanes, others_stopped_at_10=limit_others, kernel_ms=ks * 1e3, long_harts=len(long_set),
                                  instructions=sum(r.exec_count for r in res))), flush=True)
        finally:
            b.close()
