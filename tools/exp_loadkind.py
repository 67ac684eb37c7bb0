#!/usr/bin/env python3
"""Experiment: 4 different programs per wave, loaded hart by hart (wsp_batch_load_elf_mem) against loaded in bulk (wsp_batch_load_elf_many)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
nprog = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nv = 50
progs = [WL.variant(WL.sv39_program(g // nv), g % nv) for g in range(nprog)]
p0 = progs[0]
for bulk in (True, False):
    b = W.Batch(nprog, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=lanes, tlb_entries=p0.meta.get("tlb_entries", 0))
    try:
        if bulk:
            b.load_elfs([p.elf for p in progs], lo=0)
        else:
            for i, p in enumerate(progs):
                b.load_elf_bytes(p.elf, i, i + 1)
        b.set_stop(max_inst=p0.max_inst)
        b.run()
        ks, launches = b.run_stats()
        res = b.results()
        print(json.dumps(dict(programs=nprog, lanes=lanes, bulk_load=bulk, kernel_ms=ks * 1e3, instructions=sum(r.exec_count for r in res))), flush=True)
    finally:
        b.close()
