#!/usr/bin/env python3
"""Experiment: waves of 4 harts that run the SAME program, loaded as 4 separate program images (the library sees a wave of
different programs), against the same programs loaded once per wave (a uniform wave)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
nprog = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nv = 50
progs = [WL.variant(WL.sv39_program(g // nv), g % nv) for g in range(nprog)]
p0 = progs[0]
for separate in (True, False):
    n = nprog * 4
    b = W.Batch(n, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=4, tlb_entries=p0.meta.get("tlb_entries", 0))
    try:
        if separate:
            b.load_elfs([p.elf for p in progs for _ in range(4)], lo=0)
        else:
            for i, p in enumerate(progs):
                b.load_elf_bytes(p.elf, 4 * i, 4 * i + 4)
        b.set_stop(max_inst=p0.max_inst)
        if os.environ.get('EXP_STOP_OTHERS'):
            for i in range(nprog):
                keep = int(os.environ['EXP_STOP_OTHERS']) & 3
                for l in range(4):
                    if l != keep: b.set_stop(max_inst=10, lo=4 * i + l, hi=4 * i + l + 1)
        b.run(trace_hash=bool(os.environ.get("EXP_TRACE")))
        ks, launches = b.run_stats()
        print(json.dumps(dict(programs=nprog, separate_images=separate, kernel_ms=ks * 1e3)), flush=True)
    finally:
        b.close()
