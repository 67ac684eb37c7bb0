#!/usr/bin/env python3
"""Experiment: what does cutting a run into launches of N steps cost?  One hart per wave (uniform batch: the library itself does not
slice), the caller slices with run(max_steps = N)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
nprog = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
nv = 50
progs = [WL.variant(WL.sv39_program(g // nv), g % nv) for g in range(nprog)]
p0 = progs[0]
for N in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,4096,16384").split(",")]:
    b = W.Batch(nprog, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=int(os.environ.get("EXP_LANES", "1")), tlb_entries=p0.meta.get("tlb_entries", 0))
    try:
        b.load_elfs([p.elf for p in progs], lo=0)
        b.set_stop(max_inst=p0.max_inst)
        ks, launches, left = 0.0, 0, 1
        while left:
            left = b.run(N if N else W.NO_LIMIT)
            s, l = b.run_stats(); ks += s; launches += l
        print(json.dumps(dict(slice=N, launches=launches, kernel_ms=ks * 1e3)), flush=True)
    finally:
        b.close()
