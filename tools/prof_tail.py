#!/usr/bin/env python3
"""Cycle accounting (-DWSP_PROF build through WSP_AMD_LIB) of the distinct-program Sv39 fuzz at a given lanes-per-wave layout.
    WSP_AMD_LIB=.../libwhisper_amd_prof.so python tools/prof_tail.py 1600 1,4"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
nprog = int(sys.argv[1]); nv = 50
progs = [WL.variant(WL.sv39_program(g // nv), g % nv) for g in range(nprog)]
p0 = progs[0]
names = ["kernel_wave_cycles", "cold_calls", "cold_cycles", "cold_loop_cycles", "vec_calls", "vec_cycles", "cold_steps", "vls_calls", "vls_cycles", "vwarm_calls", "vwarm_cycles", "vwarm_handled", "swarm_cycles", "x13", "warm_cycles", "hot_cycles"]
for lanes in [int(x) for x in sys.argv[2].split(",")]:
    b = W.Batch(nprog, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=lanes, tlb_entries=p0.meta.get("tlb_entries", 0))
    try:
        b.load_elfs([p.elf for p in progs], lo=0)
        b.set_stop(max_inst=p0.max_inst)
        out = (C.c_uint64 * 16)()
        W.lib().wsp_debug_prof(out, 1)
        b.run()
        ks, launches = b.run_stats()
        W.lib().wsp_debug_prof(out, 1)
        d = {nm: out[i] for i, nm in enumerate(names) if out[i]}
        d.update(lanes=lanes, launches=launches, kernel_ms=ks * 1e3)
        print(json.dumps(d), flush=True)
    finally:
        b.close()
