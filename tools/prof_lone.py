#!/usr/bin/env python3
"""Cycle / call accounting (-DWSP_PROF build through WSP_AMD_LIB) of harts that run on alone in waves of 4: wave-mates (stopped after
10 instructions) are other programs, or separate images of the same program."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
nprog = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nv = 50
progs = [WL.variant(WL.sv39_program(g // nv), g % nv) for g in range(nprog)]
p0 = progs[0]
names = ["kernel_wave_cycles", "cold_calls", "cold_cycles", "cold_loop_cycles", "vec_calls", "vec_cycles", "cold_steps", "vls_calls", "vls_cycles", "vwarm_calls", "vwarm_cycles", "vwarm_handled", "swarm_cycles", "x13", "warm_cycles", "hot_cycles"]
def run(elfs, keep, tag):
    n = len(elfs)
    b = W.Batch(n, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=4, tlb_entries=p0.meta.get("tlb_entries", 0))
    try:
        b.load_elfs(elfs, lo=0)
        b.set_stop(max_inst=p0.max_inst)
        if keep is not None:
            lo = 0
            for h in sorted(keep) + [n]:
                if h > lo: b.set_stop(max_inst=10, lo=lo, hi=h)
                lo = h + 1
        out = (C.c_uint64 * 16)()
        W.lib().wsp_debug_prof(out, 1)
        b.run()
        ks, launches = b.run_stats()
        W.lib().wsp_debug_prof(out, 1)
        res = b.results()
        d = {nm: out[i] for i, nm in enumerate(names) if out[i]}
        d.update(case=tag, launches=launches, kernel_ms=ks * 1e3, instructions=sum(r.exec_count for r in res))
        print(json.dumps(d), flush=True)
        return res
    finally:
        b.close()
res = run([p.elf for p in progs], None, "different programs, all run")
longs = [i for i, r in enumerate(res) if r.exec_count >= p0.max_inst]
run([p.elf for p in progs], set(longs), "different programs, only the long harts run on")
# the same long programs, each in lane 0 of a wave whose other lanes hold separate images of it
elfs = []
for i in longs:
    elfs += [progs[i].elf] * 4
run(elfs, set(range(0, len(elfs), 4)), "separate images of the long programs, lane 0 runs on")
