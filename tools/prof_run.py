#!/usr/bin/env python3
"""Tuning aid: run one general-path workload on a -DWSP_PROF build of the library (WSP_AMD_LIB) and print the cycle accounting.
    WSP_AMD_LIB=whisper_amd/lib/libwhisper_amd_prof.so python tools/prof_run.py rvv 4096 1"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W
from whisper_amd import workloads as WL
wl, n, lanes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if wl == "rvvk":
    k = 1; reps = n; progs = [WL.rvv_kernel(1024, 2)]
else:
    gen = {"sv39": WL.sv39_program, "sv39v": lambda s: WL.variant(WL.sv39_program(s // 50), s % 50), "rvv": WL.rvv_program, "fuzz": lambda s: WL.fuzz_program(s, 400)}[wl]
    k = min(n, 256 if wl != 'sv39v' else n); reps = n // k; n = k * reps
    progs = [gen(s) for s in range(k)]
p0 = progs[0]
b = W.Batch(n, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=lanes, tlb_entries=p0.meta.get("tlb_entries", 0), vlen_bytes=p0.meta.get("vlenb", 0))
for i, p in enumerate(progs):
    b.load_elf_bytes(p.elf, i * reps, (i + 1) * reps); b.set_stop(max_inst=p.max_inst, lo=i * reps, hi=(i + 1) * reps)
out = (C.c_uint64 * 16)()
W.lib().wsp_debug_prof(out, 1)
if wl == 'rvvk':
    b.poke_each(W.RES_X, 10, [WL.compute_seed(i) for i in range(n)])
b.run()
ks, _ = b.run_stats()
W.lib().wsp_debug_prof(out, 1)
insts = sum(r.exec_count for r in b.results())
names = ["kernel_wave_cycles", "cold_calls", "cold_cycles", "cold_loop_cycles", "vec_calls", "vec_cycles", "cold_steps", "vls_calls", "vls_cycles", "vwarm_calls", "vwarm_cycles", "vwarm_handled", "swarm_cycles", "x13", "warm_cycles", "hot_cycles"]
d = {nm: out[i] for i, nm in enumerate(names)}
d.update(workload=wl, harts=n, lanes=lanes, instructions=insts, kernel_ms=ks * 1e3, gips=insts / ks / 1e9)
print(json.dumps(d))
