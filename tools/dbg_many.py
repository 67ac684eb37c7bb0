import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import whisper_amd as W
from whisper_amd import workloads as WL
progs = [WL.sv39_program(s) for s in range(4)]
p0 = progs[0]
bs = []
for many in (False, True):
    b = W.Batch(8, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, tlb_entries=8)
    if many:
        b.load_elfs([p.elf for p in progs], harts_per_elf=2)
    else:
        for i, p in enumerate(progs):
            b.load_elf_bytes(p.elf, 2 * i, 2 * i + 2)
    bs.append(b)
for h in range(8):
    p = progs[h // 2]
    for paddr, data, fl in p.segments:
        m0 = bs[0].peek(h, W.RES_MEM, paddr, len(data)); m1 = bs[1].peek(h, W.RES_MEM, paddr, len(data))
        if m0 != data: print("one-by-one mem differs", h, hex(paddr))
        if m1 != data:
            k = next(i for i in range(len(data)) if m1[i] != data[i])
            print("many mem differs", h, hex(paddr), "first at", k, "count", sum(a != b for a, b in zip(m1, data)))
    print(h, hex(bs[0].peek(h, W.RES_PC, 0)), hex(bs[1].peek(h, W.RES_PC, 0)))
for b in bs:
    b.set_stop(max_inst=200)
    print(b.run(trace_hash=True), [ (r.status, r.exec_count, hex(r.pc)) for r in b.results()])
