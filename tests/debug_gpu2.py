"""Debug helper: bisect first divergence of one hart inside a multi-hart wave."""
import sys
import simutil as S
import whisper_amd as W
from whisper_amd import workloads as WL

target = int(sys.argv[1]); lo_h = (target // 8) * 8
progs = [WL.fuzz_program(s, n=300) for s in range(lo_h, lo_h + 8)]
ti = target - lo_h


def gpu(k):
    p0 = progs[0]
    b = W.Batch(8, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=8)
    for i, p in enumerate(progs):
        b.load_elf_bytes(p.elf, i, i + 1)
        b.set_stop(max_inst=(k if i == ti else p.max_inst), lo=i, hi=i + 1)
    b.run(trace_hash=True)
    r = b.results()[ti].as_dict()
    xs = [b.peek(ti, W.RES_X, i) for i in range(32)]; fs = [b.peek(ti, W.RES_F, i) for i in range(32)]
    mem = b.peek(ti, W.RES_MEM, p0.mem_base, p0.mem_size)
    b.close()
    return r, xs, fs, mem


def orc(k, cap=0):
    p = progs[ti]
    q = WL.Program(**{**p.__dict__}); q.max_inst = k
    return S.run_oracle(q, log_cap=cap)


full, _, _ = orc(progs[ti].max_inst)
lo, hi = 0, full["exec"]
print("full equal", gpu(hi)[0] == full)
while lo < hi:
    mid = (lo + hi) // 2
    if orc(mid)[0] == gpu(mid)[0]: lo = mid + 1
    else: hi = mid
k = lo
ro, h, log = orc(k, cap=k + 1)
rg, xs, fs, mem = gpu(k)
print("first divergence after", k, ro, rg)
for c in log[-3:]:
    print("  ", S.rvoracle.format_commit(c, disasm=S.rvoracle.ref_disassemble(c.inst)))
for i in range(32):
    if h.x(i) != xs[i]: print("   x%d oracle %x gpu %x" % (i, h.x(i), xs[i]))
    if h.f(i) != fs[i]: print("   f%d oracle %x gpu %x" % (i, h.f(i), fs[i]))
om = h.read_mem(progs[ti].mem_base, progs[ti].mem_size)
d = [i for i in range(len(om)) if om[i] != mem[i]]
print("   mem diffs:", [(hex(progs[ti].mem_base + i), om[i], mem[i]) for i in d[:16]])
