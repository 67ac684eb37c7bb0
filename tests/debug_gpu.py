"""Debug helper (not a test): find the first instruction at which the GPU diverges from the oracle."""
import sys
import simutil as S
import whisper_amd as W
from whisper_amd import workloads as WL


def gpu_run(p, k, lanes=0):
    b = W.Batch(1, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size, lanes_per_wave=lanes)
    b.load_elf_bytes(p.elf)
    b.set_stop(max_inst=k)
    b.run(trace_hash=True)
    r = b.results()[0].as_dict()
    mem = b.peek(0, W.RES_MEM, p.mem_base, p.mem_size)
    xs = [b.peek(0, W.RES_X, i) for i in range(32)]
    fs = [b.peek(0, W.RES_F, i) for i in range(32)]
    b.close()
    return r, mem, xs, fs


def orc_run(p, k):
    q = WL.Program(**{**p.__dict__}); q.max_inst = k
    ro, h, log = S.run_oracle(q, log_cap=0)
    return ro, h


for seed in ([int(a) for a in sys.argv[1:]] if __name__ == "__main__" else []):
    p = WL.fuzz_program(seed, n=300)
    full, h = orc_run(p, p.max_inst)
    lo, hi = 0, full["exec"]
    r, mem, xs, fs = gpu_run(p, hi)
    print("seed", seed, "full equal:", r == full)
    while lo < hi:
        mid = (lo + hi) // 2
        ro, _ = orc_run(p, mid)
        rg, _, _, _ = gpu_run(p, mid)
        if ro == rg:
            lo = mid + 1
        else:
            hi = mid
    k = lo
    ro, h = orc_run(p, k)
    rg, mem, xs, fs = gpu_run(p, k)
    print(" first divergence after", k, "instructions", ro, rg)
    q = WL.Program(**{**p.__dict__}); q.max_inst = k
    _, h2, log = S.run_oracle(q, log_cap=k + 1)
    for c in log[-3:]:
        print("  ", S.rvoracle.format_commit(c, disasm=S.rvoracle.ref_disassemble(c.inst) if S.rvoracle.ref() else ""))
    for i in range(32):
        if h.x(i) != xs[i]: print("   x%d oracle %x gpu %x" % (i, h.x(i), xs[i]))
        if h.f(i) != fs[i]: print("   f%d oracle %x gpu %x" % (i, h.f(i), fs[i]))
    om = h.read_mem(p.mem_base, p.mem_size)
    diffs = [i for i in range(len(om)) if om[i] != mem[i]]
    print("   mem diffs:", [(hex(p.mem_base + i), om[i], mem[i]) for i in diffs[:16]])
