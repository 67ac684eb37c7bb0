"""Debug aid (not collected by pytest): smallest instruction limit at which an untraced replicated run of the compute kernel
differs from the C oracle for some hart of one wave, and the instruction that sits there.

    python tests/debug_bisect_conv.py [first_hart] [lanes] [n]"""
import sys

sys.path.insert(0, "tests")
import simutil as S
import whisper_amd as W
from whisper_amd import workloads as WL


def gpu(p, n, lanes, seeds, k):
    b = W.Batch(n, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size, lanes_per_wave=lanes)
    try:
        b.load_elf_bytes(p.elf, 0, n)
        b.set_stop(max_inst=k)
        b.poke_each(W.RES_X, 10, [s[0] for s in seeds])
        b.poke_each(W.RES_MEM, p.symbols["seed"], [s[1] for s in seeds])
        left = n
        while left:
            left = b.run(W.NO_LIMIT, trace_hash=False)
        res = [r.as_dict() for r in b.results()]
        xs = [[b.peek(h, W.RES_X, i) for i in range(32)] for h in range(n)]
        return res, xs
    finally:
        b.close()


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    p = WL.compute_kernel(iters=1)
    seeds = [(WL.compute_seed(i), i) for i in range(n)]
    harts = list(range(first, first + lanes))

    def differs(k):
        res, xs = gpu(p, n, lanes, seeds, k)
        out = []
        for h in harts:
            ro, hh, _ = S.run_oracle(p, seed_a0=seeds[h][0], seed_mem=seeds[h][1], max_inst=k, hart_id=h)
            ro = dict(ro); ro["trace_hash"] = 0
            if res[h] != ro:
                out.append((h, res[h], ro, xs[h], [hh.x(i) for i in range(32)]))
        return out

    lo, hi = 0, 1 << 26
    assert differs(hi), "no difference at the full length"
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if differs(mid):
            hi = mid
        else:
            lo = mid
    print("first differing limit:", hi)
    for h, got, want, xg, xo in differs(hi):
        print("hart", h, "\n got ", got, "\n want", want)
        for i in range(32):
            if xg[i] != xo[i]:
                print(f"   x{i}: got {xg[i]:#x} want {xo[i]:#x}")
        # the instruction executed last by the oracle: the one at the pc reached with limit hi - 1
        ro, hh, _ = S.run_oracle(p, seed_a0=seeds[h][0], seed_mem=seeds[h][1], max_inst=hi - 1, hart_id=h)
        pc = hh.pc
        seg = next(d for a, d, f in p.segments if a <= pc < a + len(d))
        base = next(a for a, d, f in p.segments if a <= pc < a + len(d))
        word = int.from_bytes(seg[pc - base:pc - base + 4], "little")
        print(f"   instruction #{hi} at pc {pc:#x}: {word:#010x}")


if __name__ == "__main__":
    main()
