#!/usr/bin/env python3
"""Diagnostic sweep (not the bench contract): aggregate GIPS of the step kernel against the hart count and the
lanes-per-wave layout, same compute-kernel workload as bench.py.  One JSON line per point on stdout.

    python tools/sweep.py --harts 4096,16384,65536,262144 --lanes 0 --iters 1
    python tools/sweep.py --harts 4096 --lanes 1,2,4,8,16 --iters 4

Every point checks three harts against nothing but each other's pass status (parity proper lives in tests/); the
oracle spot check of bench.py is repeated for hart 0 unless --no-check.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--harts", default="4096,16384,65536,262144")
    ap.add_argument("--lanes", default="0")
    ap.add_argument("--iters", type=int, default=1)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--workload", default="compute", choices=["compute", "compute_int", "compute_sv39", "compute_int_sv39", "sv39", "sv39x", "rvv", "rvvk", "fuzz"])
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--vlenb", type=int, default=64)
    ap.add_argument("--n", type=int, default=1024, help="rvvk: elements per array")
    args = ap.parse_args()

    import whisper_amd as W
    from whisper_amd import workloads as WL

    for n in [int(x) for x in args.harts.split(",")]:
        for lanes in [int(x) for x in args.lanes.split(",")]:
            if args.workload in ("compute", "compute_int", "compute_sv39", "compute_int_sv39"):
                prog = WL.compute_kernel(iters=args.iters, fp="int" not in args.workload, paged=args.workload.endswith("sv39"))
                progs, reps = [prog], n
            elif args.workload == "rvvk":
                prog = WL.rvv_kernel(n=args.n, iters=args.iters, vlenb=args.vlenb)
                progs, reps = [prog], n
            else:
                gen = {"sv39": WL.sv39_program, "sv39x": lambda s: WL.sv39_program(s, exec_body=True), "rvv": WL.rvv_program, "fuzz": lambda s: WL.fuzz_program(s, 400)}[args.workload]
                k = min(n, 256)
                progs, reps = [gen(s) for s in range(k)], n // k
                n = k * reps
            p0 = progs[0]
            b = W.Batch(n, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=lanes,
                        tlb_entries=p0.meta.get("tlb_entries", 0), vlen_bytes=p0.meta.get("vlenb", 0))
            try:
                t0 = time.perf_counter()
                if len(progs) == 1:
                    b.load_elf_bytes(p0.elf)
                    b.set_stop(max_inst=p0.max_inst)
                    b.poke_each(W.RES_X, 10, [WL.compute_seed(i) for i in range(n)])
                    b.poke_each(W.RES_MEM, p0.symbols["seed"], list(range(n)))
                else:
                    for i, p in enumerate(progs):
                        b.load_elf_bytes(p.elf, i * reps, (i + 1) * reps)
                        b.set_stop(max_inst=p.max_inst, lo=i * reps, hi=(i + 1) * reps)
                load_s = time.perf_counter() - t0
                b.snapshot()
                best = None
                for _ in range(args.reps):
                    b.restore()
                    t0 = time.perf_counter()
                    left = b.run(trace_hash=args.trace)
                    wall = time.perf_counter() - t0
                    ks, launches = b.run_stats()
                    if best is None or ks < best[0]:
                        best = (ks, wall, launches)
                res = b.results()
                insts = sum(r.exec_count for r in res)
                ok = left == 0 and all(r.status != W.ST_RUNNING for r in res)
                check = None
                if not args.no_check and len(progs) == 1:
                    import simutil as S
                    ro, _, _ = S.run_oracle(p0, seed_a0=WL.compute_seed(0), seed_mem=0, hart_id=0)
                    if not args.trace:
                        ro = dict(ro); ro["trace_hash"] = 0
                    check = ro == res[0].as_dict()
                print(json.dumps({"workload": args.workload, "vlenb": p0.meta.get("vlenb", 0), "harts": n, "lanes_per_wave": lanes, "iters": args.iters, "instructions": insts,
                                  "kernel_ms": best[0] * 1e3, "wall_ms": best[1] * 1e3, "launches": best[2], "kernel_gips": insts / best[0] / 1e9,
                                  "load_s": load_s, "all_stopped": ok, "oracle_hart0": check, "trace": args.trace}), flush=True)
            finally:
                b.close()


if __name__ == "__main__":
    main()
