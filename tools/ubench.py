#!/usr/bin/env python3
"""Diagnostic (not the bench contract): what one interpreter step of each instruction class costs.  Each program is a loop whose
body is 64 instructions of one class plus the loop branch, run by every hart of a batch; printed per class: simulated GIPS and
the clocks one wave spends per step (clocks = n_waves_per_SIMD_turn / rate, 2.4 GHz nominal).

    python tools/ubench.py [--harts 4096] [--lanes 0] [--iters 2000] [--classes add,lw,...]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from whisper_amd.rvasm import Asm, CSR          # noqa: E402
from whisper_amd import workloads as WL         # noqa: E402

BASE = WL.BASE


def make(kind, iters):
    data_base = BASE + 0x4000
    a, d = Asm(BASE), Asm(data_base)
    d.label("tohost"); d.dword(0); d.label("fromhost"); d.dword(0); d.label("seed"); d.dword(0)
    d.label("buf"); d.zeros(4096)
    S = d.symbols
    a.label("_start")
    a.li(5, 0x6000); a.csrs(CSR["mstatus"], 5)
    a.li(8, S["buf"]); a.li(9, iters); a.li(12, 3); a.li(13, 5); a.li(14, 7)
    a.fmv_d_x(1, 0); a.fmv_d_x(2, 0)
    a.label("loop")
    for k in range(64):
        r = 16 + (k & 7)
        if kind == "add": a.add(r, 12, 13)
        elif kind == "addi": a.addi(r, r, 1)
        elif kind == "addchain": a.add(16, 16, 13)
        elif kind == "xor": a.xor(r, 12, 13)
        elif kind == "addw": a.addw(r, 12, 13)
        elif kind == "sll": a.slli(r, 12, 3)
        elif kind == "mul": a.mul(r, 12, 13)
        elif kind == "mulw": a.mulw(r, 12, 13)
        elif kind == "lw": a.lw(r, 4 * (k & 15), 8)
        elif kind == "ld": a.ld(r, 8 * (k & 15), 8)
        elif kind == "ldstride": a.ld(r, 64 * (k & 31), 8)
        elif kind == "sw": a.sw(12, 4 * (k & 15), 8)
        elif kind == "sd": a.sd(12, 8 * (k & 15), 8)
        elif kind == "bne_nt": a.bne(12, 12, "loop")                # never taken
        elif kind == "beq_t":                                        # always taken, to the next instruction
            a.beq(0, 0, a.pc + 4)
        elif kind == "jal":
            a.jal(0, a.pc + 4)
        elif kind == "fadd": a.fadd_d(3, 1, 2)
        elif kind == "fld": a.fld(3, 8 * (k & 15), 8)
        elif kind == "mulh": a.mulh(r, 12, 13)
        elif kind == "div": a.div(r, 12, 13)
        elif kind == "csrr": a.csrr(r, CSR["mscratch"])
        elif kind == "c_addi": a.c_addi(r if False else 12, 1)
        elif kind == "mix":                                          # the matrix-multiply inner loop's mix
            [lambda: a.lw(16, 0, 8), lambda: a.lw(17, 64, 8), lambda: a.mulw(16, 16, 17), lambda: a.addw(18, 18, 16),
             lambda: a.addi(19, 19, 4), lambda: a.addi(20, 20, 64), lambda: a.addi(21, 21, 1), lambda: a.addi(22, 0, 16)][k & 7]()
        else: raise SystemExit(f"unknown class {kind}")
    a.addi(9, 9, -1)
    a.bnez(9, "loop")
    WL._exit_tohost(a, S["tohost"])
    return WL._finish(a, "ubench-" + kind, "rv64imafdc_zicsr", d, mem_size=1 << 16, max_inst=iters * 66 + 4096)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--harts", type=int, default=4096)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--classes", default="add,addi,addchain,xor,addw,sll,mul,mulw,lw,ld,ldstride,sw,sd,bne_nt,beq_t,jal,fld,fadd,mulh,div,csrr,mix")
    args = ap.parse_args()
    import whisper_amd as W
    out = {}
    for kind in args.classes.split(","):
        it = args.iters if kind not in ("fadd", "mulh", "div", "csrr") else max(1, args.iters // 20)
        p = make(kind, it)
        b = W.Batch(args.harts, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size, lanes_per_wave=args.lanes)
        try:
            b.load_elf_bytes(p.elf)
            b.set_stop(max_inst=p.max_inst)
            left = b.run(trace_hash=args.trace)
            ks, launches = b.run_stats()
            res = b.results()
            insts = sum(r.exec_count for r in res)
            ok = left == 0 and all(r.status == W.ST_STOP_TOHOST for r in res)
            per_hart = insts / args.harts
            rec = dict(kind=kind, harts=args.harts, lanes=args.lanes, ok=ok, insts_per_hart=per_hart, kernel_ms=ks * 1e3, gips=insts / ks / 1e9,
                       ns_per_step=ks / per_hart * 1e9, clocks_per_step_2p4GHz=ks / per_hart * 2.4e9)
            print(json.dumps(rec), flush=True)
            out[kind] = rec
        finally:
            b.close()


if __name__ == "__main__":
    main()
