#!/usr/bin/env python3
"""Diagnostic (not the bench contract): what one vector instruction of each class costs on the vector warm path.  Each program
sets vl = VLMAX at (sew, lmul) once and loops over 32 instructions of one class; printed per class: simulated GIPS and the time
one wave spends per instruction.

    python tools/vbench.py [--harts 16384] [--vlenb 64] [--iters 20] [--classes vadd,vfmacc,...]
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
CLASSES = "vadd,vadd_e8,vmul_vx,vsrl_vi,vmax,vadd_masked,vmerge,vmslt,vle32,vse32,vle8,vfmacc_vf,vfmadd_vv64,vfmul,vfmul64,vfcvt,vsetvli,vredsum,vfredusum,vid,vmv_s_x,scalar"


def make(kind, iters, vlenb):
    data_base = BASE + 0x1000
    a, d = Asm(BASE), Asm(data_base)
    d.label("tohost"); d.dword(0); d.label("fromhost"); d.dword(0); d.label("seed"); d.dword(0)
    d.align_data(64)
    d.label("buf"); d.zeros(8 * vlenb * 2)
    S = d.symbols
    sew = 8 if kind.endswith("e8") or kind == "vle8" else 64 if kind.endswith("64") else 32
    a.label("_start")
    a.li(5, 0x6600); a.csrs(CSR["mstatus"], 5)
    a.li(8, S["buf"]); a.li(9, iters); a.li(12, 3); a.li(13, 5)
    a.fmv_d_x(2, 0); a.li(6, 0x3FC00000); a.fmv_w_x(1, 6)     # f1 = 1.5f, NaN-boxed
    a.vsetvli(15, 0, Asm.vtype(sew, 8, ta=True, ma=True))     # rs1 = x0, rd != x0: vl = VLMAX
    a.vop(0x14, 2, 8, 0, 0x11); a.vop(0x14, 2, 16, 0, 0x11); a.vop(0x14, 2, 24, 0, 0x11)   # vid.v
    a.vop(0x1B, 4, 0, 8, 13)                                   # vmslt.vx v0, v8, x13: a mask with a few bits set
    if kind.startswith("vf") and kind != "vfcvt":              # ordinary floating-point values (small integers), not the bit patterns of vid
        for v in (8, 16, 24):
            a.vop(0x12, 1, v, v, 3)                            # vfcvt.f.x.v
    a.label("loop")
    for k in range(32):
        if kind in ("vadd", "vadd_e8"): a.vop(0x00, 0, 24, 8, 16)
        elif kind == "vmul_vx": a.vop(0x25, 6, 24, 8, 12)
        elif kind == "vsrl_vi": a.vop(0x28, 3, 24, 8, 7)
        elif kind == "vmax": a.vop(0x07, 0, 24, 8, 16)
        elif kind == "vadd_masked": a.vop(0x00, 0, 24, 8, 16, 0)
        elif kind == "vmerge": a.vop(0x17, 4, 24, 8, 12, 0)
        elif kind == "vmslt": a.vop(0x1B, 4, 1, 8, 12)
        elif kind in ("vle32", "vle8"): a.vle(sew, 24, 8)
        elif kind == "vse32": a.vse(32, 24, 8)
        elif kind == "vfmacc_vf": a.vop(0x2C, 5, 24, 8, 1)
        elif kind == "vfmadd_vv64": a.vop(0x28, 1, 24, 8, 16)
        elif kind in ("vfmul", "vfmul64"): a.vop(0x24, 1, 24, 8, 16)
        elif kind == "vfcvt": a.vop(0x12, 1, 24, 8, 3)
        elif kind == "vsetvli": a.vsetvli(15, 0, Asm.vtype(32, 8, ta=True, ma=True))
        elif kind == "vredsum": a.vop(0x00, 2, 24, 8, 24)
        elif kind == "vfredusum": a.vop(0x01, 1, 24, 8, 24)
        elif kind == "vid": a.vop(0x14, 2, 24, 0, 0x11)
        elif kind == "vmv_s_x": a.vop(0x10, 6, 24, 0, 12)
        elif kind == "scalar": a.add(16 + (k & 7), 12, 13)
        else: raise SystemExit(f"unknown class {kind}")
    a.addi(9, 9, -1)
    a.bnez(9, "loop")
    WL._exit_tohost(a, S["tohost"])
    return WL._finish(a, "vbench-" + kind, "rv64imafdcv_zicsr", d, mem_size=1 << 16, max_inst=iters * 34 + 4096, meta=dict(vlenb=vlenb))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--harts", type=int, default=16384)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--vlenb", type=int, default=64)
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--classes", default=CLASSES)
    args = ap.parse_args()
    import whisper_amd as W
    for kind in args.classes.split(","):
        p = make(kind, args.iters, args.vlenb)
        b = W.Batch(args.harts, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size, lanes_per_wave=args.lanes, vlen_bytes=args.vlenb)
        try:
            b.load_elf_bytes(p.elf)
            b.set_stop(max_inst=p.max_inst)
            left = b.run(trace_hash=args.trace)
            ks, launches = b.run_stats()
            res = b.results()
            insts = sum(r.exec_count for r in res)
            ok = left == 0 and all(r.status == W.ST_STOP_TOHOST for r in res)
            per_hart = insts / args.harts
            print(json.dumps(dict(kind=kind, harts=args.harts, vlenb=args.vlenb, ok=ok, insts_per_hart=per_hart, kernel_ms=ks * 1e3,
                                  gips=insts / ks / 1e9, us_per_inst_per_hart=ks / per_hart * 1e6)), flush=True)
        finally:
            b.close()


if __name__ == "__main__":
    main()
