"""Pin the CPU oracle against the reference outputs held in tests/golden/ (SURVEY.md Appendix D,
captured from the real reference) - CPU only."""
import json
import os

import pytest

from simutil import rvoracle

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_d.json")))
KIND = {"r": rvoracle.CH_R, "f": rvoracle.CH_F, "m": rvoracle.CH_M, "c": rvoracle.CH_C, "v": rvoracle.CH_V}


def _group(records):
    out = {}
    for r in records:
        out.setdefault(r["tag"], []).append(r)
    return out


def _run_fp_trap():
    g = GOLD["fp_trap"]
    # reference run: 4 GiB RAM from 0, program poked at 0x80000000, mtvec=0 (trap vector = zeroed RAM)
    h = rvoracle.Hart(isa="rv64imafdc_zicsr", mem_base=0, mem_size=0x80000000 + (1 << 16), reset_pc=g["base"])
    img = b"".join(w.to_bytes(4, "little") for w in g["words"])
    h.write_mem(g["base"], img)
    st, log = h.run(1 << 20, log_cap=64)
    return g, h, st, log


def test_fp_trap_trace_records():
    g, h, st, log = _run_fp_trap()
    gold = _group(g["records"])
    by_tag = {c.tag: c for c in log}
    assert len(log) == g["final"]["executed"]
    for tag, recs in gold.items():
        c = by_tag[tag]
        assert c.pc == recs[0]["pc"], tag
        op = f"{c.inst:08x}" if c.size == 4 else f"{c.inst & 0xffff:04x}"
        assert op == recs[0]["op"], tag
        assert rvoracle.PRIV_STR[c.priv] == recs[0]["priv"]
        want = [(KIND[r["kind"]], r["key"], r["val"]) for r in recs]
        if recs[-1]["cont"]:          # capture elided the rest of this record ("... identical ...")
            assert c.changes()[:len(want)] == want, f"tag {tag}"
        else:
            assert c.changes() == want, f"tag {tag}"
        assert (c.trap == 1) == ("(exception)" in recs[0]["text"]), tag
    # the capture states #8..#23 are identical illegal-instruction traps at pc 0: mepc=0, mcause=2, mtval=0
    assert by_tag[8].changes()[1:] == [(KIND["c"], 0x341, 0), (KIND["c"], 0x342, 2), (KIND["c"], 0x343, 0)]
    for tag in range(9, 24):
        assert by_tag[tag].changes() == by_tag[8].changes() and by_tag[tag].pc == 0 and by_tag[tag].trap == 1


def test_fp_trap_final_state():
    g, h, st, log = _run_fp_trap()
    f = g["final"]
    assert st == rvoracle.STOP_ILLEGAL and h.stop_value == 3            # "16 consecutive illegal instructions"
    assert (h.exec_count, h.retired) == (f["executed"], f["retired"])
    assert h.pc == f["pc"] and h.f(2) == f["f2"]
    assert h.csr(0x300) == f["mstatus"] and h.csr(0x342) == f["mcause"] and h.csr(0x341) == f["mepc"]
    assert h.csr(0xB00) == f["mcycle"] and h.csr(0xB02) == f["minstret"]


@pytest.mark.skipif(rvoracle.ref() is None, reason="oracle/_ref (reference Disassembler) not built")
def test_fp_trap_log_text():
    """Full text lines, disassembly by the reference's own Disassembler compiled into oracle/_ref."""
    g, h, st, log = _run_fp_trap()
    gold = _group(g["records"])
    for c in log:
        if c.tag not in gold:
            continue
        recs = gold[c.tag]
        is_fetch_fault_or_zero = c.size == 2 and (c.inst & 0xffff) == 0
        dis = "illegal" if is_fetch_fault_or_zero else rvoracle.ref_disassemble(c.inst)
        got = [ln.rstrip() for ln in rvoracle.format_commit(c, 0, dis).split("\n")]
        want = []
        for r in recs:
            op = r["op"]
            want.append((f"#{r['tag']} {r['hart']}  {r['priv']} {r['pc']:016x} {op:>8s} {r['kind']} {r['key']:016x} {r['val']:016x} "
                         f"{r['text']}").rstrip() + ("  +" if r["cont"] else ""))
        if recs[-1]["cont"]:
            got = got[:len(want)]
        assert got == want, c.tag


def test_sv39_sample_first_record_mret():
    g = GOLD["sv39_first"]
    r = g["records"][0]
    h = rvoracle.Hart(isa="rv64imafdcsu_zicsr", mem_base=0x80000000, mem_size=1 << 16)
    h.write_mem(0x80000000, int(r["op"], 16).to_bytes(4, "little"))
    ms = h.csr(0x300)
    h.set_csr(0x300, (ms & ~(3 << 11)) | (g["pokes"]["mstatus_mpp"] << 11))
    h.set_csr(0x341, g["pokes"]["mepc"])
    st, log = h.run(1, log_cap=4)
    c = log[0]
    assert (c.tag, c.pc, c.inst) == (1, r["pc"], int(r["op"], 16))
    assert c.changes() == [(KIND[r["kind"]], r["key"], r["val"])]
    assert h.pc == g["next_pc"] and rvoracle.PRIV_STR[h.priv] == g["next_priv"]


def test_sv39_sample_whole_trace():
    """The survey's Sv39 capture: mret to S at a virtual address mapped by a 1 GiB leaf, three S-mode instructions fetched
    and one store made through the translation, stop on the *physical* tohost address."""
    g = GOLD["sv39"]
    h = rvoracle.Hart(isa=g["isa"], mem_base=0x80000000, mem_size=1 << 16, tohost=g["tohost"])
    for r in g["records"]:
        pa = r["pc"] if r["pc"] >= 0x80000000 else r["pc"] - g["va_to_pa"] + 0x80000000
        h.write_mem(pa, int(r["op"], 16).to_bytes(4, "little"))
    for a, v in g["mem"].items():
        h.write_mem(int(a, 16), v.to_bytes(8, "little"))
    ms = h.csr(0x300)
    h.set_csr(0x300, (ms & ~(3 << 11)) | (g["pokes"]["mstatus_mpp"] << 11))
    h.set_csr(0x341, g["pokes"]["mepc"])
    h.set_csr(0x180, g["pokes"]["satp"])
    h.set_csr(0x3B0, g["pokes"]["pmpaddr0"])
    h.set_csr(0x3A0, g["pokes"]["pmpcfg0"])
    st, log = h.run(100, log_cap=16)
    assert st == rvoracle.STOP_TOHOST and h.stop_value == 1
    gold = _group(g["records"])
    assert len(log) == len(gold)
    for c in log:
        recs = gold[c.tag]
        assert (c.pc, f"{c.inst:08x}", rvoracle.PRIV_STR[c.priv]) == (recs[0]["pc"], recs[0]["op"], recs[0]["priv"]), c.tag
        assert c.changes() == [(KIND[r["kind"]], r["key"], r["val"]) for r in recs], c.tag


def test_sv39_sample_needs_pmp():
    """Same capture, its last paragraph: without the PMP pokes the first S-mode fetch takes an instruction access fault
    (mcause = 1), because the PTE read fails the S-mode PMP check."""
    g = GOLD["sv39"]
    h = rvoracle.Hart(isa=g["isa"], mem_base=0x80000000, mem_size=1 << 16, tohost=g["tohost"])
    h.write_mem(0x80000000, (0x30200073).to_bytes(4, "little"))
    for a, v in g["mem"].items():
        h.write_mem(int(a, 16), v.to_bytes(8, "little"))
    ms = h.csr(0x300)
    h.set_csr(0x300, (ms & ~(3 << 11)) | (1 << 11))
    h.set_csr(0x341, g["pokes"]["mepc"])
    h.set_csr(0x180, g["pokes"]["satp"])
    st, log = h.run(2, log_cap=4)
    assert log[1].trap == 1 and h.csr(0x342) == 1 and h.csr(0x343) == g["pokes"]["mepc"]


def test_rvv_sample_trace():
    """The survey's RVV capture (default VLEN 512): vsetvli x6, x0, e32,m1,ta,ma / vid.v v1 / vadd.vv v2, v1, v1 after
    mstatus.FS and VS were switched on: the vsetvli record (x6, mstatus VS -> Dirty, vl, vtype) and both register images."""
    g = GOLD["rvv"]
    h = rvoracle.Hart(isa=g["isa"], mem_base=0x80000000, mem_size=1 << 16)
    words = [0x3002a073, int(g["records"][0]["op"], 16)] + [int(v["op"], 16) for v in g["vlines"]]   # csrrs x0, mstatus, x5 first
    h.write_mem(g["first_pc"], b"".join(w.to_bytes(4, "little") for w in words))
    h.pc = g["first_pc"]
    h.set_x(5, g["x5"])
    st, log = h.run(len(words), log_cap=8)
    gold = _group(g["records"])
    c = log[1]
    recs = gold[4]
    assert (c.pc, f"{c.inst:08x}") == (recs[0]["pc"], recs[0]["op"])
    assert c.changes() == [(KIND[r["kind"]], r["key"], r["val"]) for r in recs]
    for c, v in zip(log[2:], g["vlines"]):
        assert (c.pc, f"{c.inst:08x}") == (v["pc"], v["op"])
        ch = c.changes()
        assert len(ch) == 1 and ch[0][0] == KIND["v"] and ch[0][1] == v["reg"]
        img = h.vreg(v["reg"])[::-1].hex()                     # the log prints the most significant byte first
        assert len(img) == 128 and img.startswith(v["head"]) and img.endswith(v["tail"])
    assert h.vreg(1) == b"".join(i.to_bytes(4, "little") for i in range(16))
    assert h.vreg(2) == b"".join((2 * i).to_bytes(4, "little") for i in range(16))


def test_rvv_widening_known_answers():
    """vwadd[u] / vwsub[u] / vwmul[u|su] .vv and .vx at SEW 8/16/32 against numpy (values follow from the V spec's definition;
    the reference holds no golden vector for these: parity unpinned beyond the legality rules restated from vector.cpp)."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(5)
    ops = {0x30: ("add", 0, 0), 0x31: ("add", 1, 1), 0x32: ("sub", 0, 0), 0x33: ("sub", 1, 1), 0x38: ("mul", 0, 0), 0x3A: ("mul", 1, 0),
           0x3B: ("mul", 1, 1)}
    for sew, udt, sdt, wudt in [(8, np.uint8, np.int8, np.uint16), (16, np.uint16, np.int16, np.uint32), (32, np.uint32, np.int32, np.uint64)]:
        for f6, (kind, sa, sb) in ops.items():
            for vx in (0, 1):
                a = Asm(0x80000000)
                a.csrs(0x300, 5)
                a.vsetvli(6, 0, Asm.vtype(sew, 1))
                a.vop(f6, 6 if vx else 2, 4, 8, 11 if vx else 9)
                h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
                h.write_mem(0x80000000, a.assemble())
                h.pc = 0x80000000
                h.set_x(5, 0x6600)
                xs = int(rng.integers(0, 2**63)) | (1 << (sew - 1))
                h.set_x(11, xs)
                A, B = rng.integers(0, 256, 64, dtype=np.uint8), rng.integers(0, 256, 64, dtype=np.uint8)
                h.set_vreg(8, A.tobytes()); h.set_vreg(9, B.tobytes())
                h.run(3)
                n = 64 * 8 // sew
                av = A.view(udt)
                bv = np.full(n, xs & (2**sew - 1), dtype=np.uint64).astype(udt) if vx else B.view(udt)
                wa = (av.view(sdt).astype(np.int64) if sa else av.astype(np.int64)).astype(object)
                wb = (bv.view(sdt).astype(np.int64) if sb else bv.astype(np.int64)).astype(object)
                want = [(int(x + y if kind == "add" else x - y if kind == "sub" else x * y)) & (2**(2 * sew) - 1) for x, y in zip(wa, wb)]
                got = np.frombuffer(h.vreg(4) + h.vreg(5), dtype=wudt)
                assert [int(g) for g in got] == want, (sew, hex(f6), vx)
    # SEW=64 has no double-width type, and an overlapping destination in the low half of the group is illegal
    for sew, vd, vs2 in [(64, 4, 8), (32, 8, 8), (32, 5, 8)]:
        a = Asm(0x80000000)
        a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, 1)); a.vop(0x30, 2, vd, vs2, 12)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600)
        h.run(3)
        assert h.csr(0x342) == 2, (sew, vd, vs2)


def test_rvv_fp_known_answers():
    """The eight vector multiply-add forms, vfmin / vfmax, vfsgnj* and the FP compares at SEW 32 / 64 against numpy on exactly
    representable operands plus NaN / signed-zero cases (semantics from the V spec; parity unpinned beyond that)."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(9)
    for sew, ft, ut in [(32, np.float32, np.uint32), (64, np.float64, np.uint64)]:
        n = 512 // sew
        A = rng.integers(-40, 40, n).astype(ft); B = rng.integers(-40, 40, n).astype(ft); D = rng.integers(-40, 40, n).astype(ft)
        A[0], B[0] = np.nan, 1.0; A[1], B[1] = 0.0, -0.0; A[2], B[2] = -0.0, 0.0; A[3], B[3] = np.nan, np.nan; B[4] = A[4]

        def run(f6, f3=1, vd_init=None):
            a = Asm(0x80000000)
            a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, 1)); a.vop(f6, f3, 4, 8, 9)
            h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
            h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600)
            h.set_vreg(8, A.tobytes()); h.set_vreg(9, B.tobytes()); h.set_vreg(4, (D if vd_init is None else vd_init).tobytes())
            h.run(3)
            return h
        # vs2 = A, vs1 = B, vd = D
        fma = {0x28: B * D + A, 0x29: -(B * D) - A, 0x2A: B * D - A, 0x2B: -(B * D) + A, 0x2C: B * A + D, 0x2D: -(B * A) - D, 0x2E: B * A - D,
               0x2F: -(B * A) + D}
        for f6, want in fma.items():
            got = np.frombuffer(run(f6).vreg(4), dtype=ft)
            ok = (got == want) | (np.isnan(got) & np.isnan(want))
            assert ok.all(), (sew, hex(f6), got, want)
        with np.errstate(invalid="ignore"):
            for f6, want in {0x04: np.fmin(A, B), 0x06: np.fmax(A, B)}.items():
                h = run(f6)
                got = np.frombuffer(h.vreg(4), dtype=ft)
                assert ((got == want) | (np.isnan(got) & np.isnan(want))).all(), (sew, hex(f6))
                assert np.signbit(got[1]) == (f6 == 0x04) and np.signbit(got[2]) == (f6 == 0x04)        # min(+0,-0) = -0, max = +0
                assert got.view(ut)[3] == (0x7fc00000 if sew == 32 else 0x7ff8000000000000)
                assert h.csr(0x003) & 0x1f == 0                                                       # quiet NaNs raise nothing
            for f6, want in {0x08: np.copysign(A, B), 0x09: np.copysign(A, -B), 0x0A: np.copysign(A, np.where(np.signbit(A) ^ np.signbit(B), -1.0, 1.0))}.items():
                got = np.frombuffer(run(f6).vreg(4), dtype=ut)
                assert (got == want.astype(ft).view(ut)).all(), (sew, hex(f6))
            cmp_ = {0x18: A == B, 0x19: A <= B, 0x1B: A < B, 0x1C: A != B}
            for f6, want in cmp_.items():
                h = run(f6)
                bits = int.from_bytes(h.vreg(4), "little")
                assert [(bits >> i) & 1 for i in range(n)] == [int(w) for w in want], (sew, hex(f6))
                assert (h.csr(0x003) & 0x10 != 0) == (f6 in (0x19, 0x1B))                              # le / lt signal on quiet NaNs


def test_rvv_whole_register_and_mask_ldst():
    """vl2re32.v / vs2r.v move 2 * VLENB bytes whatever vl / vtype say (even vill); vlm.v / vsm.v move ceil(vl / 8) bytes and fill
    the rest of the register with ones (the reference forces tail-agnostic); odd register numbers and nf = 3 are illegal."""
    from whisper_amd.rvasm import Asm
    import random
    rng = random.Random(3)
    blob = bytes(rng.randrange(256) for _ in range(256))

    def hart(build):
        a = Asm(0x80000000)
        a.csrs(0x300, 5)
        build(a)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.write_mem(0x80001000, blob); h.pc = 0x80000000
        h.set_x(5, 0x6600); h.set_x(10, 0x80001000); h.set_x(11, 0x80002000)
        return h
    h = hart(lambda a: (a.vsetivli(6, 3, Asm.vtype(16, 1)), a.vls(False, 32, 8, 10, lumop=8, nf=1), a.vls(True, 8, 8, 11, lumop=8, nf=1)))
    h.run(4)
    assert h.vreg(8) + h.vreg(9) == blob[:128] and h.read_mem(0x80002000, 128) == blob[:128] and h.read_mem(0x80002080, 8) == bytes(8)
    h = hart(lambda a: (a.li(7, 1 << 63), a.vsetvl(6, 0, 7), a.vls(False, 8, 4, 10, lumop=8, nf=3)))      # vill: still legal
    h.run(5)
    assert h.csr(0xC21) >> 63 == 1 and b"".join(h.vreg(i) for i in range(4, 8)) == blob and h.csr(0x342) == 0
    for v, nf in [(9, 2), (8, 3), (4, 8)]:
        h = hart(lambda a: a.vls(False, 8, v, 10, lumop=8, nf=nf - 1))
        h.run(2)
        assert h.csr(0x342) == 2, (v, nf)
    h = hart(lambda a: (a.vsetivli(6, 19, Asm.vtype(32, 2)), a.vls(False, 8, 3, 10, lumop=0xB), a.vls(True, 8, 3, 11, lumop=0xB)))
    h.run(4)
    assert h.vreg(3) == blob[:3] + b"\xff" * 61 and h.read_mem(0x80002000, 4) == blob[:3] + b"\0"


def test_rvv_mask_scan_move_extend_known_answers():
    """vmv2r.v, vcpop.m / vfirst.m, vmsbf / vmsif / vmsof, viota.m, vzext / vsext against values worked out from the V spec."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, 128, dtype=np.uint8)
    mask_bits = [0, 0, 0, 1, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0]                   # first set bit at 3, five... four set bits
    mask = np.zeros(64, dtype=np.uint8); mask[0] = 0x68; mask[1] = 0x04

    def run(build, steps):
        a = Asm(0x80000000)
        a.csrs(0x300, 5)
        build(a)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600)
        h.set_vreg(8, img[:64].tobytes()); h.set_vreg(9, img[64:].tobytes()); h.set_vreg(2, mask.tobytes())
        h.run(steps)
        return h
    h = run(lambda a: (a.vsetivli(6, 1, Asm.vtype(8, 1)), a.vop(0x27, 3, 12, 8, 1)), 3)             # vmv2r.v v12, v8: whole registers whatever vl is
    assert h.vreg(12) + h.vreg(13) == img.tobytes()
    h = run(lambda a: (a.vsetivli(6, 16, Asm.vtype(8, 1)), a.vop(0x10, 2, 10, 2, 0x10), a.vop(0x10, 2, 11, 2, 0x11), a.vop(0x10, 2, 12, 3, 0x11)), 5)
    assert (h.x(10), h.x(11), h.x(12)) == (4, 3, 2**64 - 1)
    for k, want in [(1, [1, 1, 1] + [0] * 13), (3, [1, 1, 1, 1] + [0] * 12), (2, [0, 0, 0, 1] + [0] * 12)]:
        h = run(lambda a: (a.vsetivli(6, 16, Asm.vtype(8, 1)), a.vop(0x14, 2, 4, 2, k)), 3)
        bits = int.from_bytes(h.vreg(4), "little")
        assert [(bits >> i) & 1 for i in range(16)] == want and bits >> 16 == 0                      # past vl: kept (was zero)
    h = run(lambda a: (a.vsetivli(6, 16, Asm.vtype(16, 1)), a.vop(0x14, 2, 4, 2, 0x10)), 3)           # viota.m: exclusive prefix count
    got = np.frombuffer(h.vreg(4), dtype=np.uint16)[:16]
    assert list(got) == list(np.concatenate([[0], np.cumsum(mask_bits)[:-1]]))
    for sew, f, sign in [(16, 2, 0), (32, 4, 1), (64, 8, 1), (64, 2, 0), (32, 2, 1)]:
        code = {(2, 0): 6, (2, 1): 7, (4, 0): 4, (4, 1): 5, (8, 0): 2, (8, 1): 3}[(f, sign)]
        h = run(lambda a: (a.vsetvli(6, 0, Asm.vtype(sew, 1)), a.vop(0x12, 2, 4, 8, code)), 3)
        n = 512 // sew
        src = img[:n * sew // 8 // f].view({1: np.uint8, 2: np.uint16, 4: np.uint32}[sew // 8 // f])
        if sign:
            src = src.view({1: np.int8, 2: np.int16, 4: np.int32}[sew // 8 // f])
        want = src.astype(np.int64).astype({16: np.uint16, 32: np.uint32, 64: np.uint64}[sew])
        assert h.vreg(4) == want.tobytes(), (sew, f, sign)
    h = run(lambda a: (a.vsetvli(6, 0, Asm.vtype(8, 1)), a.vop(0x12, 2, 4, 8, 6)), 3)                 # SEW 8 has no narrower source
    assert h.csr(0x342) == 2


def test_rvv_div_mulh_slide_gather_known_answers():
    """vdiv[u] / vrem[u] (x / 0, min / -1), vmulh[u|su], vslide*, vrgather against numpy / the V spec's definitions."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(13)

    def run(sew, build, A, B, steps=3, xs=0, lmul=1, D=None):
        a = Asm(0x80000000)
        a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, lmul))
        build(a)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600); h.set_x(11, xs)
        h.set_vreg(8, A.tobytes()); h.set_vreg(9, B.tobytes())
        if D is not None:
            h.set_vreg(4, D.tobytes())
        h.run(steps)
        return h
    for sew, st, ut in [(8, np.int8, np.uint8), (16, np.int16, np.uint16), (32, np.int32, np.uint32), (64, np.int64, np.uint64)]:
        n = 512 // sew
        A = rng.integers(0, 256, 64, dtype=np.uint8).view(ut).copy(); B = rng.integers(0, 256, 64, dtype=np.uint8).view(ut).copy()
        B[0] = 0; A[1] = 1 << (sew - 1); B[1] = (1 << sew) - 1; B[2] = 1; A[3] = 0
        sa, sb = A.view(st).astype(object), B.view(st).astype(object); ua, ub = A.astype(object), B.astype(object)
        m = (1 << sew) - 1

        def tdiv(x, y):
            q = abs(x) // abs(y)
            return q if (x < 0) == (y < 0) else -q
        want = {
            0x20: [(x // y if y else m) & m for x, y in zip(ua, ub)],
            0x21: [(tdiv(x, y) if y else -1) & m for x, y in zip(sa, sb)],
            0x22: [(x % y if y else x) & m for x, y in zip(ua, ub)],
            0x23: [((x - tdiv(x, y) * y) if y else x) & m for x, y in zip(sa, sb)],
            0x24: [((x * y) >> sew) & m for x, y in zip(ua, ub)],
            0x26: [((x * y) >> sew) & m for x, y in zip(sa, ub)],
            0x27: [((x * y) >> sew) & m for x, y in zip(sa, sb)],
        }
        for f6, w in want.items():
            h = run(sew, lambda a: a.vop(f6, 2, 4, 8, 9), A, B)
            got = [int(v) for v in np.frombuffer(h.vreg(4), dtype=ut)]
            assert got == [int(v) for v in w], (sew, hex(f6))
        # slides: vslideup.vi 3 keeps vd[0..2]; vslidedown.vi 3 reads zero past VLMAX; vslide1up / vslide1down insert x11
        D = rng.integers(0, 256, 64, dtype=np.uint8).view(ut).copy()
        xs = int(rng.integers(0, 2**63))
        h = run(sew, lambda a: a.vop(0x0E, 3, 4, 8, 3), A, B, D=D)
        assert list(np.frombuffer(h.vreg(4), dtype=ut)) == list(D[:3]) + list(A[:n - 3])
        h = run(sew, lambda a: a.vop(0x0F, 3, 4, 8, 3), A, B, D=D)
        assert list(np.frombuffer(h.vreg(4), dtype=ut)) == list(A[3:]) + [0, 0, 0]
        h = run(sew, lambda a: a.vop(0x0E, 6, 4, 8, 11), A, B, D=D, xs=xs)
        assert [int(v) for v in np.frombuffer(h.vreg(4), dtype=ut)] == [xs & m] + [int(v) for v in A[:n - 1]]
        h = run(sew, lambda a: a.vop(0x0F, 6, 4, 8, 11), A, B, D=D, xs=xs)
        assert [int(v) for v in np.frombuffer(h.vreg(4), dtype=ut)] == [int(v) for v in A[1:]] + [xs & m]
        # vrgather.vv: indices beyond VLMAX give zero
        I = rng.integers(0, 2 * n, n).astype(ut)
        h = run(sew, lambda a: a.vop(0x0C, 0, 4, 8, 9), A, I)
        assert [int(v) for v in np.frombuffer(h.vreg(4), dtype=ut)] == [int(A[i]) if i < n else 0 for i in I]
        h = run(sew, lambda a: a.vop(0x0C, 4, 4, 8, 11), A, I, xs=5)
        assert [int(v) for v in np.frombuffer(h.vreg(4), dtype=ut)] == [int(A[5])] * n
        h = run(sew, lambda a: a.vop(0x0C, 0, 8, 8, 9), A, I)                                        # vd overlaps the table
        assert h.csr(0x342) == 2


def test_rvv_narrowing_and_wide_operand_known_answers():
    """vnsrl / vnsra .wv / .wi and vwadd[u].wv / vwsub[u].wx against numpy; a narrow destination may overlap only the lowest
    register of its wide source."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, 192, dtype=np.uint8)

    def run(sew, build, xs=0):
        a = Asm(0x80000000)
        a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, 1))
        build(a)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600); h.set_x(11, xs)
        h.set_vreg(8, img[:64].tobytes()); h.set_vreg(9, img[64:128].tobytes()); h.set_vreg(12, img[128:].tobytes())
        h.run(3)
        return h
    for sew, ut, st, wut, wst in [(8, np.uint8, np.int8, np.uint16, np.int16), (16, np.uint16, np.int16, np.uint32, np.int32),
                                  (32, np.uint32, np.int32, np.uint64, np.int64)]:
        wide_u = img[:128].view(wut); wide_s = img[:128].view(wst); sh = img[128:].view(ut)
        m = 2 * sew - 1
        h = run(sew, lambda a: a.vop(0x2C, 0, 4, 8, 12))
        assert (np.frombuffer(h.vreg(4), dtype=ut) == (wide_u >> (sh & m).astype(wut)).astype(ut)).all(), sew
        h = run(sew, lambda a: a.vop(0x2D, 0, 4, 8, 12))
        assert (np.frombuffer(h.vreg(4), dtype=ut) == (wide_s >> (sh & m).astype(wst)).astype(ut)).all(), sew
        h = run(sew, lambda a: a.vop(0x2C, 3, 4, 8, sew // 2))
        assert (np.frombuffer(h.vreg(4), dtype=ut) == (wide_u >> wut(sew // 2)).astype(ut)).all(), sew
        h = run(sew, lambda a: a.vop(0x2C, 3, 8, 8, 1))                                            # vd = low half of the source: legal
        assert h.csr(0x342) == 0 and (np.frombuffer(h.vreg(8), dtype=ut) == (wide_u >> wut(1)).astype(ut)).all()
        h = run(sew, lambda a: a.vop(0x2C, 3, 9, 8, 1))                                            # vd = high half: illegal
        assert h.csr(0x342) == 2
        h = run(sew, lambda a: a.vop(0x34, 2, 4, 8, 12))                                           # vwaddu.wv
        assert (np.frombuffer(h.vreg(4) + h.vreg(5), dtype=wut) == wide_u + sh.astype(wut)).all(), sew
        h = run(sew, lambda a: a.vop(0x35, 2, 4, 8, 12))                                           # vwadd.wv
        assert (np.frombuffer(h.vreg(4) + h.vreg(5), dtype=wst) == wide_s + sh.view(st).astype(wst)).all(), sew
        xs = int(rng.integers(0, 2**63))
        h = run(sew, lambda a: a.vop(0x37, 6, 4, 8, 11), xs=xs)                                    # vwsub.wx
        sx = np.array([xs & (2**sew - 1)], dtype=np.uint64).astype(ut).view(st).astype(wst)[0]
        assert (np.frombuffer(h.vreg(4) + h.vreg(5), dtype=wst) == wide_s - sx).all(), sew


def test_rvv_fp_unary_known_answers():
    """vfsqrt.v, vfcvt.{x,xu}.f.v (frm = RNE), vfcvt.rtz.x.f.v, vfcvt.f.{x,xu}.v and vfmerge.vfm against numpy."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(19)
    for sew, ft, it, ut in [(32, np.float32, np.int32, np.uint32), (64, np.float64, np.int64, np.uint64)]:
        n = 512 // sew
        A = (rng.integers(-2000, 2000, n) / 8).astype(ft); A[0] = 2.5; A[1] = 3.5; A[2] = -2.5; A[3] = 1e30
        I = rng.integers(-2**31, 2**31, n).astype(it)
        mask = rng.integers(0, 256, 64, dtype=np.uint8)

        def run(f6, f3, s1, vs2, vm=1):
            a = Asm(0x80000000)
            a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, 1)); a.vop(f6, f3, 4, 8, s1, vm)
            h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
            h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600)
            h.set_vreg(8, vs2.tobytes()); h.set_vreg(0, mask.tobytes()); h.set_f(3, int(np.array([7.25], dtype=ft).view(ut)[0]) | (0 if sew == 64 else 0xFFFFFFFF00000000))
            h.run(3)
            return h
        P = np.abs(A)
        assert (np.frombuffer(run(0x13, 1, 0, P).vreg(4), dtype=ft) == np.sqrt(P)).all()
        lim_s, lim_u = 2**(sew - 1), 2**sew
        rne = [int(np.rint(float(v))) for v in A]
        got = np.frombuffer(run(0x12, 1, 1, A).vreg(4), dtype=it)
        assert [int(g) for g in got] == [min(max(v, -lim_s), lim_s - 1) for v in rne]
        got = np.frombuffer(run(0x12, 1, 0, A).vreg(4), dtype=ut)
        assert [int(g) for g in got] == [min(max(v, 0), lim_u - 1) for v in rne]
        got = np.frombuffer(run(0x12, 1, 7, A).vreg(4), dtype=it)
        assert [int(g) for g in got] == [min(max(int(float(v)), -lim_s), lim_s - 1) for v in A]
        assert (np.frombuffer(run(0x12, 1, 3, I).vreg(4), dtype=ft) == I.astype(ft)).all()
        assert (np.frombuffer(run(0x12, 1, 2, I).vreg(4), dtype=ft) == I.view(ut).astype(ft)).all()
        got = np.frombuffer(run(0x17, 5, 3, A, vm=0).vreg(4), dtype=ft)
        bits = int.from_bytes(mask.tobytes(), "little")
        assert [float(g) for g in got] == [7.25 if (bits >> i) & 1 else float(A[i]) for i in range(n)]


def test_rvv_indexed_ldst_known_answers():
    """vluxei16.v gathers SEW = 32 elements at zero-extended 16-bit byte offsets; vsoxei16.v scatters them back; an index register
    overlapping a wider destination group anywhere but its top is illegal."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(23)
    table = rng.integers(0, 2**32, 256, dtype=np.uint32)
    idx = (rng.permutation(256)[:16] * 4).astype(np.uint16)                                          # 16 elements at SEW 32, VLEN 512

    def hart(build):
        a = Asm(0x80000000)
        a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(32, 1))
        build(a)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.write_mem(0x80001000, table.tobytes()); h.pc = 0x80000000
        h.set_x(5, 0x6600); h.set_x(10, 0x80001000); h.set_x(11, 0x80003000)
        h.set_vreg(8, idx.tobytes() + bytes(32))
        return h
    h = hart(lambda a: (a.vls(False, 16, 4, 10, lumop=8, mop=1), a.vls(True, 16, 4, 11, lumop=8, mop=3)))
    h.run(4)
    got = np.frombuffer(h.vreg(4), dtype=np.uint32)
    assert (got == table[idx // 4]).all()
    out = np.frombuffer(h.read_mem(0x80003000, 1024), dtype=np.uint32)
    want = np.zeros(256, dtype=np.uint32); want[idx // 4] = table[idx // 4]
    assert (out == want).all()
    h = hart(lambda a: (a.vsetvli(6, 0, Asm.vtype(32, 2)), a.vls(False, 16, 8, 10, lumop=8, mop=1)))  # vd group v8-v9, index v8 (EMUL 1): low overlap
    h.run(4)
    assert h.csr(0x342) == 2
    h = hart(lambda a: (a.vsetvli(6, 0, Asm.vtype(32, 2)), a.vls(False, 16, 8, 10, lumop=9, mop=1)))  # index in the top register of the group: legal
    h.run(4)
    assert h.csr(0x342) in (0, 5)                                                                    # legal encoding (v9 holds zeros -> loads table[0])


def test_rvv_segment_and_fault_first_known_answers():
    """vlseg3e16.v de-interleaves, vsseg3e16.v re-interleaves, vlsseg2e32.v strides; vle32ff.v running off the end of memory trims
    vl to the elements loaded (and traps when element 0 faults)."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(29)
    blob = rng.integers(0, 2**16, 2048, dtype=np.uint16)
    MEM = 1 << 16

    def hart(build, sew=16, lmul=1, ta=False):
        a = Asm(0x80000000)
        a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, lmul, ta=ta))
        build(a)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=MEM)
        h.write_mem(0x80000000, a.assemble()); h.write_mem(0x80001000, blob.tobytes()); h.pc = 0x80000000
        h.set_x(5, 0x6600); h.set_x(10, 0x80001000); h.set_x(11, 0x80003000); h.set_x(12, 24)
        return h
    h = hart(lambda a: (a.vls(False, 16, 4, 10, nf=2), a.vls(True, 16, 4, 11, nf=2)))
    h.run(4)
    seg = blob[:96].reshape(32, 3)
    for f in range(3):
        assert (np.frombuffer(h.vreg(4 + f), dtype=np.uint16) == seg[:, f]).all()
    assert h.read_mem(0x80003000, 192) == blob[:96].tobytes()
    h = hart(lambda a: a.vls(False, 32, 4, 10, lumop=12, nf=1, mop=2), sew=32)                      # vlsseg2e32.v, stride x12 = 24 bytes
    h.run(3)
    w = blob.view(np.uint32)
    assert (np.frombuffer(h.vreg(4), dtype=np.uint32) == w[0:16 * 6:6]).all() and (np.frombuffer(h.vreg(5), dtype=np.uint32) == w[1:16 * 6:6]).all()
    for v, nf in [(30, 4), (4, 8)]:                                                                  # past v31; EMUL * NFIELDS > 8 at LMUL 2
        h = hart(lambda a: a.vls(False, 16, v, 10, nf=nf - 1), lmul=2 if nf == 8 else 1)
        h.run(3)
        assert h.csr(0x342) == 2
    end = 0x80000000 + MEM
    h = hart(lambda a: a.vls(False, 32, 4, 10, lumop=0x10), sew=32, ta=True)
    h.set_x(10, end - 20); h.write_mem(end - 20, bytes(range(20)))
    h.run(3)
    assert h.csr(0xC20) == 5 and h.csr(0x342) == 0 and h.vreg(4) == bytes(range(20)) + b"\xff" * 44
    h = hart(lambda a: a.vls(False, 32, 4, 10, lumop=0x10), sew=32)
    h.set_x(10, end)
    h.run(3)
    assert h.csr(0x342) == 5 and h.csr(0xC20) == 16


def test_rvv_wmacc_wredsum_compress_ei16_known_answers():
    """vwmacc[u|su|us], vwredsum[u].vs, vcompress.vm, vrgatherei16.vv, vfredmin / vfredmax, vfslide1up / vfslide1down vs numpy."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(31)
    A = rng.integers(0, 256, 64, dtype=np.uint8); B = rng.integers(0, 256, 64, dtype=np.uint8); D = rng.integers(0, 256, 128, dtype=np.uint8)
    mask = rng.integers(0, 256, 64, dtype=np.uint8)

    def run(sew, build, xs=0, fval=0, v12=None):
        a = Asm(0x80000000)
        a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, 1))
        build(a)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600); h.set_x(11, xs); h.set_f(3, fval)
        h.set_vreg(8, A.tobytes()); h.set_vreg(9, B.tobytes()); h.set_vreg(4, D[:64].tobytes()); h.set_vreg(5, D[64:].tobytes())
        h.set_vreg(2, mask.tobytes())
        if v12 is not None:
            h.set_vreg(12, v12.tobytes())
        h.run(3)
        return h
    for sew, ut, st, wut in [(8, np.uint8, np.int8, np.uint16), (16, np.uint16, np.int16, np.uint32), (32, np.uint32, np.int32, np.uint64)]:
        a2, a1, acc = A.view(ut), B.view(ut), D.view(wut).astype(object)
        u2, u1 = a2.astype(object), a1.astype(object); s2, s1 = a2.view(st).astype(object), a1.view(st).astype(object)
        m = (1 << (2 * sew)) - 1
        for f6, x2, x1 in [(0x3C, u2, u1), (0x3D, s2, s1), (0x3F, u2, s1)]:                       # vwmaccu, vwmacc, vwmaccsu (signed vs1 x unsigned vs2)
            h = run(sew, lambda a: a.vop(f6, 2, 4, 8, 9))
            got = np.frombuffer(h.vreg(4) + h.vreg(5), dtype=wut)
            assert [int(g) for g in got] == [int(c + p * q) & m for c, p, q in zip(acc, x2, x1)], (sew, hex(f6))
        xs = int(rng.integers(0, 2**63)) | (1 << (sew - 1))
        h = run(sew, lambda a: a.vop(0x3E, 6, 4, 8, 11), xs=xs)                                    # vwmaccus.vx: unsigned x11 x signed vs2
        got = np.frombuffer(h.vreg(4) + h.vreg(5), dtype=wut)
        assert [int(g) for g in got] == [int(c + p * (xs & ((1 << sew) - 1))) & m for c, p in zip(acc, s2)], sew
        h = run(sew, lambda a: a.vop(0x3C, 2, 4, 4, 9))                                            # vd is read: a narrow source may not sit inside it
        assert h.csr(0x342) == 2
        for f6, x2 in [(0x30, u2), (0x31, s2)]:                                                    # vwredsumu.vs / vwredsum.vs: vd[0] = vs1[0] + sum(vs2)
            h = run(sew, lambda a: a.vop(f6, 0, 12, 8, 9))
            want = (int(B.view(wut)[0]) + sum(int(v) for v in x2)) & m
            assert int(np.frombuffer(h.vreg(12), dtype=wut)[0]) == want, (sew, hex(f6))
    n = 16
    a32 = A.view(np.uint32)
    h = run(32, lambda a: a.vop(0x17, 2, 12, 8, 2), v12=np.zeros(16, dtype=np.uint32))              # vcompress.vm v12, v8, v2
    bits = int.from_bytes(mask.tobytes(), "little")
    packed = [int(a32[i]) for i in range(n) if (bits >> i) & 1]
    assert [int(v) for v in np.frombuffer(h.vreg(12), dtype=np.uint32)] == packed + [0] * (n - len(packed))
    idx = rng.integers(0, 24, 32).astype(np.uint16)                                                # EEW 16 indices for SEW 32: EMUL 1/2, 16 used
    h = run(32, lambda a: a.vop(0x0E, 0, 4, 8, 12), v12=idx)
    assert [int(v) for v in np.frombuffer(h.vreg(4), dtype=np.uint32)] == [int(a32[i]) if i < 16 else 0 for i in idx[:16]]
    F = (rng.integers(-500, 500, 16) / 4).astype(np.float32); F[3] = -0.0
    A[:] = F.view(np.uint8); B[:] = np.full(16, 1000.0, dtype=np.float32).view(np.uint8)
    h = run(32, lambda a: a.vop(0x05, 1, 12, 8, 9))
    assert np.frombuffer(h.vreg(12), dtype=np.float32)[0] == F.min()
    B[:] = np.full(16, -1000.0, dtype=np.float32).view(np.uint8)
    h = run(32, lambda a: a.vop(0x07, 1, 12, 8, 9))
    assert np.frombuffer(h.vreg(12), dtype=np.float32)[0] == F.max()
    fbits = int(np.array([3.75], dtype=np.float32).view(np.uint32)[0]) | 0xFFFFFFFF00000000
    h = run(32, lambda a: a.vop(0x0E, 5, 12, 8, 3), fval=fbits)
    assert [float(v) for v in np.frombuffer(h.vreg(12), dtype=np.float32)] == [3.75] + [float(v) for v in F[:15]]
    h = run(32, lambda a: a.vop(0x0F, 5, 12, 8, 3), fval=fbits)
    assert [float(v) for v in np.frombuffer(h.vreg(12), dtype=np.float32)] == [float(v) for v in F[1:]] + [3.75]


def test_rvv_fixed_point_known_answers():
    """vsadd[u] / vssub[u] / vaadd[u] / vasub[u] / vsmul / vssrl / vssra / vnclip[u] against exact Python integers, under each of the
    four vxrm rounding modes; vxsat is set exactly when some element saturated."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(37)

    def rnd(v, d, rm):                                   # the V spec's roundoff on an exact integer
        if d == 0:
            return v
        low = v & ((1 << d) - 1); half = 1 << (d - 1); q = v >> d
        inc = {0: low >= half, 1: low > half or (low == half and q & 1), 2: False, 3: low != 0 and not q & 1}[rm]
        return q + int(inc)
    for sew, ut, st in [(8, np.uint8, np.int8), (32, np.uint32, np.int32), (64, np.uint64, np.int64)]:
        n = 512 // sew
        A = rng.integers(0, 256, 64, dtype=np.uint8).view(ut).copy(); B = rng.integers(0, 256, 64, dtype=np.uint8).view(ut).copy()
        A[0] = B[0] = 1 << (sew - 1); A[1] = (1 << sew) - 1; B[1] = 1; A[2] = (1 << (sew - 1)) - 1; B[2] = 1
        W = rng.integers(0, 256, 128, dtype=np.uint8)
        ua, ub = [int(v) for v in A], [int(v) for v in B]
        sa, sb = [int(v) for v in A.view(st)], [int(v) for v in B.view(st)]
        m, smax, smin = (1 << sew) - 1, (1 << (sew - 1)) - 1, -(1 << (sew - 1))
        clamp_s = lambda v: max(smin, min(smax, v))
        for rm in range(4):
            def run(f6, f3, vs2=8):
                a = Asm(0x80000000)
                a.csrs(0x300, 5); a.li(7, rm); a.csrw(0x00A, 7); a.vsetvli(6, 0, Asm.vtype(sew, 1)); a.vop(f6, f3, 4, vs2, 9)
                h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
                h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600)
                h.set_vreg(8, A.tobytes()); h.set_vreg(9, B.tobytes()); h.set_vreg(12, W[:64].tobytes()); h.set_vreg(13, W[64:].tobytes())
                h.run(6)
                return [int(v) for v in np.frombuffer(h.vreg(4), dtype=ut)], h.csr(0x009)
            cases = {
                0x20: [min(x + y, m) for x, y in zip(ua, ub)], 0x22: [max(x - y, 0) for x, y in zip(ua, ub)],
                0x21: [clamp_s(x + y) & m for x, y in zip(sa, sb)], 0x23: [clamp_s(x - y) & m for x, y in zip(sa, sb)],
                0x27: [clamp_s(rnd(x * y, sew - 1, rm)) & m for x, y in zip(sa, sb)],
                0x2A: [rnd(x, y & (sew - 1), rm) & m for x, y in zip(ua, ub)], 0x2B: [rnd(x, y & (sew - 1), rm) & m for x, y in zip(sa, ub)],
            }
            sat_expected = {0x20: any(x + y > m for x, y in zip(ua, ub)), 0x22: any(x < y for x, y in zip(ua, ub)),
                            0x21: any(clamp_s(x + y) != x + y for x, y in zip(sa, sb)), 0x23: any(clamp_s(x - y) != x - y for x, y in zip(sa, sb)),
                            0x27: True, 0x2A: False, 0x2B: False}
            for f6, want in cases.items():
                got, vxsat = run(f6, 0)
                assert got == want and vxsat == int(sat_expected[f6]), (sew, rm, hex(f6))
            for f6, want in {0x08: [rnd(x + y, 1, rm) & m for x, y in zip(ua, ub)], 0x09: [rnd(x + y, 1, rm) & m for x, y in zip(sa, sb)],
                             0x0A: [rnd(x - y, 1, rm) & m for x, y in zip(ua, ub)], 0x0B: [rnd(x - y, 1, rm) & m for x, y in zip(sa, sb)]}.items():
                got, vxsat = run(f6, 2)
                assert got == want and vxsat == 0, (sew, rm, hex(f6))
            if sew < 64:
                wu = [int.from_bytes(W[i * sew // 4:(i + 1) * sew // 4].tobytes(), "little") for i in range(n)]
                ws = [v - (1 << (2 * sew)) if v >> (2 * sew - 1) else v for v in wu]
                got, _ = run(0x2E, 0, vs2=12)
                assert got == [min(rnd(x, y & (2 * sew - 1), rm), m) for x, y in zip(wu, ub)], (sew, rm, "vnclipu")
                got, _ = run(0x2F, 0, vs2=12)
                assert got == [clamp_s(rnd(x, y & (2 * sew - 1), rm)) & m for x, y in zip(ws, ub)], (sew, rm, "vnclip")


def test_rvv_carry_ops_known_answers():
    """vadc / vsbc / vmadc / vmsbc: multi-word addition the way the V spec's example chains them, against Python integers."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(41)
    for sew, ut in [(8, np.uint8), (64, np.uint64)]:
        n = 512 // sew
        A = rng.integers(0, 256, 64, dtype=np.uint8).view(ut).copy(); B = rng.integers(0, 256, 64, dtype=np.uint8).view(ut).copy()
        A[0] = (1 << sew) - 1; B[0] = 0; A[1] = 5; B[1] = 5
        cin = rng.integers(0, 256, 64, dtype=np.uint8); cin[0] |= 3
        m = (1 << sew) - 1
        ua, ub = [int(v) for v in A], [int(v) for v in B]
        c = [(int.from_bytes(cin.tobytes(), "little") >> i) & 1 for i in range(n)]

        def run(f6, vm, vd=4):
            a = Asm(0x80000000)
            a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, 1)); a.vop(f6, 0, vd, 8, 9, vm)
            h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
            h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600)
            h.set_vreg(8, A.tobytes()); h.set_vreg(9, B.tobytes()); h.set_vreg(0, cin.tobytes())
            h.run(3)
            return h
        assert [int(v) for v in np.frombuffer(run(0x10, 0).vreg(4), dtype=ut)] == [(x + y + k) & m for x, y, k in zip(ua, ub, c)]
        assert [int(v) for v in np.frombuffer(run(0x12, 0).vreg(4), dtype=ut)] == [(x - y - k) & m for x, y, k in zip(ua, ub, c)]
        bits = lambda h: [(int.from_bytes(h.vreg(4), "little") >> i) & 1 for i in range(n)]
        assert bits(run(0x11, 0)) == [int(x + y + k > m) for x, y, k in zip(ua, ub, c)]
        assert bits(run(0x11, 1)) == [int(x + y > m) for x, y in zip(ua, ub)]
        assert bits(run(0x13, 0)) == [int(x - y - k < 0) for x, y, k in zip(ua, ub, c)]
        assert bits(run(0x13, 1)) == [int(x < y) for x, y in zip(ua, ub)]
        assert run(0x10, 1).csr(0x342) == 2 and run(0x10, 0, vd=0).csr(0x342) == 2              # reserved: vm = 1, vd = v0


def test_rvv_fp_widening_known_answers():
    """vfwadd / vfwsub (.vv, .wv), vfwmul, the four widening multiply-adds, vfwcvt.f.f.v / vfncvt.f.f.w, vfwredosum, vfclass at
    SEW 32 against numpy in float64 (exact for these operands); SEW 16 and SEW 64 are illegal without Zvfh / a 128-bit type."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(43)
    A = (rng.integers(-4000, 4000, 16) / 16).astype(np.float32); B = (rng.integers(-4000, 4000, 16) / 16).astype(np.float32)
    D = (rng.integers(-4000, 4000, 16) / 16).astype(np.float64)
    A[0] = np.inf; B[0] = 1.0; A[1] = np.nan

    def run(f6, s1=9, f3=1, sew=32, vs2=8, vd=4, wide2=None):
        a = Asm(0x80000000)
        a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, 1)); a.vop(f6, f3, vd, vs2, s1)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600)
        h.set_vreg(8, A.tobytes()); h.set_vreg(9, B.tobytes()); h.set_vreg(4, D[:8].tobytes()); h.set_vreg(5, D[8:].tobytes())
        if wide2 is not None:
            h.set_vreg(12, wide2[:8].tobytes()); h.set_vreg(13, wide2[8:].tobytes())
        h.run(3)
        return h
    wide = lambda h, v=4: np.frombuffer(h.vreg(v) + h.vreg(v + 1), dtype=np.float64)
    same = lambda g, w: bool(((g == w) | (np.isnan(g) & np.isnan(w))).all())
    a64, b64 = A.astype(np.float64), B.astype(np.float64)
    with np.errstate(invalid="ignore"):
        assert same(wide(run(0x30)), a64 + b64) and same(wide(run(0x32)), a64 - b64) and same(wide(run(0x38)), a64 * b64)
        assert same(wide(run(0x34, vs2=12, wide2=D)), D + b64) and same(wide(run(0x36, vs2=12, wide2=D)), D - b64)
        for f6, want in {0x3C: b64 * a64 + D, 0x3D: -(b64 * a64) - D, 0x3E: b64 * a64 - D, 0x3F: -(b64 * a64) + D}.items():
            assert same(wide(run(f6)), want), hex(f6)
        assert same(wide(run(0x12, s1=0x0C)), a64)
        h = run(0x12, s1=0x14, vs2=12, wide2=D)
        assert (np.frombuffer(h.vreg(4), dtype=np.float32)[:16] == D.astype(np.float32)).all()
        h = run(0x33, vd=12, wide2=D)                                                              # vfwredosum.vs v12, v8, v9: v12[0] = v9[0] (f64) + sum
        acc = np.frombuffer(B.tobytes(), dtype=np.float64)[0]
        for v in a64:
            acc = acc + v
        got = np.frombuffer(h.vreg(12), dtype=np.float64)[0]
        assert (np.isnan(got) and np.isnan(acc)) or got == acc
    cls = [int(v) for v in np.frombuffer(run(0x13, s1=0x10).vreg(4), dtype=np.uint32)]
    assert cls[0] == 0x80 and cls[1] == 0x200 and all(c in (0x02, 0x40, 0x08, 0x10) for c in cls[2:])
    assert run(0x30, sew=16).csr(0x342) == 2 and run(0x30, sew=64).csr(0x342) == 2


def test_rvv_indexed_segment_known_answers():
    """vluxseg2ei16.v gathers two-field records at 16-bit byte offsets; vsoxseg2ei16.v scatters them; the index register may not
    overlap the destination registers at all."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(47)
    table = rng.integers(0, 2**32, 512, dtype=np.uint32)
    idx = (rng.permutation(256)[:16] * 8).astype(np.uint16)

    def hart(build):
        a = Asm(0x80000000)
        a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(32, 1))
        build(a)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.write_mem(0x80001000, table.tobytes()); h.pc = 0x80000000
        h.set_x(5, 0x6600); h.set_x(10, 0x80001000); h.set_x(11, 0x80004000)
        h.set_vreg(8, idx.tobytes() + bytes(32))
        return h
    h = hart(lambda a: (a.vls(False, 16, 4, 10, lumop=8, mop=1, nf=1), a.vls(True, 16, 4, 11, lumop=8, mop=3, nf=1)))
    h.run(4)
    rec = table.reshape(256, 2)
    assert (np.frombuffer(h.vreg(4), dtype=np.uint32) == rec[idx // 8, 0]).all() and (np.frombuffer(h.vreg(5), dtype=np.uint32) == rec[idx // 8, 1]).all()
    out = np.frombuffer(h.read_mem(0x80004000, 2048), dtype=np.uint32).reshape(256, 2)
    want = np.zeros((256, 2), dtype=np.uint32); want[idx // 8] = rec[idx // 8]
    assert (out == want).all()
    h = hart(lambda a: a.vls(False, 16, 7, 10, lumop=8, mop=1, nf=1))          # fields land in v7, v8: v8 holds the indices
    h.run(3)
    assert h.csr(0x342) == 2


def test_rvv_fp_int_width_conversions_known_answers():
    """vfwcvt.x.f.v (f32 -> i64), vfwcvt.f.x.v (i16 -> f32, i32 -> f64), vfncvt.x.f.w (f32 -> i16 with clamping, f64 -> i32),
    vfncvt.rtz.xu.f.w, vfncvt.f.x.w (i64 -> f32) against numpy."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(53)

    def run(sew, code, v8, v9=None, vs2=8):
        a = Asm(0x80000000)
        a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(sew, 1)); a.vop(0x12, 1, 4, vs2, code)
        h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
        h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600)
        h.set_vreg(8, v8.tobytes())
        if v9 is not None:
            h.set_vreg(9, v9.tobytes())
        h.run(3)
        return h
    F = (rng.integers(-80000, 80000, 16) / 4).astype(np.float32); F[0] = 2.5; F[1] = -3.5; F[2] = 1e10; F[3] = np.nan
    rne = lambda v: int(np.rint(np.float64(v)))
    h = run(32, 0x09, F)                                                                           # vfwcvt.x.f.v
    got = np.frombuffer(h.vreg(4) + h.vreg(5), dtype=np.int64)
    assert [int(g) for g in got] == [2**63 - 1 if np.isnan(v) else rne(v) for v in F]
    I16 = rng.integers(-2**15, 2**15, 32).astype(np.int16)
    h = run(16, 0x0B, I16)                                                                         # vfwcvt.f.x.v at SEW 16: i16 -> f32
    assert (np.frombuffer(h.vreg(4) + h.vreg(5), dtype=np.float32) == I16.astype(np.float32)).all()
    h = run(16, 0x0A, I16)                                                                         # vfwcvt.f.xu.v
    assert (np.frombuffer(h.vreg(4) + h.vreg(5), dtype=np.float32) == I16.view(np.uint16).astype(np.float32)).all()
    I32 = rng.integers(-2**31, 2**31, 16).astype(np.int32)
    h = run(32, 0x0B, I32)
    assert (np.frombuffer(h.vreg(4) + h.vreg(5), dtype=np.float64) == I32.astype(np.float64)).all()
    W = np.concatenate([F, (rng.integers(-80000, 80000, 16) / 4).astype(np.float32)])              # 32 f32 = the wide source at SEW 16
    h = run(16, 0x11, W[:16], W[16:])                                                              # vfncvt.x.f.w: f32 -> i16, clamped, NV on overflow
    got = np.frombuffer(h.vreg(4), dtype=np.int16)
    assert [int(g) for g in got] == [32767 if np.isnan(v) else max(-32768, min(32767, rne(v))) for v in W]
    assert h.csr(0x003) & 0x10
    h = run(16, 0x16, W[:16], W[16:])                                                              # vfncvt.rtz.xu.f.w: f32 -> u16, truncating
    got = np.frombuffer(h.vreg(4), dtype=np.uint16)
    assert [int(g) for g in got] == [65535 if np.isnan(v) else max(0, min(65535, int(np.float64(v)))) for v in W]
    D = (rng.integers(-2**40, 2**40, 16) / 8).astype(np.float64)
    h = run(32, 0x11, D[:8], D[8:])                                                                # f64 -> i32
    got = np.frombuffer(h.vreg(4), dtype=np.int32)
    assert [int(g) for g in got] == [max(-2**31, min(2**31 - 1, rne(v))) for v in D]
    I64 = rng.integers(-2**62, 2**62, 16).astype(np.int64)
    h = run(32, 0x13, I64[:8], I64[8:])                                                            # vfncvt.f.x.w: i64 -> f32
    assert (np.frombuffer(h.vreg(4), dtype=np.float32) == I64.astype(np.float32)).all()
    assert run(8, 0x0B, I16).csr(0x342) == 2 and run(64, 0x09, D[:8]).csr(0x342) == 2              # f16 side without Zvfh; no 128-bit side


def test_rvv_estimates_known_answers():
    """vfrsqrt7.v / vfrec7.v: 7-bit estimates (relative error < 2^-7 on normal inputs, exact values at 1.0), and the special cases of
    the V spec: zeros -> infinities with DZ, infinities -> zeros, negative / NaN inputs, 1 / subnormal overflowing per frm."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(59)
    for sew, ft, ut in [(32, np.float32, np.uint32), (64, np.float64, np.uint64)]:
        n = 512 // sew
        X = np.exp(rng.uniform(-60, 60, n)).astype(ft)
        X[0] = 1.0; X[1] = 0.0; X[2] = -0.0; X[3] = np.inf; X[4] = -np.inf; X[5] = -2.0; X[6] = np.nan
        X[7] = np.finfo(ft).tiny / 16                                                              # subnormal with more than two leading zeros

        def run(code, frm=0):
            a = Asm(0x80000000)
            a.csrs(0x300, 5); a.li(7, frm << 5); a.csrw(0x003, 7); a.vsetvli(6, 0, Asm.vtype(sew, 1)); a.vop(0x13, 1, 4, 8, code)
            h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
            h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600); h.set_vreg(8, X.tobytes())
            h.run(6)
            return np.frombuffer(h.vreg(4), dtype=ft), h.csr(0x001)
        with np.errstate(all="ignore"):
            r, fl = run(5)                                                                         # vfrec7.v, RNE
            assert r[0] == 0.99609375 and r[1] == np.inf and np.isneginf(r[2]) and r[3] == 0 and r[4] == 0 and np.signbit(r[4])
            assert abs(r[5] * -2.0 - 1) < 2.0**-7 and np.isnan(r[6]) and r[7] == np.inf and fl == 0x08 | 0x05
            assert (np.abs(r[8:].astype(np.float64) * X[8:].astype(np.float64) - 1) < 2.0**-7).all()
            r, fl = run(5, frm=1)                                                                  # RTZ: overflow gives the largest finite value
            assert r[7] == np.finfo(ft).max
            r, fl = run(4)                                                                         # vfrsqrt7.v
            assert r[0] == 0.99609375 and r[1] == np.inf and np.isneginf(r[2]) and r[3] == 0 and np.isnan(r[4]) and np.isnan(r[5]) and np.isnan(r[6])
            assert fl == 0x08 | 0x10
            x = X[7:].astype(np.float64)
            rr = r[7:].astype(np.float64); assert (np.abs(rr * x * rr - 1) < 2.0**-5.9).all()


def test_rvv_round_to_odd_narrowing():
    """vfncvt.rod.f.f.w: f64 -> f32 truncated toward zero with the last bit set when inexact (overflow gives the largest finite
    value), checked against a numpy construction on normal, subnormal-result, overflowing and exact operands."""
    import numpy as np
    from whisper_amd.rvasm import Asm
    rng = np.random.default_rng(61)
    D = np.concatenate([np.exp(rng.uniform(-110, 95, 10)) * rng.choice([-1, 1], 10), [1.5, -2.0**-140, 1e39, -1e300, 2.0**-149 * 1.25, 0.1]]).astype(np.float64)
    a = Asm(0x80000000)
    a.csrs(0x300, 5); a.vsetvli(6, 0, Asm.vtype(32, 1)); a.vop(0x12, 1, 4, 8, 0x15)
    h = rvoracle.Hart(isa="rv64imafdcv_zicsr", mem_base=0x80000000, mem_size=1 << 16)
    h.write_mem(0x80000000, a.assemble()); h.pc = 0x80000000; h.set_x(5, 0x6600)
    h.set_vreg(8, D[:8].tobytes()); h.set_vreg(9, D[8:].tobytes())
    h.run(3)
    got = np.frombuffer(h.vreg(4), dtype=np.uint32)
    want = []
    with np.errstate(all="ignore"):
        for x in D:
            y = np.float32(x)
            if np.float64(y) == x:
                want.append(int(y.view(np.uint32))); continue
            t = np.nextafter(y, np.float32(0)) if abs(np.float64(y)) > abs(x) else y
            want.append(int(t.view(np.uint32)) | 1)
    assert [int(g) for g in got] == want


@pytest.mark.skipif(rvoracle.ref() is None, reason="oracle/_ref (reference Disassembler) not built")
def test_tracefmt_prints_the_reference_log_lines():
    """whisper_amd.tracefmt (the host formatter of the device commit ring) on records shaped like the ring's, taken here from the
    oracle's log of the FP / trap capture: every line equals the reference's captured log line, disassembly column included."""
    from whisper_amd import tracefmt
    g, h, st, log = _run_fp_trap()
    gold = _group(g["records"])
    for c in log:
        if c.tag not in gold:
            continue
        recs = gold[c.tag]
        rec = (c.pc, c.inst, c.size, c.priv, c.trap, list(c.changes()) or [(1, 0, 0)])
        dis = (lambda inst: "illegal") if c.size == 2 and (c.inst & 0xffff) == 0 else (lambda inst: rvoracle.ref_disassemble(inst))
        got = tracefmt.format_record(c.tag, 0, rec, disasm=dis)
        want = [(f"#{r['tag']} {r['hart']}  {r['priv']} {r['pc']:016x} {r['op']:>8s} {r['kind']} {r['key']:016x} {r['val']:016x} {r['text']}").rstrip()
                + ("  +" if r["cont"] else "") for r in recs]
        if recs[-1]["cont"]:
            got = got[:len(want)]
        assert got == want, c.tag
