"""The hand-written converged loop (whisper_amd/csrc/wsp_conv_core.h) against the oracle.

That loop runs UNTRACED waves whose harts share one program image - the bench's mode - so every test here runs with
trace_hash=False and loads each program into a whole wave (or several); compared per hart: status, stop value, pc, executed /
retired counts, state digest (all x / f registers and CSRs) and memory digest.  The traced twins of these runs (the C++ loop)
live in test_gpu_parity.py."""
import pytest

import simutil as S
import whisper_amd as W
from whisper_amd import workloads as WL
from whisper_amd.rvasm import Asm, CSR

pytestmark = pytest.mark.gpu


def _run_replicated(progs, reps, lanes, seeds=None, max_steps=W.NO_LIMIT):
    p0 = progs[0]
    n = len(progs) * reps
    b = W.Batch(n, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=lanes, tlb_entries=p0.meta.get("tlb_entries", 0),
                vlen_bytes=p0.meta.get("vlenb", 0))
    try:
        for i, p in enumerate(progs):
            b.load_elf_bytes(p.elf, i * reps, (i + 1) * reps)
            b.set_stop(max_inst=p.max_inst, lo=i * reps, hi=(i + 1) * reps)
        if seeds is not None:
            b.poke_each(W.RES_X, 10, [s[0] for s in seeds])
            b.poke_each(W.RES_MEM, p0.symbols["seed"], [s[1] for s in seeds])
        left = n
        while left:
            left = b.run(max_steps, trace_hash=False)
        return [r.as_dict() for r in b.results()]
    finally:
        b.close()


def _oracle(p, hart_id=0, seed=None):
    ro, _, _ = S.run_oracle(p, seed_a0=seed[0] if seed else None, seed_mem=seed[1] if seed else None, hart_id=hart_id)
    ro = dict(ro); ro["trace_hash"] = 0
    return ro


@pytest.mark.parametrize("n,lanes,steps", [(96, 4, W.NO_LIMIT), (200, 64, W.NO_LIMIT), (64, 1, W.NO_LIMIT), (96, 8, 5003)])
def test_compute_kernel_untraced(n, lanes, steps):
    """Per-instance seeds: the harts of a wave diverge in the CRC / list sections and reconverge (min-pc scheduling); the last
    case runs in bounded slices, so the register file goes through LDS and HBM many times."""
    p = WL.compute_kernel(iters=1)
    seeds = [(WL.compute_seed(i), i) for i in range(n)]
    res = _run_replicated([p], n, lanes, seeds=seeds, max_steps=steps)
    bad = [(i, res[i]) for i in range(n) if res[i] != _oracle(p, i, seeds[i])]
    assert not bad, f"{len(bad)} of {n} harts differ; first: {bad[0]} want {_oracle(p, bad[0][0], seeds[bad[0][0]])}"


def test_compute_kernel_4096_untraced_bench_layout():
    """Exactly what bench.py times: 4096 instances, lanes_per_wave = 0 (4 harts per wave), untraced; every hart against the oracle."""
    p = WL.compute_kernel(iters=1)
    n = 4096
    seeds = [(WL.compute_seed(i), i) for i in range(n)]
    res = _run_replicated([p], n, 0, seeds=seeds)
    bad = [i for i in range(n) if res[i] != _oracle(p, i, seeds[i])]
    assert not bad, f"{len(bad)} of {n} harts differ; first: hart {bad[0]}: {res[bad[0]]}"


def test_archtest_programs_untraced_replicated():
    """One program per I / M / C mnemonic with corner operands: every handler of the loop's table, four replicas per wave."""
    progs = WL.archtest_suite()
    res = _run_replicated(progs, 4, 4)
    for i, p in enumerate(progs):
        want = _oracle(p)
        for k in range(4):
            got = dict(res[4 * i + k])
            assert got == want, (p.name, k, got, want)


@pytest.mark.parametrize("kind", ["fuzz", "rv32", "rv32f", "bitmanip", "pmp", "sv39"])
def test_random_programs_untraced_replicated(kind):
    """Random programs (traps, illegal words, accesses outside the arena, CSR traffic, S / U code): the loop hands every instruction
    outside its subset to the warm / general paths and takes the harts back."""
    gen = {"fuzz": lambda s: WL.fuzz_program(s, 300), "rv32": lambda s: WL.rv32_program(s), "rv32f": lambda s: WL.rv32_program(s, fp=True),
           "bitmanip": lambda s: WL.bitmanip_program(s), "pmp": WL.pmp_program, "sv39": WL.sv39_program}[kind]
    progs = [gen(s) for s in range(300, 364)]
    res = _run_replicated(progs, 8, 8)
    for i, p in enumerate(progs):
        want = _oracle(p, hart_id=8 * i)
        for k in range(8):
            got = dict(res[8 * i + k])
            if kind in ("pmp", "sv39", "fuzz"):
                want_k = want                                    # these programs do not read mhartid
            else:
                want_k = _oracle(p, hart_id=8 * i + k)
            assert got == want_k, (kind, p.name, k, got, want_k)


def _edge_program():
    """What the loop must NOT execute itself, next to what it must: misaligned and out-of-arena accesses, loads from and stores to
    tohost / fromhost, stores into the text, jalr to an odd address and out of the text, a branch target outside the text taken and
    not taken, x0 as destination and source, jal / jalr links, every load / store width at the arena's last bytes."""
    BASE = WL.BASE
    data_base = BASE + 0x2000
    a, d = Asm(BASE), Asm(data_base)
    d.label("tohost"); d.dword(0); d.label("fromhost"); d.dword(0); d.label("seed"); d.dword(0)
    d.label("buf"); d.zeros(256)
    d.label("res"); d.zeros(512)
    S_ = d.symbols
    a.label("_start")
    a.la(5, "trap"); a.csrw(CSR["mtvec"], 5)
    a.li(5, 0x6000); a.csrs(CSR["mstatus"], 5)
    a.li(8, S_["buf"]); a.li(9, S_["res"]); a.li(20, 0x1122334455667788); a.li(21, -2); a.li(22, 0)
    # x0 forms
    a.add(0, 20, 21); a.add(10, 0, 20); a.sd(10, 0, 9); a.sub(11, 0, 20); a.sd(11, 8, 9); a.lw(0, 0, 8)
    # every width, aligned, then misaligned (allowed by default: general path), then the last bytes of the arena
    a.sd(20, 0, 8); a.lb(12, 1, 8); a.lbu(13, 7, 8); a.lh(14, 2, 8); a.lhu(15, 6, 8); a.lw(16, 4, 8); a.lwu(17, 4, 8); a.ld(18, 0, 8)
    for i, r in enumerate((12, 13, 14, 15, 16, 17, 18)):
        a.sd(r, 16 + 8 * i, 9)
    a.lw(12, 1, 8); a.ld(13, 3, 8); a.lh(14, 7, 8); a.sw(20, 9, 8); a.sh(20, 15, 8); a.sd(21, 21, 8)
    a.sd(12, 80, 9); a.sd(13, 88, 9); a.sd(14, 96, 9)
    a.li(5, BASE + (1 << 16) - 8); a.sd(20, 0, 5); a.ld(12, 0, 5); a.lw(13, 4, 5); a.lb(14, 7, 5); a.sd(12, 104, 9)
    a.lw(15, 6, 5)                                   # crosses the end of the arena: access fault
    a.ld(15, 8, 5)                                   # outside
    a.sb(20, 8, 5)                                   # outside
    a.li(5, 0x1000); a.lw(15, 0, 5); a.sw(20, 0, 5)  # far outside
    a.li(5, -8); a.ld(15, 0, 5)                      # wraps
    # tohost / fromhost: loads read, a store of 0 to tohost is an ordinary store for the general path
    a.li(5, S_["tohost"]); a.ld(12, 0, 5); a.lw(13, 4, 5); a.ld(14, 8, 5); a.sd(0, 8, 5); a.sw(0, 12, 5)
    # stores into the text make the hart decode from RAM from then on (rewrites an instruction with itself)
    a.la(5, "patch"); a.lw(12, 0, 5); a.sw(12, 0, 5)
    a.label("patch"); a.addi(22, 22, 5); a.sd(22, 112, 9)
    # links
    a.jal(1, "l1"); a.label("l1"); a.sd(1, 120, 9)
    a.la(5, "l2"); a.jalr(1, 5, 0); a.label("l2"); a.sd(1, 128, 9)
    a.la(5, "l3"); a.addi(5, 5, 1); a.jalr(5, 5, 0); a.label("l3"); a.sd(5, 136, 9)    # odd target: bit 0 cleared; rd == rs1
    # jalr out of the text and out of the arena: instruction access fault, the handler returns to ra
    a.la(1, "l4"); a.li(5, 0x4000); a.jalr(0, 5, 0); a.label("l4")
    a.la(1, "l5"); a.li(5, data_base + 64); a.jalr(0, 5, 0); a.label("l5")             # inside the arena, outside the text: zeros = illegal
    a.addi(22, 22, 7); a.sd(22, 144, 9)
    a.li(12, 3); a.fmv_d_x(1, 20); a.fsd(1, 152, 9); a.fld(2, 152, 9); a.fsw(2, 164, 9); a.flw(3, 164, 9); a.fsd(3, 168, 9)
    a.fsd(1, 3, 8); a.fld(4, 3, 8); a.fsd(4, 176, 9)                                    # misaligned FP access
    a.csrr(12, CSR["mcause"]); a.sd(12, 184, 9); a.csrr(12, CSR["mepc"]); a.sd(12, 192, 9)
    WL._exit_tohost(a, S_["tohost"])
    # trap handler: count, then skip the instruction (return to ra for fetch faults)
    a.label("trap")
    a.addi(22, 22, 1)
    a.csrr(30, CSR["mcause"]); a.li(31, 1); a.beq(30, 31, "fetchfault"); a.li(31, 2); a.beq(30, 31, "illegal_at_data")
    a.csrr(30, CSR["mepc"]); a.lhu(31, 0, 30); a.andi(31, 31, 3); a.addi(30, 30, 2); a.li(29, 3); a.bne(31, 29, "t2"); a.addi(30, 30, 2)
    a.label("t2"); a.csrw(CSR["mepc"], 30); a.mret()
    a.label("fetchfault"); a.csrw(CSR["mepc"], 1); a.mret()
    a.label("illegal_at_data"); a.csrw(CSR["mepc"], 1); a.mret()
    return WL._finish(a, "conv-edge", "rv64imafdc_zicsr", d, mem_size=1 << 16, max_inst=20000)


def test_edge_cases_untraced_replicated():
    p = _edge_program()
    want = _oracle(p)
    assert want["status"] == W.ST_STOP_TOHOST and want["value"] == 1
    for reps, lanes in ((4, 4), (64, 64), (1, 1)):
        res = _run_replicated([p], reps, lanes)
        for k in range(reps):
            assert dict(res[k]) == want, (reps, k, res[k], want)
    # the traced run (C++ loop) agrees with the same oracle run including the commit digest
    b = W.Batch(2, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size, lanes_per_wave=2)
    try:
        b.load_elf_bytes(p.elf); b.set_stop(max_inst=p.max_inst)
        assert b.run(trace_hash=True) == 0
        ro, _, _ = S.run_oracle(p)
        assert b.results()[0].as_dict() == ro
    finally:
        b.close()
