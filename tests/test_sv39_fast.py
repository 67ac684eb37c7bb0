"""The TLB-hit fast path for S / U-mode code under Sv39 (Lane::fast_step_translated; on the GPU wsp_swarm): VirtMem::translate's hit
case (VirtMem.cpp:261-310), Tlb::findEntryUpdateTime (Tlb.hpp:91-100), the fetch / load / store glue of Hart.cpp:3052-3206,
1988-2154 - including the reference's second lookup for the upper half of a 4-byte instruction at a 2-byte-aligned pc whenever
VA != PA (Hart.cpp:3144-3156).  The state digest folds in every TLB word, so a replacement counter that is off by one fails."""
import pytest

import simutil as S
import whisper_amd as W
from simutil import rvoracle
from whisper_amd import workloads as WL


def test_fast_translated_steps_equal_the_oracle_on_paging_fuzz():
    """Sv39 fuzz programs (tiny TLB, stale entries, faults of every kind, SUM / MXR / satp changes) with the body pre-decoded so
    that S / U-mode steps take the fast step whenever they qualify: result record == the oracle's, and the path is exercised."""
    fast = 0
    for seed in range(48):
        p = WL.sv39_program(seed, exec_body=True)
        ro, _, _ = S.run_oracle(p)
        rh = S.run_hostsim(p)[0]
        assert rh == ro, (seed, {k: (ro[k], rh[k]) for k in ro if ro[k] != rh[k]})
        fast += S.last_fast_tr_steps
    assert fast > 20000


@pytest.mark.parametrize("fp", [False, True])
def test_compute_kernel_in_s_mode_under_sv39(fp):
    """The bench kernel booted the way an OS runs it: > 98 % of its steps are fast translated steps; == the oracle."""
    p = WL.compute_kernel(iters=1, fp=fp, paged=True)
    ro, h, _ = S.run_oracle(p, seed_a0=WL.compute_seed(3), seed_mem=3)
    rh = S.run_hostsim(p, seed_a0=WL.compute_seed(3), seed_mem=3)[0]
    assert ro["status"] == rvoracle.STOP_TOHOST and ro["value"] == 1 and h.priv == 1
    assert rh == ro
    assert S.last_fast_tr_steps > 0.98 * ro["exec"]


@pytest.mark.gpu
def test_paged_compute_kernel_on_the_gpu():
    """4096 S-mode instances (per-instance seeds) through the C ABI, traced and untraced: every hart passes; sampled harts == the
    oracle (state digest with TLB contents, memory digest, commit digest)."""
    n = 4096
    p = WL.compute_kernel(iters=1, fp=True, paged=True)
    for trace in (True, False):
        b = W.Batch(n, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size)
        try:
            b.load_elf_bytes(p.elf)
            b.set_stop(max_inst=p.max_inst)
            b.poke_each(W.RES_X, 10, [WL.compute_seed(i) for i in range(n)])
            b.poke_each(W.RES_MEM, p.symbols["seed"], list(range(n)))
            assert b.run(trace_hash=trace) == 0
            res = b.results()
            assert all(r.status == W.ST_STOP_TOHOST and r.stop_value == 1 for r in res)
            for i in (0, 1, 63, 64, 1000, 4095):
                ro, _, _ = S.run_oracle(p, seed_a0=WL.compute_seed(i), seed_mem=i, hart_id=i)
                got = res[i].as_dict()
                if not trace:
                    ro = dict(ro, trace_hash=0)
                assert got == ro, (trace, i)
        finally:
            b.close()


@pytest.mark.gpu
def test_paging_fuzz_with_predecoded_bodies_on_the_gpu():
    """96 distinct Sv39 fuzz programs x 8 replicas in one batch, bodies pre-decoded: == the oracle, hart for hart."""
    progs = [WL.sv39_program(s, exec_body=True) for s in range(96)]
    reps = 8
    p0 = progs[0]
    b = W.Batch(len(progs) * reps, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, tlb_entries=p0.meta.get("tlb_entries", 0))
    try:
        b.load_elfs([p.elf for p in progs], harts_per_elf=reps)
        for i, p in enumerate(progs):
            b.set_stop(max_inst=p.max_inst, lo=i * reps, hi=(i + 1) * reps)
        assert b.run(trace_hash=True) == 0
        res = b.results()
        for i, p in enumerate(progs):
            for k in (0, reps - 1):
                h = i * reps + k
                ro, _, _ = S.run_oracle(p, hart_id=h)
                assert res[h].as_dict() == ro, (i, k)
    finally:
        b.close()
