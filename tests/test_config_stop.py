"""Whisper JSON configuration front end (wsp_config_parse / wsp_batch_desc.config_json) and the retired-count / stop-address stop
conditions.  CPU part: the parser's three classes of tag (honoured, accepted, rejected) and, per honoured switch, the device core
(host build, tests/host_shim) against the oracle on programs in which the switch changes the outcome.  GPU part (-m gpu): the same
through the C ABI."""
import json

import pytest

import simutil as S
from simutil import rvoracle
import whisper_amd as W
from whisper_amd import workloads as WL
from whisper_amd.rvasm import Asm

F_NO_MISAL, F_CLR_ILL, F_CLR_EBREAK, F_CANCEL_LR, F_KEEP_MPRV, F_NO_SV39 = (1 << k for k in range(16, 22))
SWITCHES = [({"enable_misaligned_data": False}, F_NO_MISAL), ({"clear_mtval_on_illegal_instruction": True}, F_CLR_ILL),
            ({"clear_mtval_on_ebreak": "true"}, F_CLR_EBREAK), ({"cancel_lr_on_trap": 1}, F_CANCEL_LR),
            ({"clear_mprv_on_ret": False}, F_KEEP_MPRV), ({"address_translation_modes": ["bare"]}, F_NO_SV39)]


def test_config_parse_honoured_tags():
    text = """{ // comments are allowed (nlohmann ignore_comments)
      "isa": "rv64imafdcvsu_zicsr", "xlen": 64, "reset_vec": "0x80000100", /* hex strings */
      "memmap": { "size": "0x40000", "page_size": 4096 }, "tlb_entries": 64,
      "vector": { "bytes_per_vec": 32, "max_bytes_per_elem": 8, "min_bytes_per_elem": 1, "tail_agnostic_policy": "ones" },
      "enable_misaligned_data": false, "clear_mtval_on_illegal_instruction": true, "cancel_lr_on_trap": "1",
      "abi_names": true, "trace_ptw": true, "nmi_vec": "0x1000", "enable_pmp_tor": true, "harts": 1,
      "csr": { "mtvec": { "reset": "0x80000040" }, "mscratch": { "reset": 5 } } }"""
    c = W.parse_config(text)
    assert (c.xlen, c.isa, c.mem_size, c.has_reset_pc, c.reset_pc) == (64, b"rv64imafdcvsu_zicsr", 0x40000, 1, 0x80000100)
    assert (c.tlb_entries, c.vlen_bytes, c.abi_names) == (64, 32, 1)
    assert c.features == F_NO_MISAL | F_CLR_ILL | F_CANCEL_LR
    assert [(c.csr_num[i], c.csr_value[i]) for i in range(c.n_csr_resets)] == [(0x305, 0x80000040), (0x340, 5)]
    for cfg, bit in SWITCHES:
        assert W.parse_config(json.dumps(cfg)).features == bit
    assert W.parse_config("{}").features == 0


@pytest.mark.parametrize("cfg", [
    {"enable_triggers": True}, {"memmap": {"consoleio": "0x10000000"}}, {"clint": "0x2000000"}, {"aclint": {"base": 0}},
    {"harts": 2}, {"vector": {"bytes_per_vec": 128, "max_bytes_per_elem": 8}}, {"vector": {"bytes_per_vec": 32, "max_bytes_per_elem": 4}},
    {"vector": {"bytes_per_vec": 32, "max_bytes_per_elem": 8, "mask_agnostic_policy": "undisturb"}},
    {"csr": {"mstatus": {"mask": "0x0"}}}, {"csr": {"mhpmcounter3": {"reset": 1}}}, {"physical_memory_protection_grain": 16},
    {"force_rounding_mode": "rne"}, {"enable_pmp_na4": False}, {"some_future_tag": 1}, {"reservation_bytes": 64}])
def test_config_rejects_what_is_not_modelled(cfg):
    with pytest.raises(W.WspError) as ei:
        W.parse_config(json.dumps(cfg))
    assert ei.value.code == W.EUNSUPPORTED
    tag = next(iter(cfg))
    assert tag in str(ei.value)


def test_config_malformed():
    for text, code in (("{", W.EFORMAT), ("[1]", W.EINVAL), ('{"tlb_entries": 48}', W.EINVAL), ('{"isa": 5}', W.EINVAL),
                       ('{"vector": {"max_bytes_per_elem": 8}}', W.EINVAL), ('{"reset_vec": "zz"}', W.EINVAL)):
        with pytest.raises(W.WspError) as ei:
            W.parse_config(text)
        assert ei.value.code == code, text


def _both(p, **kw):
    a, _, _ = S.run_hostsim(p, **kw)
    b, _, log = S.run_oracle(p, log_cap=40000, **kw)
    assert a == b, (p.name, kw, a, b)
    return b, log


def _mtvals(log):
    return [(dict((key, val) for k, key, val in c.changes() if k == 5)) for c in log if c.trap == 1]


def test_switches_change_the_outcome_and_device_core_follows():
    """Each switch on the fuzz / PMP / Sv39 programs: device core == oracle with the switch, and the run differs from the default
    run in the way the reference's code says (Hart.cpp:2013, 3256, 12097, 3570, 12283)."""
    progs = [WL.fuzz_program(s, 300) for s in range(40)]
    base = [_both(p) for p in progs]
    # misaligned data: cause 4 / 6 appear only with the switch
    n46 = 0
    for p, (r0, log0) in zip(progs, base):
        r1, log1 = _both(p, features=F_NO_MISAL)
        assert not any(t.get(0x342) in (4, 6) for t in _mtvals(log0))
        n46 += sum(t.get(0x342) in (4, 6) for t in _mtvals(log1))
    assert n46 > 20
    # mtval of illegal instructions / ebreak
    nz = 0
    for p, (r0, log0) in zip(progs, base):
        r1, log1 = _both(p, features=F_CLR_ILL | F_CLR_EBREAK)
        nz += sum(t.get(0x342) in (2, 3) and t.get(0x343, 0) != 0 for t in _mtvals(log0))
        assert not any(t.get(0x342) in (2, 3) and 0x343 in t and t[0x343] != 0 for t in _mtvals(log1))
    assert nz > 20
    # LR cancelled by traps (the generated programs seldom trap between an LR and its SC: test_cancel_lr_on_trap_directed does)
    for s in range(30):
        _both(WL.pmp_program(s), features=F_CANCEL_LR)
    # MPRV kept over mret / no sv39: Sv39 programs (MPRV services, satp writes)
    d_mprv = d_sv = 0
    for s in range(24):
        p = WL.sv39_program(s)
        r0, _ = _both(p)
        r1, _ = _both(p, features=F_KEEP_MPRV)
        r2, _ = _both(p, features=F_NO_SV39)
        d_mprv += r0 != r1; d_sv += r0 != r2
    assert d_sv >= 20 and d_mprv >= 0


def _lr_trap_sc_program():
    """lr.w ; ecall (handler returns past it) ; sc.w -> a1: 0 by default, 1 with cancel_lr_on_trap."""
    a = Asm(0x80000000)
    a.la(5, "handler"); a.csrw(0x305, 5)
    a.la(6, "cell"); a.lr_w(7, 6); a.ecall(); a.sc_w(11, 7, 6)
    a.la(5, "tohost"); a.li(6, 1); a.sd(6, 0, 5); a.label("spin"); a.j("spin")
    a.align(4); a.label("handler"); a.csrr(28, 0x341); a.addi(28, 28, 4); a.csrw(0x341, 28); a.mret()
    a.align(8); a.label("cell"); a.dword(0x1234); a.label("tohost"); a.dword(0)
    return WL._finish(a, "lr_trap_sc", "rv64imac_zicsr")


def _mprv_over_mret_program():
    """M sets mstatus.MPRV, returns to U with mret, U traps back: a1 = mstatus in the handler.  MPRV survives the mret only with
    clear_mprv_on_ret = false (Hart.cpp:12283)."""
    a = Asm(0x80000000)
    a.la(5, "handler"); a.csrw(0x305, 5)
    a.li(5, -1); a.csrw(0x3B0, 5); a.li(5, 0x1f); a.csrw(0x3A0, 5)          # PMP open for U
    a.li(5, 1 << 17); a.csrs(0x300, 5); a.li(5, 3 << 11); a.csrc(0x300, 5)
    a.la(5, "user"); a.csrw(0x341, 5); a.mret()
    a.align(4); a.label("user"); a.ecall()
    a.align(4); a.label("handler"); a.csrr(11, 0x300)
    a.la(5, "tohost"); a.li(6, 1); a.sd(6, 0, 5); a.label("spin"); a.j("spin")
    a.align(8); a.label("tohost"); a.dword(0)
    return WL._finish(a, "mprv_over_mret", "rv64imacsu_zicsr")


def test_clear_mprv_on_ret_directed():
    p = _mprv_over_mret_program()
    for feat, want in ((0, 0), (F_KEEP_MPRV, 1)):
        r, h, _ = S.run_oracle(p, features=feat)
        assert r["status"] == rvoracle.STOP_TOHOST and (h.x(11) >> 17) & 1 == want
        a, xs, _ = S.run_hostsim(p, features=feat)
        assert a == r and (xs[11] >> 17) & 1 == want


def test_cancel_lr_on_trap_directed():
    p = _lr_trap_sc_program()
    for feat, want in ((0, 0), (F_CANCEL_LR, 1)):
        r, h, _ = S.run_oracle(p, features=feat)
        assert r["status"] == rvoracle.STOP_TOHOST and h.x(11) == want
        a, xs, _ = S.run_hostsim(p, features=feat)
        assert a == r and xs[11] == want


def test_stop_address_and_retired_limit_device_core_vs_oracle():
    """Hart.cpp:6025-6026: `while (pc != address and executed < instLim and retired < retLim)`."""
    p = WL.compute_kernel(iters=1)
    _, _, log = S.run_oracle(p, log_cap=3000, max_inst=3000)
    pcs = [c.pc for c in log]
    for k in (0, 1, 17, 500, 2999):
        r, _ = _both(p, stop_pc=pcs[k], max_inst=3000)
        first = pcs.index(pcs[k])
        assert r["status"] == rvoracle.STOP_ADDR and r["pc"] == pcs[k] and r["exec"] == first
    # retired lags executed by the trapped instructions: fuzz programs trap a lot
    seen = 0
    for s in range(30):
        q = WL.fuzz_program(s, 300)
        full, _ = _both(q)
        if full["exec"] == full["retired"] or full["retired"] < 60:
            continue
        for lim in (1, 50, full["retired"] - 1):
            r, _ = _both(q, max_ret=lim)
            assert r["status"] == rvoracle.STOP_LIMIT and r["retired"] == lim and r["exec"] >= lim
            seen += r["exec"] > lim
        r, _ = _both(q, max_ret=full["retired"] + 5)
        assert r == full
    assert seen > 10


# ---------------------------------------------------------------------------------------------------------------- GPU
def _gpu_run(progs, config=None, stops=None, lanes=0, **kw):
    p0 = progs[0]
    b = W.Batch(len(progs), isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=lanes,
                tlb_entries=p0.meta.get("tlb_entries", 0), config=config, **kw)
    try:
        for i, p in enumerate(progs):
            b.load_elf_bytes(p.elf, i, i + 1)
            st = stops[i] if stops else {}
            b.set_stop(max_inst=st.get("max_inst", p.max_inst), max_ret=st.get("max_ret", W.NO_LIMIT), stop_pc=st.get("stop_pc"), lo=i, hi=i + 1)
        assert b.run(trace_hash=True) == 0
        return [r.as_dict() for r in b.results()]
    finally:
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,bit", SWITCHES)
def test_gpu_config_switch_parity(cfg, bit):
    kind = WL.sv39_program if bit in (F_KEEP_MPRV, F_NO_SV39) else WL.fuzz_program
    progs = [_lr_trap_sc_program()] * 5 if bit == F_CANCEL_LR else [_mprv_over_mret_program()] * 3 if bit == F_KEEP_MPRV else [kind(s) for s in range(48)]
    res = _gpu_run(progs, config=json.dumps(cfg))
    plain = 0
    for i, p in enumerate(progs):
        ro, _, _ = S.run_oracle(p, hart_id=i, features=bit)
        assert res[i] == ro, (cfg, p.name)
        plain += S.run_oracle(p, hart_id=i)[0] != ro
    assert plain >= 1          # the switch mattered on this set


@pytest.mark.gpu
def test_gpu_config_file_fields_and_csr_reset():
    """isa, memmap.size, reset_vec, tlb_entries and csr reset values taken from the configuration text; descriptor fields override."""
    p = _lr_trap_sc_program()
    cfg = {"isa": p.isa, "memmap": {"size": hex(p.mem_size)}, "reset_vec": hex(p.entry + 0), "cancel_lr_on_trap": True,
           "csr": {"mscratch": {"reset": "0xabcdef"}, "mtvec": {"reset": hex(p.mem_base + 0x40)}}}
    b = W.Batch(3, isa=None, mem_base=p.mem_base, mem_size=0, config=json.dumps(cfg))
    try:
        assert b.peek(2, W.RES_CSR, 0x340) == 0xabcdef and b.peek(0, W.RES_CSR, 0x305) == p.mem_base + 0x40
        assert b.peek(1, W.RES_PC, 0) == p.entry
        b.load_elf_bytes(p.elf)
        assert b.run() == 0
        assert [b.peek(i, W.RES_X, 11) for i in range(3)] == [1, 1, 1]
    finally:
        b.close()
    with pytest.raises(W.WspError) as ei:
        W.Batch(1, isa=None, mem_base=p.mem_base, mem_size=1 << 16, config='{"isa": "rv64imac", "enable_triggers": true}')
    assert ei.value.code == W.EUNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 8])
def test_gpu_stop_address_and_retired_limit(lanes):
    p = WL.compute_kernel(iters=1)
    _, _, log = S.run_oracle(p, log_cap=3000, max_inst=3000)
    pcs = [c.pc for c in log]
    # one shared program image, a different stop address per hart (and some without): the OP_STOPCHK records of the others are skipped
    ks = [0, 1, 17, 500, 2999, None, 40, None]
    stops = [({"stop_pc": pcs[k], "max_inst": 3000} if k is not None else {"max_inst": 3000}) for k in ks]
    res = _gpu_run([p] * len(ks), stops=stops, lanes=lanes)
    for i, k in enumerate(ks):
        ro, _, _ = S.run_oracle(p, hart_id=i, max_inst=3000, stop_pc=pcs[k] if k is not None else None)
        assert res[i] == ro, (i, k)
        assert res[i]["status"] == (W.ST_STOP_ADDR if k is not None else W.ST_STOP_LIMIT)
    progs = [WL.fuzz_program(s, 300) for s in range(40)]
    full = [S.run_oracle(q, hart_id=i)[0] for i, q in enumerate(progs)]
    stops = [{"max_ret": max(1, f["retired"] // 2)} for f in full]
    res = _gpu_run(progs, stops=stops, lanes=lanes)
    lag = 0
    for i, q in enumerate(progs):
        ro, _, _ = S.run_oracle(q, hart_id=i, max_ret=stops[i]["max_ret"])
        assert res[i] == ro, i
        lag += ro["exec"] > ro["retired"]
    assert lag > 10


@pytest.mark.gpu
@pytest.mark.parametrize("trace", [True, False])
def test_gpu_stop_address_is_virtual_under_sv39(trace):
    """The stop address is compared with the pc (Hart.cpp:6025), a virtual address: S / U code mapped at VA != PA runs on the
    translated warm path, whose records sit at physical slots - the pc compare there ends the run, not a record patch."""
    progs, stops, want = [], [], []
    for s in range(24):
        p = WL.sv39_program(s)
        _, _, log = S.run_oracle(p, log_cap=p.max_inst)
        body = [k for k, c in enumerate(log) if 0x10000 <= c.pc < 0x14000]
        if len(body) < 40:
            continue
        k = body[len(body) // 2 + (s % 7)]
        progs.append(p); stops.append({"stop_pc": log[k].pc})
        ro, _, _ = S.run_oracle(p, hart_id=len(progs) - 1, stop_pc=log[k].pc)
        assert ro["status"] == rvoracle.STOP_ADDR and ro["pc"] == log[k].pc
        want.append(ro)
    assert len(progs) >= 12
    p0 = progs[0]
    b = W.Batch(len(progs), isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, tlb_entries=p0.meta.get("tlb_entries", 0))
    try:
        for i, p in enumerate(progs):
            b.load_elf_bytes(p.elf, i, i + 1)
            b.set_stop(max_inst=p.max_inst, stop_pc=stops[i]["stop_pc"], lo=i, hi=i + 1)
        assert b.run(trace_hash=trace) == 0
        res = [r.as_dict() for r in b.results()]
    finally:
        b.close()
    for i, ro in enumerate(want):
        ro = dict(ro)
        if not trace:
            ro["trace_hash"] = 0
        assert res[i] == ro, (i, progs[i].name)
