"""PMP restatement (oracle) vs the reference's own PmpManager compiled into oracle/_ref: region matching for random
TOR/NA4/NAPOT/locked configurations at M, S and U privilege, and pmpcfg legalisation.  CPU only."""
import ctypes as C
import random

import pytest

from simutil import rvoracle

R = rvoracle.ref()
pytestmark = pytest.mark.skipif(R is None or not hasattr(R, "ref_pmp_new"), reason="oracle/_ref not built")
if R is not None and hasattr(R, "ref_pmp_new"):
    R.ref_pmp_new.restype = C.c_void_p
    R.ref_pmp_new.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    R.ref_pmp_free.argtypes = [C.c_void_p]
    R.ref_pmp_perm.argtypes = [C.c_void_p, C.c_uint, C.c_uint64]
    R.ref_pmp_legalize_cfg.restype = C.c_uint64
    R.ref_pmp_legalize_cfg.argtypes = [C.c_uint64, C.c_uint64]

BASE = 0x80000000
PMASK = 0x003FFFFFFFFFFFFF


def _napot(base, log2size):
    return ((base >> 2) | ((1 << (log2size - 3)) - 1)) & PMASK


def _random_config(rng):
    cfg = [0] * 64
    addr = [0] * 64
    n = rng.choice([1, 2, 3, 5, 8, 16])
    slots = sorted(rng.sample(range(64), n))
    for ix in slots:
        typ = rng.choice([1, 1, 2, 3, 3])
        rwx = rng.choice([0, 1, 3, 4, 5, 7])
        lock = 0x80 if rng.random() < 0.25 else 0
        cfg[ix] = lock | (typ << 3) | rwx
        a = BASE + rng.randrange(0, 1 << 16)
        if typ == 1:
            addr[ix] = (a >> 2) & PMASK
            if ix and rng.random() < 0.7:
                addr[ix - 1] = ((a - rng.randrange(0, 1 << 14)) >> 2) & PMASK
        elif typ == 2:
            addr[ix] = (a >> 2) & PMASK
        else:
            lg = rng.choice([3, 4, 6, 10, 12, 16, 20, 31, 40, 55, 56])
            addr[ix] = _napot(a & ~((1 << lg) - 1), lg) if rng.random() < 0.8 else rng.getrandbits(54)
    for ix in range(64):                      # sprinkle addresses under Off entries too
        if not cfg[ix] and rng.random() < 0.1:
            addr[ix] = rng.getrandbits(54)
    return cfg, addr


def _load(h, cfg, addr):
    for ix in range(64):                      # addresses first: no entry is on (or locked) yet
        h.set_csr(0x3B0 + ix, addr[ix])
    for k in range(8):
        v = 0
        for j in range(8):
            v |= cfg[8 * k + j] << (8 * j)
        h.set_csr(0x3A0 + 2 * k, v)


def test_region_matching_matches_reference_pmp_manager():
    rng = random.Random(0x504D50)
    for trial in range(300):
        cfg, addr = _random_config(rng)
        h = rvoracle.Hart(isa="rv64imacsu_zicsr")
        _load(h, cfg, addr)
        c8 = (C.c_uint64 * 8)(*[h.csr(0x3A0 + 2 * k) for k in range(8)])
        a64 = (C.c_uint64 * 64)(*[h.csr(0x3B0 + i) for i in range(64)])
        m = R.ref_pmp_new(c8, a64)
        try:
            probes = [0, 3, 4, BASE - 4, BASE, BASE + (1 << 16) - 1, (1 << 56) - 1, (1 << 64) - 1]
            for ix in range(64):
                if cfg[ix]:
                    lo = (addr[ix] << 2) & ((1 << 64) - 1)
                    probes += [lo - 4, lo - 1, lo, lo + 3, lo + 4, (addr[ix - 1] << 2) if ix else 0]
            probes += [BASE + rng.randrange(-(1 << 12), 1 << 17) for _ in range(40)]
            for a in probes:
                a &= (1 << 64) - 1
                for priv in (0, 1, 3):
                    want = R.ref_pmp_perm(m, priv, a)
                    got = rvoracle.pmp_perm(h, priv, a)
                    assert got == want, f"trial {trial} addr {a:#x} priv {priv}: oracle {got} reference {want}; cfg={[hex(c) for c in cfg if c]}"
        finally:
            R.ref_pmp_free(m)


def test_pmpcfg_legalisation_and_locks_match_reference():
    rng = random.Random(7)
    h = rvoracle.Hart(isa="rv64imacsu_zicsr")
    prev = 0
    for _ in range(400):
        nxt = rng.getrandbits(64)
        if rng.random() < 0.5:
            nxt &= 0x1F1F1F1F1F1F1F1F | (rng.getrandbits(64) & 0x8080808080808080 if rng.random() < 0.3 else 0)
        want = R.ref_pmp_legalize_cfg(prev, nxt) & 0x9F9F9F9F9F9F9F9F
        h.set_csr(0x3A2, nxt)
        got = h.csr(0x3A2)
        assert got == want, f"prev {prev:#x} next {nxt:#x}: oracle {got:#x} reference {want:#x}"
        prev = got
    # locked entry: its pmpaddr and (TOR) the one below are write-protected (CsRegs.cpp:6773-6793)
    h = rvoracle.Hart(isa="rv64imacsu_zicsr")
    h.set_csr(0x3B0, 0x1000); h.set_csr(0x3B1, 0x2000); h.set_csr(0x3B2, 0x3000)
    h.set_csr(0x3A0, (0x80 | (1 << 3) | 7) << 8)          # entry 1: locked TOR rwx
    for n, v in ((0x3B0, 0x1111), (0x3B1, 0x2222), (0x3B2, 0x3333)):
        h.set_csr(n, v)
    assert [h.csr(0x3B0), h.csr(0x3B1), h.csr(0x3B2)] == [0x1000, 0x2000, 0x3333]
