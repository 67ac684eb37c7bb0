"""wsp_batch_results_noalloc (the entry point that allocates nothing: digest partial sums in the library's static device array,
records written into the caller's device buffer) against wsp_batch_results, bit-exact, on two calls in a row - the second on a
smaller batch, so stale partial sums left in the shared array by the first call would show."""
import ctypes as C

import pytest

import whisper_amd as W
from whisper_amd import workloads as WL

pytestmark = pytest.mark.gpu


class _Hip:
    """The caller's side of the test: its own device buffer, through the HIP runtime the library already loaded (not torch: a
    second HIP runtime initialised late in a process that has been running kernels for a while does not find the GPU)."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def malloc(self, n):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), n) == 0
        return p.value

    def fill(self, p, byte, n):
        assert self.rt.hipMemset(p, byte, n) == 0 and self.rt.hipDeviceSynchronize() == 0

    def read(self, p, n):
        buf = C.create_string_buffer(n)
        assert self.rt.hipMemcpy(buf, p, n, 2) == 0          # hipMemcpyDeviceToHost
        return buf.raw

    def free(self, p):
        self.rt.hipFree(p)


def _batch(n, lanes):
    p = WL.compute_kernel(iters=1)
    b = W.Batch(n, isa=p.isa, mem_base=p.mem_base, mem_size=p.mem_size, lanes_per_wave=lanes)
    b.load_elf_bytes(p.elf)
    b.set_stop(max_inst=20000 + 7 * n)
    b.poke_each(W.RES_X, 10, [WL.compute_seed(i) for i in range(n)])
    b.poke_each(W.RES_MEM, p.symbols["seed"], list(range(n)))
    assert b.run(trace_hash=True) == 0
    return b


def test_results_noalloc_matches_results_twice():
    big = _batch(300, 16)
    small = _batch(70, 16)
    hip = _Hip()
    rs = C.sizeof(W.Result)
    dev = hip.malloc(300 * rs)                               # caller-owned output, allocated before the calls
    try:
        for b in (big, small):
            want = bytes(b.results())
            hip.fill(dev, 0xA5, 300 * rs)
            b.results_noalloc(dev)
            raw = hip.read(dev, 300 * rs)
            got = raw[: b.n * rs]
            assert got == want, f"{b.n} harts: records differ"
            assert raw[b.n * rs:] == b"\xA5" * ((300 - b.n) * rs)
        recs = (W.Result * small.n).from_buffer_copy(got)
        assert all(r.status == W.ST_STOP_LIMIT and r.mem_hash for r in recs)
    finally:
        hip.free(dev); big.close(); small.close()


def test_results_noalloc_rejects_oversized_batch():
    b = W.Batch(16385, mem_size=4096, lanes_per_wave=1)          # 16385 waves x 64 slots > the 1048576-slot scratch array
    try:
        hip = _Hip()
        dev = hip.malloc(64)
        with pytest.raises(W.WspError) as e:
            b.results_noalloc(dev)
        assert e.value.code == W.ERANGE
        assert len(b.results()) == 16385                          # the allocating entry point still serves it
    finally:
        hip.free(dev); b.close()
