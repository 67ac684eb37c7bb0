// TEST TOOL (never loaded by the whisper_amd package): compiles the device semantics header
// whisper_amd/csrc/wsp_core.h for the host and steps ONE hart serially, so that the decoder,
// softfloat and per-lane step logic that the HIP kernel runs can be diffed against the oracle
// on the CPU-only build machine.  The product path is the HIP kernel in wsp_lib.hip; GPU parity
// tests (pytest -m gpu) go through the C ABI, not through this file.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../whisper_amd/csrc/wsp_core.h"
using namespace wsp;

struct HostMem {
  u64 x[32], f[32], cold_[CI_COUNT];
  u8* ram; u64* tlb_; u8* vreg_;
  u64 rx(u32 r) const { return x[r]; }
  void wx(u32 r, u64 v) { x[r] = v; }
  u64 rf(u32 r) const { return f[r]; }
  void wf(u32 r, u64 v) { f[r] = v; }
  u64& cold(u32 i) { return cold_[i]; }
  u64& tlb(u32 i) { return tlb_[i]; }
  u32 xlane_or(u32 v) const { return v; }            // one lane per hart on the host
  u32 xlane_min(u32 v) const { return v; }
  u64 xlane_xchg64(u64 v, u32) const { return v; }
  void xlane_fence() const {}
  u64 vld(u32 o, u32 esz) const { u64 v = 0; memcpy(&v, vreg_ + o, (size_t)1 << esz); return v; }
  void vst(u32 o, u32 esz, u64 v) { memcpy(vreg_ + o, &v, (size_t)1 << esz); }
  u8 ld8(u64 o) const { return ram[o]; }
  u16 ld16(u64 o) const { u16 v; memcpy(&v, ram + o, 2); return v; }
  u32 ld32(u64 o) const { u32 v; memcpy(&v, ram + o, 4); return v; }
  u64 ld64(u64 o) const { u64 v; memcpy(&v, ram + o, 8); return v; }
  void st8(u64 o, u8 v) { ram[o] = v; }
  void st16(u64 o, u16 v) { memcpy(ram + o, &v, 2); }
  void st32(u64 o, u32 v) { memcpy(ram + o, &v, 4); }
  void st64(u64 o, u64 v) { memcpy(ram + o, &v, 8); }
};

struct Sim { Lane<HostMem> L; std::vector<u8> ram; std::vector<u64> tlb; std::vector<u8> vreg; std::vector<Rec> recs; u64 max_exec; u64 max_ret = ~0ULL; int use_fast = 1; u64 fast_tr_steps = 0; };

template <bool TRACE> static void run_t(Sim* s, u64 max_exec)
{
  Lane<HostMem>& L = s->L;
  L.update_hot_flag();
  Lane<HostMem>::FastTr ft; bool tr_live = false;
  while (L.h.status == ST_RUNNING) {
    if (L.h.exec >= max_exec || L.h.retired >= s->max_ret) { L.h.status = ST_STOP_LIMIT; break; }
    if (L.h.pc == L.stop_pc) { L.h.status = ST_STOP_ADDR; break; }
    // same structure as the GPU kernel: hot harts try the pre-decoded fetch + inline subset first
    Rec rec; rec.w0 = 0; rec.imm = 0; rec.raw = 0; rec.pad = 0;
    if (s->use_fast && (L.h.flags & HF_HOT) && L.hot_fetch(s->recs.data(), rec)) {
      const u64 cur_pc = L.h.pc;
      const u64 a = L.m.rx(rec_rs1(rec));
      const u64 b = (rec.w0 & F_IMM) ? (u64)(i64)rec.imm : L.m.rx(rec_rs2(rec));
      L.h.pc = cur_pc + ((rec.w0 & F_SIZE4) ? 4 : 2);
      if (L.exec_fast<TRACE>(rec.w0 & 0xff, rec, a, b, cur_pc)) {
        L.h.exec++; L.h.cycle++; L.h.instret++; L.h.retired++; L.trace_fast<TRACE>(rec, cur_pc); continue;
      }
      L.h.pc = cur_pc;        // not in the inline subset: nothing was changed, take the general step
    }
    // S / U-mode code under Sv39: the TLB-hit fast step (the kernel's wsp_swarm), its memo kept while fast steps follow one another
    if (s->use_fast && !s->recs.empty() && L.fast_tr_allowed() && !(L.h.pc == L.stop_pc)) {
      if (!tr_live) { L.fast_tr_begin(ft); tr_live = true; }
      if (L.fast_step_translated<TRACE>(s->recs.data(), ft)) { L.h.exec++; L.h.cycle++; L.h.instret++; L.h.retired++; ++s->fast_tr_steps; continue; }
    }
    tr_live = false;
    L.cold_step<TRACE>(s->use_fast && !s->recs.empty() ? s->recs.data() : nullptr);
  }
}

extern "C" {
Sim* hs_create3(u32 isa, u64 mem_base, u64 mem_size, u64 tohost, u32 hart_id, u32 tlb_entries, u32 vlenb);
Sim* hs_create2(u32 isa, u64 mem_base, u64 mem_size, u64 tohost, u32 hart_id, u32 tlb_entries) { return hs_create3(isa, mem_base, mem_size, tohost, hart_id, tlb_entries, 0); }
Sim* hs_create3(u32 isa, u64 mem_base, u64 mem_size, u64 tohost, u32 hart_id, u32 tlb_entries, u32 vlenb)
{
  Sim* s = new Sim();
  s->ram.assign(mem_size, 0);
  memset(&s->L, 0, sizeof s->L);
  s->L.m.ram = s->ram.data();
  s->L.tlb_n = (isa & ISA_S) ? (tlb_entries ? tlb_entries : 2048) : 0;
  s->tlb.assign(2 * (size_t)s->L.tlb_n + 2, 0);
  s->L.m.tlb_ = s->tlb.data();
  s->L.vlenb = (isa & ISA_V) ? (vlenb ? vlenb : 64) : 0;
  s->vreg.assign(32 * (size_t)s->L.vlenb + 8, 0);
  s->L.m.vreg_ = s->vreg.data();
  s->L.m.cold_[CI_VTYPE] = 1ULL << 63;
  s->L.mem_base = mem_base; s->L.mem_size = mem_size; s->L.tohost = tohost; s->L.isa = isa; s->L.hart_id = hart_id;
  s->L.h.pc = mem_base; s->L.h.mstatus = (isa & ISA_RV32) ? (3ULL << MST_MPP_SH) : MSTATUS_RESET; s->L.h.priv = PRV_M;
  s->max_exec = ~0ULL; s->L.stop_pc = ~0ULL; s->L.vsub = 0; s->L.vgrp = 1;
  return s;
}
Sim* hs_create(u32 isa, u64 mem_base, u64 mem_size, u64 tohost, u32 hart_id) { return hs_create2(isa, mem_base, mem_size, tohost, hart_id, 0); }
void hs_destroy(Sim* s) { delete s; }
int hs_write_mem(Sim* s, u64 addr, const void* src, u64 n)
{ if (!s->L.in_mem(addr, n)) return -1; memcpy(s->ram.data() + (addr - s->L.mem_base), src, n); return 0; }
int hs_read_mem(Sim* s, u64 addr, void* dst, u64 n)
{ if (!s->L.in_mem(addr, n)) return -1; memcpy(dst, s->ram.data() + (addr - s->L.mem_base), n); return 0; }
// pre-decode [addr, addr+n) exactly as wsp_predecode_kernel does
void hs_predecode(Sim* s, u64 addr, u64 n)
{
  n &= ~1ULL;
  const u8* img = s->ram.data() + (addr - s->L.mem_base);
  s->recs.assign(n / 2, Rec());
  for (u64 off = 0; off + 2 <= n; off += 2) {
    u32 lo = img[off] | ((u32)img[off + 1] << 8); Rec r;
    if ((lo & 3) != 3) r = decode16(lo, s->L.isa);
    else if (off + 4 <= n) r = decode32(lo | ((u32)img[off + 2] << 16) | ((u32)img[off + 3] << 24), s->L.isa);
    else { r = mk(OP_SLOWFETCH, 0, 0, 0, 0); r.raw = lo; }
    s->recs[off / 2] = r;
  }
  s->L.text_lo = addr; s->L.text_hi = addr + n;
}
void hs_use_fast(Sim* s, int f) { s->use_fast = f; }
u64 hs_tlb_word(Sim* s, u32 i) { return i < s->tlb.size() ? s->tlb[i] : 0; }
u64 hs_fast_tr_steps(Sim* s) { return s->fast_tr_steps; }     // steps taken by the TLB-hit fast step (Lane::fast_step_translated)
void hs_set_stop(Sim* s, u64 max_ret, u64 stop_pc) { s->max_ret = max_ret; s->L.stop_pc = stop_pc; }
void hs_set_pc(Sim* s, u64 pc) { s->L.h.pc = pc & ~1ULL; }
void hs_set_x(Sim* s, u32 i, u64 v) { if (i) s->L.m.x[i & 31] = s->L.rv32() ? (u64)(i64)(i32)v : v; }
u64 hs_get_x(Sim* s, u32 i) { return s->L.rv32() ? (u32)s->L.m.x[i & 31] : s->L.m.x[i & 31]; }
u64 hs_get_f(Sim* s, u32 i) { return s->L.m.f[i & 31]; }
void hs_set_f(Sim* s, u32 i, u64 v) { s->L.m.f[i & 31] = v; }
u64 hs_get_pc(Sim* s) { return s->L.h.pc; }
u64 hs_csr(Sim* s, u32 num) { return s->L.csr_peek(num); }
u64 hs_cold_raw(Sim* s, u32 ix) { return ix < CI_COUNT ? s->L.m.cold_[ix] : 0; }
void hs_poke_csr(Sim* s, u32 num, u64 v) { s->L.csr_store(num, v); s->L.s.chg = 0; }
int hs_run(Sim* s, u64 max_exec, int trace)
{
  if (s->L.h.status == ST_STOP_LIMIT || s->L.h.status == ST_STOP_ADDR) s->L.h.status = ST_RUNNING;   // (wsp_batch_resume)
  if (trace) run_t<true>(s, max_exec); else run_t<false>(s, max_exec);
  return (int)s->L.h.status;
}
u64 hs_stop_value(Sim* s) { return s->L.h.stop_value; }
u64 hs_exec(Sim* s) { return s->L.h.exec; }
u64 hs_retired(Sim* s) { return s->L.h.retired; }
u64 hs_thash(Sim* s) { return s->L.h.thash; }
u32 hs_priv(Sim* s) { return s->L.h.priv; }
u64 hs_state_hash(Sim* s)
{
  Lane<HostMem>& L = s->L; u64 h = 0;
  h = fold(h, L.h.pc);
  for (int i = 1; i < 32; ++i) h = fold(h, L.m.x[i]);
  for (int i = 0; i < 32; ++i) h = fold(h, L.m.f[i]);
  h = fold(h, L.h.priv); h = fold(h, L.h.fcsr);
  const int c1[] = { CI_STVEC, CI_SCOUNTEREN, CI_SENVCFG, CI_SSCRATCH, CI_SEPC, CI_SCAUSE, CI_STVAL, CI_SATP };
  for (int c : c1) h = fold(h, L.m.cold_[c]);
  h = fold(h, L.h.mstatus);
  const int c2[] = { CI_MEDELEG, CI_MIDELEG, CI_MIE, CI_MTVEC, CI_MCOUNTEREN, CI_MENVCFG, CI_MCOUNTINHIBIT, CI_MSCRATCH, CI_MEPC, CI_MCAUSE, CI_MTVAL, CI_MIP };
  for (int c : c2) h = fold(h, L.m.cold_[c]);
  h = fold(h, L.h.cycle); h = fold(h, L.h.instret); h = fold(h, L.h.exec);
  for (int i = 0; i < 72; ++i) h = fold(h, L.m.cold_[CI_PMPCFG0 + i]);
  if (L.isa & ISA_S) { u64 t = 0; for (u32 i = 0; i < 2 * L.tlb_n; ++i) t = fold(t, s->tlb[i]); h = fold(h, t); }
  if (L.isa & ISA_V) {
    h = fold(h, L.m.cold_[CI_VSTART]); h = fold(h, L.m.cold_[CI_VCSR]); h = fold(h, L.m.cold_[CI_VL]); h = fold(h, L.m.cold_[CI_VTYPE]);
    for (u32 i = 0; i < 32 * L.vlenb / 8; ++i) { u64 w; memcpy(&w, s->vreg.data() + 8 * i, 8); h = fold(h, w); }
  }
  return h;
}
void hs_get_vreg(Sim* s, u32 i, void* dst) { memcpy(dst, s->vreg.data() + (size_t)(i & 31) * s->L.vlenb, s->L.vlenb); }
u64 hs_mem_hash(Sim* s)
{
  u64 sum = 0, nw = s->L.mem_size / 8;
  for (u64 i = 0; i < nw; ++i) { u64 w; memcpy(&w, s->ram.data() + i * 8, 8); sum += mix(w ^ (GOLDEN * (i + 1))); }
  return sum;
}
void hs_decode(u32 isa, const u32* words, u64 n, Rec* out) { for (u64 i = 0; i < n; ++i) out[i] = decode(words[i], isa); }
}
