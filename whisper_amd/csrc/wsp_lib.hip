// whisper_amd: HIP kernels (gfx950) + host front end + C ABI (include/whisper_amd.h).
//
// Execution model: one lane = one hart, one 64-lane wavefront per workgroup steps up
// to 64 harts in lock step.  Integer and FP register files are staged in LDS as [lane][33 slots]
// (stride 33 dwords-pairs: converged register numbers fall into distinct banks), a handful of
// per-hart values stay in VGPRs for the whole slice, cold CSRs / RAM arenas / pre-decoded
// programs live in HBM.  Dispatch picks the op of the first pending lane, makes it wave-uniform
// with readlane and runs that op for every lane that fetched the same op; converged harts cost
// one pass per step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include <mutex>
// -DWSP_PROF: cycle accounting of the step kernel's paths (tuning builds only; read back with wsp_debug_prof)
#ifdef WSP_PROF
__device__ unsigned long long wsp_prof[16];
#define WSP_PROF_ADD(k, v) do { if ((int)threadIdx.x == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&wsp_prof[k], (unsigned long long)(v)); } while (0)
#define WSP_PROF_T() ((unsigned long long)clock64())
#else
#define WSP_PROF_ADD(k, v) do { } while (0)
#define WSP_PROF_T() 0ULL
#endif
#include "wsp_core.h"
#include "wsp_conv_core.h"
#include "wsp_config.h"
#include "wsp_disasm.h"
#include "../../include/whisper_amd.h"

using namespace wsp;

// ------------------------------------------------------------------------------------------ layout
enum { H64_PC, H64_MSTATUS, H64_CYCLE, H64_INSTRET, H64_EXEC, H64_RETIRED, H64_THASH, H64_LR_ADDR, H64_EXEC_LAST_ILL, H64_STOP_VALUE, H64_COUNT };
enum { H32_FCSR, H32_PRIV, H32_STATUS, H32_CONSEC_ILL, H32_HTIF_DEV, H32_FLAGS, H32_COUNT };

struct ProgDev { u64 text_lo, text_hi; const Rec* recs; const HRec* hrecs; };   // hrecs: hot records of the hand-written loop, or null

struct DevState {
  u32 n_harts, n_waves, lpw, lpw_shift, S, isa, hart_id_base, tlb_n;
  u32 mtab_off, pad0;  // byte offset in the launch's LDS of the lane table (ln_of): behind everything else, set per launch (launch_step)
  u64 mem_base, mem_size;
  u64* x; u64* f; u64* hot64; u32* hot32; u64* cold;
  u64* tlb;            // [n_harts][tlb_n][2] packed TLB entries (wsp_core.h); null when the ISA has no S
  u8* vreg;            // [n_harts][32][vlenb] vector registers, element 0 first; null when the ISA has no V
  u32 vlenb, ring_cap;
  u64* ring; u32* ring_pos;   // commit ring: [n_harts][ring_cap] words and the words written per hart; null when off
  u64* tohost; u64* max_exec; u32* prog;
  u64* max_ret; u64* stop_pc;   // retired-instruction limit and stop address per slot (Hart.cpp:6025-6026); ~0 = none
  u8* arena;
  const ProgDev* progs;
  u32* running;
};

__host__ __device__ static inline u32 slot_of(const DevState& d, u32 hart) { return (hart >> d.lpw_shift) * 64 + (hart & (d.lpw - 1)); }
__host__ __device__ static inline size_t arena_off(const DevState& d, u32 hart, u64 off)
{
  u32 w = hart >> d.lpw_shift, l = hart & (d.lpw - 1);
  return (size_t)w * d.lpw * d.mem_size + ((((size_t)(off >> 3) << d.lpw_shift) + l) << 3) + (off & 7);
}

// ------------------------------------------------------------------------------------------ device memory policy
// LDS register block of one lane: WSP_XSLOTS u64 slots.  x1..x31 sit in slots 1..31; slot 0 is a write sink (the hot loop
// writes rd = x0 results there unconditionally), slot 32 is the constant zero that every read of x0 is redirected to
// (pre-decoded operand offsets, Rec.pad).  The f block uses slots 0..31.
#define WSP_XSLOTS 33
extern __shared__ __align__(16) u64 wsp_lds[];
// The vector register files of a wave's harts live in LDS while the step kernel runs (behind the x and f blocks; copied in from
// DevState::vreg at kernel start and back at the end): an element access of the vector unit's loops costs an LDS round trip
// instead of a dependent trip to HBM / L2.  Addressed as an LDS offset so that the accesses compile to ds_read / ds_write.
__device__ __forceinline__ u8* wsp_vreg_lds(u32 off) { return (u8*)wsp_lds + off; }
struct DevMem {
  u64* xs; u64* fs; u64* cold_; u64* tlb_; u8* arena; u32 S, lpw_shift;
  u64* vcsr_;          // vstart, vcsr, vl, vtype of this hart: in LDS while the step kernel runs (null when the ISA has no V)
  u32 voff;            // byte offset in LDS of this hart's vector register file ([32][vlenb], element 0 first)
  __device__ __forceinline__ u64 rx(u32 r) const { return xs[r ? r : 32]; }
  __device__ __forceinline__ void wx(u32 r, u64 v) { xs[r] = v; }
  __device__ __forceinline__ u64 rf(u32 r) const { return fs[r]; }
  __device__ __forceinline__ void wf(u32 r, u64 v) { fs[r] = v; }
#ifndef WSP_VCSR_LDS
#define WSP_VCSR_LDS 1
#endif
  __device__ __forceinline__ u64& cold(u32 i) { return (WSP_VCSR_LDS && i >= CI_VSTART && i <= CI_VTYPE) ? vcsr_[i - CI_VSTART] : cold_[(size_t)i * S]; }
  __device__ __forceinline__ u64& tlb(u32 i) { return tlb_[i]; }
  // element of 1 << esz bytes at byte offset o (a multiple of the size) of this hart's vector register file
  __device__ __forceinline__ u64 vld(u32 o, u32 esz) const
  { const u8* q = wsp_vreg_lds(voff + o); return esz == 0 ? *q : esz == 1 ? *(const u16*)q : esz == 2 ? *(const u32*)q : *(const u64*)q; }
  __device__ __forceinline__ void vst(u32 o, u32 esz, u64 v)
  { u8* q = wsp_vreg_lds(voff + o); if (esz == 0) *q = (u8)v; else if (esz == 1) *(u16*)q = (u16)v; else if (esz == 2) *(u32*)q = (u32)v; else *(u64*)q = v; }
  // Sub-group of a hart in a V batch (Lane::vgrp lanes: this hart's lane l and the lanes l + lpw, l + 2 lpw, ... of the wave, all
  // executing the same instruction): combine a value over it.  Only ever called with the whole sub-group active.
  __device__ __forceinline__ u32 xlane_or(u32 v) const
  { for (u32 o = 1u << lpw_shift; o < 64; o <<= 1) v |= (u32)__shfl_xor((int)v, (int)o, 64); return v; }
  __device__ __forceinline__ u32 xlane_min(u32 v) const
  { for (u32 o = 1u << lpw_shift; o < 64; o <<= 1) { const u32 t = (u32)__shfl_xor((int)v, (int)o, 64); v = t < v ? t : v; } return v; }
  // the value of the sub-group lane whose index differs in bit `stage` (a butterfly stage of a reduction over the sub-group)
  __device__ __forceinline__ u64 xlane_xchg64(u64 v, u32 stage) const
  {
    const int o = (int)((1u << lpw_shift) << stage);
    return (u64)(u32)__shfl_xor((int)(u32)v, o, 64) | ((u64)(u32)__shfl_xor((int)(u32)(v >> 32), o, 64) << 32);
  }
  __device__ __forceinline__ void xlane_fence() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
  __device__ __forceinline__ u8* p(u64 off) const { return arena + (((off >> 3) << lpw_shift) << 3) + (off & 7); }
  __device__ __forceinline__ u8 ld8(u64 o) const { return *p(o); }
  __device__ __forceinline__ u16 ld16(u64 o) const { return *(const u16*)p(o); }
  __device__ __forceinline__ u32 ld32(u64 o) const { return *(const u32*)p(o); }
  __device__ __forceinline__ u64 ld64(u64 o) const { return *(const u64*)p(o); }
  __device__ __forceinline__ void st8(u64 o, u8 v) { *p(o) = v; }
  __device__ __forceinline__ void st16(u64 o, u16 v) { *(u16*)p(o) = v; }
  __device__ __forceinline__ void st32(u64 o, u32 v) { *(u32*)p(o) = v; }
  __device__ __forceinline__ void st64(u64 o, u64 v) { *(u64*)p(o) = v; }
};

// ------------------------------------------------------------------------------------------ step kernel
// Occupancy target: ONE wave per SIMD, i.e. the whole register file (256 VGPRs + 256 AGPRs) for each wave.  Round 1 ran one hart
// per wave and needed 4 waves per SIMD to fill the machine (128 VGPRs; measured then: 2 -> 4.8, 4 -> 8.1, 8 -> 6.3 GIPS at 4096
// harts).  Since the converged loop a wave carries up to 64 harts and one wave per SIMD already saturates the hot loop (196 GIPS
// at 65536 harts = 1024 waves; batches of more waves simply run in turns), while under the 128-register cap the general path
// kept the hart's `Lane` in scratch memory - thousands of waves' scratch lives in HBM - and paid ~51 k cycles per general step
// (-DWSP_PROF).  Measured (profiles/r02/ab_waves_per_eu.jsonl, 4 -> 1): M/S/U fuzz 2.0 -> 6.6 GIPS, Sv39 fuzz 1.2 -> 2.8,
// random RVV programs 0.59 -> 1.06, compute kernel 12.8 -> 13.0 at 4096 harts and 183 -> 197 at 262144.
#ifndef WSP_WAVES_PER_EU
#define WSP_WAVES_PER_EU 1
#endif
#ifndef WSP_CONVERGED_LOOP
#define WSP_CONVERGED_LOOP 1
#endif
#ifndef WSP_CONV_ASM
#define WSP_CONV_ASM 1                                   // 0: untraced converged waves run the C++ loop too (A/B measurements)
#endif

// Mirror lanes.  Measured on gfx950 (tools/lab/valu_rate.hip): a wave that has its SIMD to itself issues a vector ALU instruction
// every ~1.9 ns with 15 or more lanes enabled in exec, but only every ~8.6 ns with 4 or fewer - and small batches put 1 .. 8 harts
// in a wave, while in a wave of many different programs the few harts that run long (a fuzz program caught in a trap loop until its
// instruction limit) end up alone: tools/exp_replicas.py measured 2.4 us per instruction for such a hart with 15 copies beside it
// against ~10 us alone.  So in batches without V (whose idle lanes serve as vector sub-groups) the lanes that have no running hart
// of their own at the start of a launch run COPIES of the wave's running harts: same state loaded, same instructions executed in
// lock step, the same values stored to the same addresses at the same time (LDS register files, arena, HBM state rows, TLB, commit
// ring: every store of a copy is the owner's store again; nothing on these paths is an atomic), results of the copies ignored.
// The assignment is made once per launch (wsp_step_kernel) and kept in LDS: wsp_mtab[lane] = the lane whose hart this lane works
// on | 0x100 for a copy | 0x200 for a lane with nothing to do.  Waves whose running harts share one program get copies up to 16
// enabled lanes (they execute together anyway); waves of different programs use every idle lane, dealt out in turn.  Batches in
// which waves mix programs run in short slices (wsp_batch_run), so that the lanes of harts that have finished are dealt out again.
#define WSP_MIRROR_LANES 16u
// (the table sits behind the register files in the launch's dynamic LDS, not in a static array: the hand-written loop addresses the
// register files by LDS offsets that take wsp_lds to be at LDS address 0)
__device__ __forceinline__ u32* wsp_mtab_of(const DevState& d) { return (u32*)((char*)wsp_lds + d.mtab_off); }
__device__ __forceinline__ u32 ln_of(const DevState& d) { return wsp_mtab_of(d)[threadIdx.x] & 0xffu; }

template <class LaneT> __device__ __forceinline__ void wsp_lane_setup(LaneT& L, const DevState& d, u32 wave, u32 ln, u32 slot, const Rec*& recs)
{
  L.m.xs = wsp_lds + ln * WSP_XSLOTS;
  L.m.fs = wsp_lds + (d.lpw + ln) * WSP_XSLOTS;
  L.m.cold_ = d.cold + slot;
  L.m.tlb_ = d.tlb + (size_t)(wave * d.lpw + ln) * 2 * d.tlb_n; L.tlb_n = d.tlb_n;
  L.m.voff = d.lpw * (2 * WSP_XSLOTS * 8) + ln * 32 * d.vlenb; L.vlenb = d.vlenb;
  L.m.vcsr_ = wsp_lds + d.lpw * (2 * WSP_XSLOTS + 4 * d.vlenb) + ln * 4;
  if (d.ring) { L.ring_ = d.ring + (size_t)(wave * d.lpw + ln) * d.ring_cap; L.ring_pos_ = d.ring_pos + wave * d.lpw + ln; L.ring_cap = d.ring_cap; }
  L.m.arena = d.arena + (size_t)wave * d.lpw * d.mem_size + (size_t)ln * 8;
  L.m.S = d.S; L.m.lpw_shift = d.lpw_shift;
  L.mem_base = d.mem_base; L.mem_size = d.mem_size; L.isa = d.isa; L.hart_id = d.hart_id_base + wave * d.lpw + ln;
  L.tohost = d.tohost[slot]; L.stop_pc = d.stop_pc[slot];
  const u32 pg = d.prog[slot];
  recs = nullptr; L.text_lo = 0; L.text_hi = 0;
  if (pg != ~0u) { ProgDev p = d.progs[pg]; recs = p.recs; L.text_lo = p.text_lo; L.text_hi = p.text_hi; }
}

__device__ __forceinline__ void wsp_load_hot(Lane<DevMem>& L, const DevState& d, u32 slot)
{
  const u32 S = d.S;
  L.h.pc = d.hot64[H64_PC * S + slot]; L.h.mstatus = d.hot64[H64_MSTATUS * S + slot]; L.h.cycle = d.hot64[H64_CYCLE * S + slot];
  L.h.instret = d.hot64[H64_INSTRET * S + slot]; L.h.exec = d.hot64[H64_EXEC * S + slot]; L.h.retired = d.hot64[H64_RETIRED * S + slot];
  L.h.thash = d.hot64[H64_THASH * S + slot]; L.h.lr_addr = d.hot64[H64_LR_ADDR * S + slot];
  L.h.exec_last_ill = d.hot64[H64_EXEC_LAST_ILL * S + slot]; L.h.stop_value = d.hot64[H64_STOP_VALUE * S + slot];
  L.h.fcsr = d.hot32[H32_FCSR * S + slot]; L.h.priv = d.hot32[H32_PRIV * S + slot]; L.h.status = d.hot32[H32_STATUS * S + slot];
  L.h.consec_ill = d.hot32[H32_CONSEC_ILL * S + slot]; L.h.htif_dev = d.hot32[H32_HTIF_DEV * S + slot]; L.h.flags = d.hot32[H32_FLAGS * S + slot];
}

__device__ __forceinline__ void wsp_store_hot(const Lane<DevMem>& L, const DevState& d, u32 slot)
{
  const u32 S = d.S;
  d.hot64[H64_PC * S + slot] = L.h.pc; d.hot64[H64_MSTATUS * S + slot] = L.h.mstatus; d.hot64[H64_CYCLE * S + slot] = L.h.cycle;
  d.hot64[H64_INSTRET * S + slot] = L.h.instret; d.hot64[H64_EXEC * S + slot] = L.h.exec; d.hot64[H64_RETIRED * S + slot] = L.h.retired;
  d.hot64[H64_THASH * S + slot] = L.h.thash; d.hot64[H64_LR_ADDR * S + slot] = L.h.lr_addr;
  d.hot64[H64_EXEC_LAST_ILL * S + slot] = L.h.exec_last_ill; d.hot64[H64_STOP_VALUE * S + slot] = L.h.stop_value;
  d.hot32[H32_FCSR * S + slot] = L.h.fcsr; d.hot32[H32_PRIV * S + slot] = L.h.priv; d.hot32[H32_STATUS * S + slot] = L.h.status;
  d.hot32[H32_CONSEC_ILL * S + slot] = L.h.consec_ill; d.hot32[H32_HTIF_DEV * S + slot] = L.h.htif_dev; d.hot32[H32_FLAGS * S + slot] = L.h.flags;
}

// What the hot loop keeps in registers per hart.  Everything else of wsp::Hot stays in HBM between
// cold steps; exec/cycle/instret/retired are synchronised lazily: a hot step advances all four by
// exactly one (HF_HOT guarantees no inhibit, no trap), so the HBM copies are brought up to date
// from the difference between the register exec counter and the stored one.
struct HotRegs { u64 pc, exec, thash, stop_at, slice_end; u32 flags, status, fcsr; };
// In HotRegs.flags on the way INTO wsp_cold only (never stored): the translated warm path (wsp_swarm) has just made progress with
// this hart, so the general path executes the one instruction that stopped it and hands the hart back, instead of keeping it for
// a whole burst.
static constexpr u32 HR_SWARM_LIVE = 1u << 31;

// First value of the executed-instruction counter at which the hart must not start another instruction: the end of this launch's
// slice, the instruction-count limit, or the retired-instruction limit, whichever comes first (Hart.cpp:6025-6026).  retired lags
// exec by the number of trapped instructions, which only the general path can change: it calls this again after every step.
__device__ __forceinline__ u64 wsp_stop_at(u64 slice_end, u64 max_exec, u64 max_ret, u64 exec, u64 retired)
{
  u64 r = max_ret + (exec - retired);
  if (r < max_ret) r = ~0ULL;
  const u64 a = slice_end < max_exec ? slice_end : max_exec;
  return a < r ? a : r;
}

// The general path, out of line: rebuilds the full lane from HBM, runs one complete general step
// (interrupt poll, fetch, full exec(), retire, commit record) and stores the lane back.
// A lane that needed one general step usually needs the next one too (S / U-mode code, self-modified code, a trap handler,
// vector code), and what a general step costs is mostly the round trips that bring the lane's state in from HBM and take it
// back.  So the lane stays here, state in registers, until it is hot again at a pre-decoded pc, stops, has used its budget,
// or has run WSP_COLD_BURST steps (the other lanes of the wave wait meanwhile).
#ifndef WSP_COLD_BURST
#define WSP_COLD_BURST 64u
#endif
#ifndef WSP_HOT_QUANTUM
#define WSP_HOT_QUANTUM 128u                             // hot-loop iterations per round while other lanes of the wave wait (step kernel)
#endif
// One compiled body serves traced and untraced runs (HF_NOTRACE in the lane's flags makes commit() return at once): what an
// untraced run executes here is exactly what the traced parity tests execute.  The arguments fit the 16 argument registers (2 for
// dp + 13 for HotRegs), so HotRegs travels in registers; wave and lane are recomputed here.
#ifdef WSP_INSPECT   // tuning aid (tools/inspect_step.sh): the step kernels alone, the out-of-line paths as external calls
__device__ HotRegs wsp_cold(const DevState* dp, HotRegs hr);
#else
__device__ __noinline__ HotRegs wsp_cold(const DevState* dp, HotRegs hr)
{
  const DevState& d = *dp;
  const unsigned long long pt0 = WSP_PROF_T();
  const u32 wave = blockIdx.x, ln = ln_of(d);
  const u32 slot = wave * 64 + ln;
  Lane<DevMem> L; const Rec* recs;
  wsp_lane_setup(L, d, wave, ln, slot, recs);
  wsp_load_hot(L, d, slot);
  const u64 dlt = hr.exec - L.h.exec;              // hot steps since the last sync: each advanced all four counters by one
  L.h.cycle += dlt; L.h.instret += dlt; L.h.retired += dlt;
  L.h.exec = hr.exec; L.h.pc = hr.pc; L.h.thash = hr.thash; L.h.flags = hr.flags & ~HR_SWARM_LIVE; L.h.status = hr.status; L.h.fcsr = hr.fcsr;
  const u32 burst = (hr.flags & HR_SWARM_LIVE) ? 1u : WSP_COLD_BURST;
  const u64 max_exec = d.max_exec[slot], max_ret = d.max_ret[slot];
  u64 stop_at = hr.stop_at;
  u32 k = 0;
  const unsigned long long pt1 = WSP_PROF_T();
  do {
    L.s.chg = 0; L.s.ireg = 0; L.s.freg = 0; L.s.gen_csr = 0; L.s.vreg = 0; L.s.vgroup = 0; L.s.ival = 0; L.s.fval = 0; L.s.maddr = 0; L.s.mval = 0; L.s.lva = 0; L.s.lpa = 0;
    L.template cold_step<true>(recs);
    stop_at = wsp_stop_at(hr.slice_end, max_exec, max_ret, L.h.exec, L.h.retired);
  } while (++k < burst && L.h.status == ST_RUNNING && L.h.exec < stop_at &&
           !((L.h.flags & HF_HOT) && L.h.pc - L.text_lo < L.text_hi - L.text_lo));
  WSP_PROF_ADD(3, WSP_PROF_T() - pt1); WSP_PROF_ADD(6, k);
  wsp_store_hot(L, d, slot);
  HotRegs r; r.pc = L.h.pc; r.exec = L.h.exec; r.thash = L.h.thash; r.stop_at = stop_at; r.slice_end = hr.slice_end; r.flags = L.h.flags; r.status = L.h.status; r.fcsr = L.h.fcsr;
  WSP_PROF_ADD(2, WSP_PROF_T() - pt0); WSP_PROF_ADD(1, 1);
  return r;
}
#endif

// The "warm" path: one F/D computational instruction (arithmetic, convert, compare, move, classify) of a hart that is
// hot and whose mstatus.FS is already Dirty.  Such an instruction reads and writes only the LDS register files and
// fflags, cannot trap once its rounding mode is known to be legal, and advances all counters by one like any hot step:
// no HBM state is touched, and the function is small enough to need no callee-saved registers.  handled = 0 sends the
// instruction to the general path (reserved rounding mode: illegal-instruction trap).
// CSR accesses the warm path takes for a hot (M-mode) hart: registers that are plain storage behind their masks - a write has no
// effect on anything but the register - mstatus / sstatus (the path loads the hart's mstatus for these, stores it back and
// recomputes the hot / FS / VS flags), and reads of whatever else does not depend on state the warm path does not hold (the
// counters: synchronised lazily; mhartid; the vector CSRs: staged in LDS).  fflags / frm / fcsr only with mstatus.FS already Dirty
// (else the access changes mstatus).  Everything else - satp, mie / mip, the PMP file, mcountinhibit ... - is the general path's.
__device__ __forceinline__ bool wsp_csr_warm_ok(u32 num, bool write, bool fs_dirty)
{
  if (num >= C_FFLAGS && num <= C_FCSR) return fs_dirty;
  switch (num) {
    case C_MSCRATCH: case C_MEPC: case C_MCAUSE: case C_MTVAL: case C_SSCRATCH: case C_SEPC: case C_SCAUSE: case C_STVAL: return true;
    case C_MSTATUS: case C_SSTATUS: return true;
    case C_MCYCLE: case C_MINSTRET: case C_CYCLE: case C_TIME: case C_INSTRET: case C_MHARTID: return false;
    default: break;
  }
  if ((num >= 0x008 && num <= 0x00F) || (num >= 0xC20 && num <= 0xC22) || (num >= 0xB80 && num <= 0xB9F) || (num >= 0xC80 && num <= 0xC9F)) return false;
  return !write;
}
__device__ __forceinline__ u32 wsp_uni32(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 wsp_uni64(u64 v) { return (u64)wsp_uni32((u32)v) | ((u64)wsp_uni32((u32)(v >> 32)) << 32); }
struct WarmRet { u64 thash, pc; u32 fcsr, handled, flags; };
#ifdef WSP_INSPECT
template <bool TRACE> __device__ WarmRet wsp_warm(const DevState* dp, u32 w0, i32 imm, u32 raw, u64 pc, u64 thash, u32 fcsr, u32 flags);
#else
template <bool TRACE>
__device__ __noinline__ WarmRet wsp_warm(const DevState* dp, u32 w0, i32 imm, u32 raw, u64 pc, u64 thash, u32 fcsr, u32 flags)
{
  const DevState& d = *dp;
  // Every lane brings its own instruction: a wave of 64 different fuzz programs stands at up to 64 different csrr / csrw / mret /
  // mul words at once, and one call per distinct word (tried in round 3 for the scalar dispatch it allows) made the warm path 26 %
  // of that workload's wave cycles (-DWSP_PROF); the paths through this function are few, the words many.
  const u32 ln = ln_of(d), slot = blockIdx.x * 64 + ln, isa = wsp_uni32(d.isa);
  const size_t S = wsp_uni32(d.S);
  Lane<DevMem> L;
  L.m.xs = wsp_lds + ln * WSP_XSLOTS; L.m.fs = wsp_lds + (wsp_uni32(d.lpw) + ln) * WSP_XSLOTS;
  L.m.cold_ = d.cold + slot; L.m.S = d.S; L.m.vcsr_ = nullptr; L.m.tlb_ = nullptr; L.tlb_n = 0; L.m.voff = 0; L.vlenb = 0; L.m.arena = nullptr; L.m.lpw_shift = 0;
  L.isa = isa; L.mem_base = 0; L.mem_size = 0; L.tohost = ~0ULL; L.text_lo = 0; L.text_hi = 0; L.hart_id = 0;
  L.h.pc = pc; L.h.mstatus = MSTATUS_RESET | (3ULL << MST_FS_SH) | MST_SD; L.h.fcsr = fcsr; L.h.thash = thash;
  L.h.flags = flags; L.h.status = ST_RUNNING; L.h.priv = PRV_M; L.h.exec = 0; L.h.cycle = 0; L.h.instret = 0; L.h.retired = 0;
  L.s.chg = 0; L.s.ireg = 0; L.s.freg = 0; L.s.gen_csr = 0; L.s.vreg = 0; L.s.vgroup = 0; L.s.ival = 0; L.s.fval = 0; L.s.maddr = 0; L.s.mval = 0; L.s.lva = 0; L.s.lpa = 0;
  Rec rec; rec.w0 = w0; rec.imm = imm; rec.pad = 0; rec.raw = raw;
  const u32 op = w0 & 0xff, size = (w0 & F_SIZE4) ? 4 : 2;
  WarmRet r; r.thash = thash; r.fcsr = fcsr; r.handled = 0; r.pc = pc + size; r.flags = flags;
  // Zicsr and mret work on the hart's own mstatus (one HBM word in, and out again if it changed); the other cases never look at
  // more of it than FS, which is Dirty whenever they are called
  const bool sys = (op >= OP_CSRRW && op <= OP_CSRRC) || op == OP_MRET;
  u64 mst0 = 0;
  if (sys) { mst0 = d.hot64[(size_t)H64_MSTATUS * S + slot]; L.h.mstatus = mst0; }
  if (op == OP_ZB) {                                         // bit manipulation: registers only, cannot trap
    const u64 b = (w0 & F_IMM) ? (u64)(i64)imm : L.m.rx(rec_rs2(rec));
    L.wx(rec_rd(rec), L.exec_zb(rec_rs3(rec), (w0 & F_IMM) ? 0 : imm, L.m.rx(rec_rs1(rec)), b));
  } else if (op >= OP_MUL && op <= OP_REMUW) {               // the M-extension forms the hot loops leave out (mulh*, div*, rem*)
    L.wx(rec_rd(rec), L.muldiv(op, L.m.rx(rec_rs1(rec)), L.m.rx(rec_rs2(rec))));
  } else if (op >= OP_CSRRW && op <= OP_CSRRC) {             // Zicsr on plain-storage registers (wsp_csr_warm_ok), M mode
    const u32 num = (u32)imm & 0xfff;
    const bool write = op == OP_CSRRW || rec_rs1(rec) != 0;
    if (!wsp_csr_warm_ok(num, write, (flags & HF_FS_DIRTY) != 0) || L.csr_class(num) != 1) return r;
    L.exec_csr(rec, op - OP_CSRRW, (w0 & F_IMM) ? (u64)rec_rs1(rec) : (L.m.rx(rec_rs1(rec)) & L.amask()), pc);
  } else if (op == OP_MRET) {                                // mret from M mode (Hart.cpp:12247-12336): mstatus, privilege, pc = mepc
    u64 v = L.h.mstatus; const u32 saved = (u32)(v >> MST_MPP_SH) & 3;
    v = (v & ~MST_MIE) | ((v & MST_MPIE) ? MST_MIE : 0);
    const u32 least = L.ext(ISA_U) ? PRV_U : L.ext(ISA_S) ? PRV_S : PRV_M;
    v = (v & ~(3ULL << MST_MPP_SH)) | ((u64)least << MST_MPP_SH);
    v |= MST_MPIE;
    if (saved != PRV_M && !L.ext(CFG_KEEP_MPRV)) v &= ~MST_MPRV;
    L.csr_store(C_MSTATUS, v); L.s.chg |= CH_MSTATUS;
    r.pc = L.m.cold(CI_MEPC) & L.epc_rmask(); L.h.priv = saved;
    d.hot32[(size_t)H32_PRIV * S + slot] = saved;
  } else {
    const bool uses_rm = (op >= OP_FADD && op <= OP_FMA) || op == OP_FCVT_F2I || op == OP_FCVT_I2F || op == OP_FCVT_FF;
    int rm = 0;
    if (uses_rm && !L.fp_rm((u32)imm & 7, rm)) return r;
    L.exec_fp(rec, L.m.rx(rec_rs1(rec)), pc);
  }
  L.template commit<TRACE>(pc, raw, size, PRV_M, 0);
  if (sys) {
    if (L.h.mstatus != mst0) d.hot64[(size_t)H64_MSTATUS * S + slot] = L.h.mstatus;
    L.update_hot_flag();                                     // privilege / MPRV / FS / VS may have changed (as the general step does)
    r.flags = L.h.flags;
  }
  r.thash = L.h.thash; r.fcsr = L.h.fcsr; r.handled = 1;
  return r;
}
#endif

// The vector warm path: one vector instruction (OP-V, unit-stride / strided / mask load or store) of a hart that is hot and whose
// mstatus.VS is already Dirty, executed by the hart's whole SUB-GROUP - its own lane l and the otherwise idle lanes l + lpw,
// l + 2 lpw, ... of the wave (64 / lpw lanes; V batches run at most 16 harts per wave).  Every lane of the sub-group builds the
// same minimal hart (registers only: no HBM state is loaded or stored, the vector registers are in LDS) and runs the vector unit
// (Lane::vec_call) in lock step; the unit's element loops marked "lanes" are split, lane k of the sub-group taking the elements
// k, k + vgrp, ... (the lanes-per-element mapping of the reference's loops, vector.cpp:808-855, vector-float.cpp:667-691,
// vector-ls.cpp:156-247).  The unit runs dry (Lane<DevMem, true>): whatever would trap, stop the hart or change mstatus changes nothing and
// comes back as handled = 0; the general path then executes that instruction in full.  mstatus is synthetic: M mode, VS Dirty, FS
// Dirty or Off (Off sends an FP vector instruction of a hart whose FS is not Dirty yet to the general path).
// Returns in `handled` the number of instructions completed: the call goes on with the instructions that follow while they are
// vector instructions too (at most max_n, the hart's remaining budget), so that a strip-mined loop body costs one call.
// Lanes are grouped by instruction word: each distinct word present is executed once, for the lanes that hold it, with the word
// in scalar registers - the unit's decode and dispatch then run on the scalar unit (the sub-groups of a wave of replicas hold the
// same word, and even a mixed wave pays no more than the divergent form would).
struct VWarmArgs { const Rec* recs; u64 text_lo, text_hi, tohost, stop_pc; };
#ifdef WSP_INSPECT
__device__ WarmRet wsp_vwarm(const DevState* dp, VWarmArgs va, u32 w0, i32 imm, u32 raw, u64 pc, u64 thash, u32 fcsr, u32 flags, u32 max_n);
#else
__device__ __noinline__ WarmRet wsp_vwarm(const DevState* dp, VWarmArgs va, u32 w0, i32 imm, u32 raw, u64 pc, u64 thash, u32 fcsr, u32 flags, u32 max_n)
{
  const DevState& d = *dp;
  // what is the same for every lane of the wave goes through readfirstlane: scalar registers, not one vector register per lane
  const u32 lpw = wsp_uni32(d.lpw), lpw_shift = wsp_uni32(d.lpw_shift), S = wsp_uni32(d.S), vlenb = wsp_uni32(d.vlenb);
  const u32 wave = blockIdx.x, ln = threadIdx.x & (lpw - 1), slot = wave * 64 + ln;
  Lane<DevMem, true> L;
  L.m.xs = wsp_lds + ln * WSP_XSLOTS;
  L.m.fs = wsp_lds + (lpw + ln) * WSP_XSLOTS;
  L.m.cold_ = (u64*)(size_t)wsp_uni64((u64)(size_t)d.cold) + slot;
  L.m.tlb_ = nullptr; L.tlb_n = 0;
  L.m.voff = lpw * (2 * WSP_XSLOTS * 8) + ln * 32 * vlenb; L.vlenb = vlenb;
  L.m.vcsr_ = wsp_lds + lpw * (2 * WSP_XSLOTS + 4 * vlenb) + ln * 4;
  L.ring_ = nullptr; L.ring_pos_ = nullptr; L.ring_cap = 0;      // (no ring on this path: HF_RING keeps a hart off it)
  L.mem_base = wsp_uni64(d.mem_base); L.mem_size = wsp_uni64(d.mem_size); L.isa = wsp_uni32(d.isa);
  L.m.arena = (u8*)(size_t)wsp_uni64((u64)(size_t)d.arena) + (size_t)wave * lpw * L.mem_size + (size_t)ln * 8;
  L.m.S = S; L.m.lpw_shift = lpw_shift;
  L.hart_id = wsp_uni32(d.hart_id_base) + wave * lpw + ln;
  L.tohost = va.tohost; L.stop_pc = ~0ULL; L.text_lo = va.text_lo; L.text_hi = va.text_hi;
  L.vsub = threadIdx.x >> lpw_shift; L.vgrp = 64u >> lpw_shift;
  L.h.mstatus = MSTATUS_RESET | (3ULL << MST_VS_SH) | ((flags & HF_FS_DIRTY) ? (3ULL << MST_FS_SH) : 0) | MST_SD;
  L.h.cycle = 0; L.h.instret = 0; L.h.exec = 0; L.h.retired = 0; L.h.thash = thash; L.h.lr_addr = 0; L.h.exec_last_ill = 0; L.h.stop_value = 0;
  L.h.fcsr = fcsr; L.h.priv = PRV_M; L.h.status = ST_RUNNING; L.h.consec_ill = 0; L.h.htif_dev = 0;
  L.h.flags = flags;                                             // HF_NOTRACE (set per launch) makes commit() return at once
  u32 n_done = 0;
  bool alive = true;
  const u32 text_len = (u32)(va.text_hi - va.text_lo);
  // (WSP_VW_GROUP: execute each distinct instruction word once with the word in scalar registers.  Measured slower than letting
  // the lanes diverge - 0.39 vs 0.49 GIPS on the vector-kernel workload - and off by default; the burst is worth +43 %.)
#ifndef WSP_VW_GROUP
#define WSP_VW_GROUP 0
#endif
#ifndef WSP_VW_BURST
#define WSP_VW_BURST 1
#endif
  if (!WSP_VW_BURST) max_n = 1;
  for (;;) {
    bool done = false;
#define WSP_VW_ONE(W0, IMM, RAW) do {                                                                                             \
      L.h.pc = pc + 4;                                                                                                            \
      L.s.chg = 0; L.s.ireg = 0; L.s.freg = 0; L.s.gen_csr = 0; L.s.vreg = 0; L.s.vgroup = 0; L.s.ival = 0; L.s.fval = 0; L.s.maddr = 0; L.s.mval = 0; L.s.lva = 0; L.s.lpa = 0; \
      Rec rec; rec.w0 = (W0); rec.imm = (IMM); rec.pad = 0; rec.raw = (RAW);                                                      \
      const u32 op = rec.w0 & 0xff;                                                                                               \
      const u64 a = L.m.rx(rec_rs1(rec)), b = (rec.w0 & F_IMM) ? (u64)(i64)rec.imm : L.m.rx(rec_rs2(rec));                         \
      if (op == OP_VEC) L.exec_vector_body(rec, a, b, pc); else L.exec_vector_ldst_body(rec, a, b, pc, op == OP_VST);             \
      done = !(L.s.chg & CH_BAIL);                                                                                                \
      if (done) L.template commit<true>(pc, rec.raw, 4, PRV_M, 0);                                                                \
    } while (0)
#if WSP_VW_GROUP
    // ---- one instruction per live lane: every distinct word present, once, with the word in scalar registers
    bool todo = alive;
    unsigned long long pend = __ballot(alive);
    while (pend) {
      const int leader = __ffsll((long long)pend) - 1;
      const u32 uraw = (u32)__builtin_amdgcn_readlane((int)raw, leader);
      const bool mine = todo && raw == uraw;
      if (mine) {
        const u32 uw0 = (u32)__builtin_amdgcn_readlane((int)w0, leader); const i32 uimm = (i32)__builtin_amdgcn_readlane((int)imm, leader);
        WSP_VW_ONE(uw0, uimm, uraw);
        todo = false;
      }
      pend &= ~__ballot(mine);
    }
#else
    // the record of the sequential successor is requested before this instruction executes: its latency (a dependent trip to
    // L2 / HBM per instruction of the burst otherwise) hides behind the element loops
    const u64 nidx = pc + 4 - va.text_lo;
    const bool nin = alive && nidx < (u64)text_len;
    uint4 qn; qn.x = 0; qn.y = 0; qn.z = 0; qn.w = 0;
    if (nin) qn = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(va.recs) + (nidx << 3));
    if (alive) WSP_VW_ONE(w0, imm, raw);
#endif
#undef WSP_VW_ONE
    // ---- go on while the next instruction is a vector instruction too
    if (alive && done) {
      ++n_done; pc += 4;
#if WSP_VW_GROUP
      const u64 idx = pc - va.text_lo;
      alive = n_done < max_n && idx < (u64)text_len;
      uint4 qn; qn.x = 0; qn.y = 0; qn.z = 0; qn.w = 0;
      if (alive) qn = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(va.recs) + (idx << 3));
#else
      alive = n_done < max_n && nin;
#endif
      if (alive) {
        w0 = qn.x; imm = (i32)qn.y; raw = qn.w;
        const u32 op = w0 & 0xff;
        alive = op >= OP_VEC && op <= OP_VST;
      }
    } else alive = false;
    if (!__any(alive)) break;
  }
  WarmRet r; r.thash = L.h.thash; r.fcsr = L.h.fcsr; r.handled = n_done;
  return r;
}
#endif

// The translated warm path: S / U-mode harts under Sv39 (never hot: the hot loops run M-mode code at physical addresses).  Each
// lane runs its own hart through Lane::fast_step_translated - TLB-hit fetch and data translation, PMP by page, the pre-decoded
// record by physical address, the inline instruction subset - for as long as every step is such a step (at most max_n); the
// first instruction that is not (a TLB miss, a fault, a CSR access, ecall, a misaligned access ...) ends the lane's run with
// nothing changed, and the general path executes it.  The hart is register resident: x registers in LDS, pc / privilege /
// mstatus / digest in registers, the TLB and the PMP rows read from HBM; nothing of the hart's HBM state is loaded or stored
// beyond that (the four counters advance with the executed count, as for hot steps).
#ifndef WSP_SWARM
#define WSP_SWARM 1                                      // 0: build without the translated warm path (A/B measurements)
#endif
struct SWarmRet { u64 pc, thash; u32 n, fcsr; };
#ifdef WSP_INSPECT
template <bool TRACE> __device__ SWarmRet wsp_swarm(const DevState* dp, VWarmArgs va, u64 pc, u64 thash, u32 flags, u32 fcsr, u32 max_n);
#else
template <bool TRACE>
__device__ __noinline__ SWarmRet wsp_swarm(const DevState* dp, VWarmArgs va, u64 pc, u64 thash, u32 flags, u32 fcsr, u32 max_n)
{
  const DevState& d = *dp;
  const u32 lpw = wsp_uni32(d.lpw), lpw_shift = wsp_uni32(d.lpw_shift), S = wsp_uni32(d.S), tlb_n = wsp_uni32(d.tlb_n);
  const u32 wave = blockIdx.x, ln = ln_of(d), slot = wave * 64 + ln;
  Lane<DevMem> L;
  L.m.xs = wsp_lds + ln * WSP_XSLOTS;
  L.m.fs = wsp_lds + (lpw + ln) * WSP_XSLOTS;
  L.m.cold_ = (u64*)(size_t)wsp_uni64((u64)(size_t)d.cold) + slot;
  L.m.tlb_ = (u64*)(size_t)wsp_uni64((u64)(size_t)d.tlb) + (size_t)(wave * lpw + ln) * 2 * tlb_n; L.tlb_n = tlb_n;
  L.m.voff = 0; L.vlenb = 0; L.m.vcsr_ = nullptr;
  L.ring_ = nullptr; L.ring_pos_ = nullptr; L.ring_cap = 0;
  L.mem_base = wsp_uni64(d.mem_base); L.mem_size = wsp_uni64(d.mem_size); L.isa = wsp_uni32(d.isa);
  L.m.arena = (u8*)(size_t)wsp_uni64((u64)(size_t)d.arena) + (size_t)wave * lpw * L.mem_size + (size_t)ln * 8;
  L.m.S = S; L.m.lpw_shift = lpw_shift;
  L.hart_id = 0; L.tohost = va.tohost; L.stop_pc = ~0ULL; L.text_lo = va.text_lo; L.text_hi = va.text_hi;
  L.vsub = 0; L.vgrp = 1;
  L.h.pc = pc; L.h.thash = thash; L.h.flags = flags; L.h.status = ST_RUNNING; L.h.fcsr = fcsr;
  L.h.priv = d.hot32[H32_PRIV * S + slot]; L.h.mstatus = d.hot64[(size_t)H64_MSTATUS * S + slot];
  L.h.cycle = 0; L.h.instret = 0; L.h.exec = 0; L.h.retired = 0; L.h.lr_addr = 0; L.h.exec_last_ill = 0; L.h.stop_value = 0; L.h.consec_ill = 0; L.h.htif_dev = 0;
  L.s.chg = 0; L.s.ireg = 0; L.s.freg = 0; L.s.gen_csr = 0; L.s.vreg = 0; L.s.vgroup = 0; L.s.ival = 0; L.s.fval = 0; L.s.maddr = 0; L.s.mval = 0; L.s.lva = 0; L.s.lpa = 0;
  SWarmRet r; r.pc = pc; r.thash = thash; r.n = 0; r.fcsr = fcsr;
  if (!L.fast_tr_allowed()) return r;
  Lane<DevMem>::FastTr t; L.fast_tr_begin(t);
  u32 n = 0;
  // the stop address is a virtual one: the record patch of wsp_batch_set_stop sits at the physical slot of the same number, so
  // with VA != PA only this compare ends the run there (Hart.cpp:6063 compares the pc)
  while (n < max_n && L.h.pc != va.stop_pc && L.template fast_step_translated<TRACE>(va.recs, t)) ++n;
  r.pc = L.h.pc; r.thash = L.h.thash; r.n = n; r.fcsr = L.h.fcsr;
  return r;
}
#endif

// ------------------------------------------------------------------------------------------ compiled semantics called from the hand-written loop
// Pure functions of their arguments (no hart state, no memory): the loop (wsp_conv_core.h) marshals operands into the argument
// registers of the standard calling convention and calls them by name.  Every lane active at the call holds the same instruction,
// so the record goes through scalar registers and the dispatch inside runs as scalar branches.  The register file of the f
// operands is a three-entry stand-in (PureMem): exec_fp reads f[rs1] / f[rs2] / f[rs3] and writes one result.
struct PureMem : DevMem {
  u64 fv[3]; u32 fr[3]; u64 out;
  __device__ __forceinline__ u64 rf(u32 r) const { return r == fr[0] ? fv[0] : r == fr[1] ? fv[1] : fv[2]; }
  __device__ __forceinline__ void wf(u32, u64 v) { out = v; }
  __device__ __forceinline__ void wx(u32, u64 v) { out = v; }
};
struct WspPureRet { u64 val; u32 fcsr, ok; };
#ifdef WSP_INSPECT
extern "C" __device__ WspPureRet wsp_pure_fp(u32 w0, i32 imm, u64 xa, u64 f1, u64 f2, u64 f3, u32 fcsr, u32 isa);
extern "C" __device__ u64 wsp_pure_int(u32 w0, i32 imm, u64 a, u64 b, u32 isa);
#else
extern "C" __device__ __noinline__ __attribute__((used)) WspPureRet wsp_pure_fp(u32 w0, i32 imm, u64 xa, u64 f1, u64 f2, u64 f3, u32 fcsr, u32 isa)
{
  w0 = wsp_uni32(w0); imm = (i32)wsp_uni32((u32)imm); isa = wsp_uni32(isa);
  {
    // The common double-precision arithmetic first, without the hart object: round to nearest even and ordinary operands take
    // the FP unit's paths (hw_* in wsp_softfloat.h: the machine result plus the exact inexact flag); a lane outside their domain,
    // or under another rounding mode, goes on to the general routine below.
    using namespace wsf;
    const u32 op0 = w0 & 0xff;
    if ((w0 & F_DBL) && (op0 == OP_FADD || op0 == OP_FSUB || op0 == OP_FMUL || op0 == OP_FDIV || op0 == OP_FMA || op0 == OP_FCVT_I2F)) {
      u32 rm0 = (u32)imm & 7; if (rm0 == 7) rm0 = (fcsr >> 5) & 7;
      u64 z = 0; unsigned fl = 0; bool hw = false;
      if (rm0 == RM_RNE) {
        if (op0 == OP_FADD) hw = hw_add64(f1, f2, z, fl);
        else if (op0 == OP_FSUB) hw = hw_add64(f1, f2 ^ F64::SIGN, z, fl);
        else if (op0 == OP_FMUL) hw = hw_mul64(f1, f2, z, fl);
        else if (op0 == OP_FDIV) hw = hw_div64(f1, f2, z, fl);
        else if (op0 == OP_FMA) { const u32 k = ((u32)imm >> 3) & 3; hw = hw_fma64((k == 2 || k == 3) ? f1 ^ F64::SIGN : f1, f2, (k == 1 || k == 3) ? f3 ^ F64::SIGN : f3, z, fl); }
        else { const u32 rs2 = (w0 >> 18) & 31; z = hw_int_to_f64(xa, !(rs2 & 1), (rs2 & 2) != 0, fl); hw = true; }
      }
      if (hw) { WspPureRet q; q.val = z; q.fcsr = fcsr | fl; q.ok = 1; return q; }
    }
  }
  Lane<PureMem, true> L;
  L.isa = isa; L.h.fcsr = fcsr; L.h.mstatus = MSTATUS_RESET | (3ULL << MST_FS_SH) | MST_SD; L.h.flags = HF_NOTRACE; L.h.status = ST_RUNNING; L.h.priv = PRV_M;
  L.s.chg = 0;
  Rec rec; rec.w0 = w0; rec.imm = imm; rec.pad = 0; rec.raw = 0;
  L.m.fr[0] = rec_rs1(rec); L.m.fr[1] = rec_rs2(rec); L.m.fr[2] = rec_rs3(rec); L.m.fv[0] = f1; L.m.fv[1] = f2; L.m.fv[2] = f3; L.m.out = 0;
  WspPureRet r; r.val = 0; r.fcsr = fcsr; r.ok = 0;
  const u32 op = w0 & 0xff;
  const bool uses_rm = (op >= OP_FADD && op <= OP_FMA) || op == OP_FCVT_F2I || op == OP_FCVT_I2F || op == OP_FCVT_FF;
  int rm = 0;
  if (uses_rm && !L.fp_rm((u32)imm & 7, rm)) return r;          // reserved rounding mode: the general path raises the trap
  L.exec_fp(rec, xa, 0);
  if (L.s.chg & CH_BAIL) return r;
  r.val = L.m.out; r.fcsr = L.h.fcsr; r.ok = 1;
  return r;
}
extern "C" __device__ __noinline__ __attribute__((used)) u64 wsp_pure_int(u32 w0, i32 imm, u64 a, u64 b, u32 isa)
{
  w0 = wsp_uni32(w0); imm = (i32)wsp_uni32((u32)imm); isa = wsp_uni32(isa);
  Lane<PureMem, true> L; L.isa = isa;
  Rec rec; rec.w0 = w0; rec.imm = imm; rec.pad = 0; rec.raw = 0;
  const u32 op = w0 & 0xff;
  u64 v;
  if (op == OP_ZB) v = L.exec_zb(rec_rs3(rec), (w0 & F_IMM) ? 0 : imm, a, (w0 & F_IMM) ? (u64)(i64)imm : b);
  else v = L.muldiv(op, a, b);
  return (isa & ISA_RV32) ? (u64)(i64)(i32)v : v;                // Lane::wx: an RV32 hart's registers hold the value sign-extended
}
#endif

// Wave-wide minimum of a u32 (all 64 lanes participate).
__device__ __forceinline__ u32 wave_min_u32(u32 v)
{
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const u32 t = (u32)__shfl_xor((int)v, o, 64); v = t < v ? t : v; }
  return v;
}

// The same through DPP row shifts / broadcasts (no LDS permute traffic): result in every lane's return value (read from lane 63).
// All 64 lanes must be active at the call.
__device__ __forceinline__ u32 wave_min_dpp(u32 v)
{
  int x = (int)v;
#define WSP_DPPMIN(ctrl, rmask) { const u32 t_ = (u32)__builtin_amdgcn_update_dpp(-1, x, ctrl, rmask, 0xf, false); x = (int)(t_ < (u32)x ? t_ : (u32)x); }
  WSP_DPPMIN(0x111, 0xf) WSP_DPPMIN(0x112, 0xf) WSP_DPPMIN(0x114, 0xf) WSP_DPPMIN(0x118, 0xf)     // row_shr 1, 2, 4, 8: lane 15 of each row = row minimum
  WSP_DPPMIN(0x142, 0xa) WSP_DPPMIN(0x143, 0xc)                                                   // row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3
#undef WSP_DPPMIN
  return (u32)__builtin_amdgcn_readlane(x, 63);
}

// Record image is plain global memory.  Untraced runs load {w0, imm, pad}; traced runs also the raw instruction bits.
typedef u32 wsp_u32x4 __attribute__((ext_vector_type(4)));
typedef u32 wsp_u32x3 __attribute__((ext_vector_type(3)));
template <bool TRACE> struct RecQ;
template <> struct RecQ<true> { typedef wsp_u32x4 T; static __device__ __forceinline__ u32 raw(const T& q) { return q.w; } };
template <> struct RecQ<false> { typedef wsp_u32x3 T; static __device__ __forceinline__ u32 raw(const T&) { return 0; } };

template <bool TRACE, bool FP>
__global__ void __launch_bounds__(64, WSP_WAVES_PER_EU) wsp_step_kernel(DevState d, const DevState* dp, u64 max_steps, u32 notrace)
{
  // TRACE = false is the lean build of the inline loops for untraced batches in which every wave runs one program image (the
  // host picks it, wsp_batch_run).  Every other untraced batch runs the TRACE = true kernel with notrace set: one compiled body of
  // the divergent dispatch loop for traced and untraced runs, so the traced parity tests cover what untraced runs execute.
  const bool tr = TRACE && !notrace;
  const unsigned long long pk0 = WSP_PROF_T();
  const u32 lane = threadIdx.x, wave = blockIdx.x;
  const u32 hart = wave * d.lpw + lane;
  const u32 S = d.S;
  const bool has_own = lane < d.lpw && hart < d.n_harts;
  // which hart this lane works on during this launch (see "Mirror lanes" above)
  bool mirror = false;
  u32 run_pg = ~0u; bool run_same = false;               // the running harts' common program image, if they have one
  {
    u32 ent = has_own ? lane : 0x200u;
    if (!d.vreg) {
      const bool own_run = has_own && d.hot32[H32_STATUS * S + wave * 64 + lane] == (u32)ST_RUNNING;
      const unsigned long long rm = __ballot(own_run);
      const u32 nrun = (u32)__popcll(rm);
      const u32 pgl = own_run ? d.prog[wave * 64 + lane] : 0;
      const u32 pgf = nrun ? (u32)__builtin_amdgcn_readlane((int)pgl, __ffsll((long long)rm) - 1) : ~0u;
      const bool same = pgf != ~0u && __all(!own_run || pgl == pgf);     // (all lanes vote: outside the branch below)
      run_pg = pgf; run_same = same;
      if (nrun && !own_run) {
        const u32 target = same ? WSP_MIRROR_LANES : 64u;
        const u32 rank = (u32)__popcll(~rm & ((1ULL << lane) - 1));       // this lane's place among the lanes without a running hart
        if (nrun + rank < target) {
          unsigned long long t = rm;
          for (u32 k = rank % nrun; k; --k) t &= t - 1;                   // drop the k lowest running lanes
          ent = (u32)(__ffsll((long long)t) - 1) | 0x100u;
          mirror = true;
        }
      }
    }
    wsp_mtab_of(d)[lane] = ent;
  }
  const bool active = has_own && !mirror;                // this lane's own hart (running or not): its state is loaded and stored here
  // a lane with nothing to do looks at hart 0's slot; in a V batch at the slot of the hart whose sub-group it belongs to (wsp_vwarm)
  const u32 own = lane & (d.lpw - 1);
  const bool live = active || mirror;
  const u32 ln = live ? (wsp_mtab_of(d)[lane] & 0xffu) : (d.vreg && wave * d.lpw + own < d.n_harts) ? own : 0;
  const u32 slot = wave * 64 + ln;

  Lane<DevMem> L; const Rec* recs;
  wsp_lane_setup(L, d, wave, ln, slot, recs);
  L.h.pc = d.hot64[H64_PC * S + slot]; L.h.exec = d.hot64[H64_EXEC * S + slot]; L.h.thash = d.hot64[H64_THASH * S + slot];
  L.h.flags = d.hot32[H32_FLAGS * S + slot]; L.h.status = live ? d.hot32[H32_STATUS * S + slot] : (u32)ST_STOP_LIMIT;
  L.h.priv = d.hot32[H32_PRIV * S + slot]; L.h.mstatus = d.hot64[H64_MSTATUS * S + slot]; L.h.fcsr = d.hot32[H32_FCSR * S + slot];
  L.update_pmp_flags();                                  // host pokes may have changed priv / mstatus / mip / mie / pmpcfg / flags
  L.h.flags = d.ring ? (L.h.flags | HF_RING) : (L.h.flags & ~(u32)HF_RING);
  L.h.flags = tr ? (L.h.flags & ~(u32)HF_NOTRACE) : (L.h.flags | HF_NOTRACE);
  L.update_hot_flag();
  u64 stop_at, slice_end;
  {
    const u64 exec0 = L.h.exec;
    slice_end = exec0 + max_steps < exec0 ? ~0ULL : exec0 + max_steps;
    stop_at = wsp_stop_at(slice_end, d.max_exec[slot], d.max_ret[slot], exec0, d.hot64[H64_RETIRED * S + slot]);
    if (stop_at < exec0) stop_at = exec0;
  }

  if (active) {
    L.m.xs[0] = 0; L.m.xs[32] = 0;
#pragma unroll 4
    for (u32 r = 1; r < 32; ++r) L.m.wx(r, d.x[(size_t)r * S + slot]);
    if (FP) {
#pragma unroll 4
      for (u32 r = 0; r < 32; ++r) L.m.wf(r, d.f[(size_t)r * S + slot]);
    }
  }
  // vector register files of the wave's harts: one contiguous run in HBM -> LDS, all 64 lanes copying
  const u32 vwords = d.vreg ? (min(d.n_harts - wave * d.lpw, d.lpw) * 32 * d.vlenb) >> 3 : 0;
  u64* const vlds = wsp_lds + d.lpw * (2 * WSP_XSLOTS);
  u64* const vhbm = (u64*)(d.vreg + (size_t)wave * d.lpw * 32 * d.vlenb);
  for (u32 i = lane; i < vwords; i += 64) vlds[i] = vhbm[i];
  if (WSP_VCSR_LDS && d.vreg && active) for (u32 k = 0; k < 4; ++k) L.m.vcsr_[k] = d.cold[(size_t)(CI_VSTART + k) * S + slot];
  __syncthreads();

  const bool zca = (d.isa & ISA_C) != 0;                 // without C, taken control transfers can trap: general path
  // RV32 batch (ISA_RV32): addresses wrap at 32 bits, register values are kept sign-extended, commit records carry 32-bit values
  const bool rv32 = (d.isa & ISA_RV32) != 0;
  const u64 am = rv32 ? 0xffffffffULL : ~0ULL;
  // Every hart of the wave runs the same program image: the converged loop below applies (one pc for the running group).
  // (Among the harts that are RUNNING at the start of this launch: a wave of different programs whose other harts have finished
  // runs its last hart - and the copies of it - through the converged loop too.  V batches: among all the wave's harts.)
  const u32 pg0 = d.vreg ? (u32)__builtin_amdgcn_readfirstlane((int)d.prog[wave * 64]) : (u32)__builtin_amdgcn_readfirstlane((int)run_pg);
  const bool same_prog = d.vreg ? (pg0 != ~0u && __all(!active || d.prog[slot] == pg0)) : run_same;

  // Outer loop: { hot inner loop (no calls) ; one general step for the lanes that asked for it }.
  //
  // Hot inner loop.  Entered only when every running lane of the wave is hot (HF_HOT) with its pc inside its pre-decoded
  // text; runs under an exec mask of exactly those lanes for n = min remaining budget iterations (wave-uniform trip
  // count: nothing in the loop can stop a hart, so "running" does not change inside).  The pc is kept as a 32-bit offset
  // into the text.  Per iteration: load the 16-byte record, read both source registers from LDS through the
  // pre-computed byte offsets in Rec.pad, then run each distinct op present once (wave-uniform switch).  A lane that
  // meets anything outside the inline subset (an op that is not inlined, a misaligned / out-of-arena / tohost / text
  // access, a control transfer leaving the text) changes nothing, raises `nc` and the loop is left after that iteration;
  // the general step then executes that same instruction in full.
  u32 swarm_credit = 0;                                  // > 0: wsp_swarm's last runs with this hart were long ones (see the call site)
  u32 swarm_skip = 0, swarm_backoff = 0;                 // general-path rounds to let pass before wsp_swarm is tried again
  while (true) {
    const bool run = L.h.status == ST_RUNNING && L.h.exec < stop_at;
    if (!__any(run)) break;                              // every lane of the wave has stopped or used its budget
    const u64 off64 = L.h.pc - L.text_lo;
    const u32 text_len = (u32)(L.text_hi - L.text_lo);   // loader keeps every text image below 1 GiB
    const bool can = run && (L.h.flags & HF_HOT) && off64 < (u64)text_len;
    bool need_cold = run && !can;

    const unsigned long long ph0 = WSP_PROF_T();
    // The hot lanes run whenever there are any.  While other lanes of the wave wait for one of the out-of-line paths (a wave of
    // different programs: some lane is nearly always in S / U-mode code or at a trap) the hot lanes get a quantum of
    // WSP_HOT_QUANTUM iterations, then the other paths take their turn: every class of lane advances in every round.
    if (__any(can)) {
      const u64 left = stop_at - L.h.exec;
      u32 n = (u32)__builtin_amdgcn_readfirstlane((int)wave_min_u32(can ? (left > 0x40000000ULL ? 0x40000000u : (u32)left) : 0xffffffffu));
      if (__any(need_cold) && n > WSP_HOT_QUANTUM) n = WSP_HOT_QUANTUM;
      if (WSP_CONVERGED_LOOP && same_prog) {
        // ============================================================================================================
        // Converged loop (all harts of the wave share one program image).  One pc - `upc`, an SGPR - is scheduled per
        // iteration: the record comes through the scalar cache and is decoded on the scalar unit (op, register offsets,
        // immediate, size all wave-uniform), and exactly the lanes whose own pc `vpc` equals `upc` execute it, together,
        // on the vector unit.  Lanes at other pcs wait; a lane whose pc becomes `upc` joins by itself (the membership
        // test is one compare per iteration).  After a control transfer the next `upc` is the minimum pc over all
        // eligible lanes, so that lanes which left a loop or a branch arm early wait at the join point for the others
        // (min-pc reconvergence); after any other instruction it is simply the sequential successor.  Every iteration
        // executes at least one lane (upc is always some eligible lane's pc), nothing inside can stop a hart, and the trip
        // count is bounded by the smallest remaining budget, so the loop ends.
        typedef wsp_u32x4 SQ;
        typedef const __attribute__((address_space(4))) SQ* SQP;            // constant address space: scalar loads
        const ProgDev pd = d.progs[pg0];
        const u64 tlo = (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)pd.text_lo) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(pd.text_lo >> 32)) << 32);
        const u64 thi = (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)pd.text_hi) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(pd.text_hi >> 32)) << 32);
        const u64 recs_a = (u64)(size_t)pd.recs;
        const char* const recs_u = (const char*)(size_t)((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)recs_a) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(recs_a >> 32)) << 32));
        const u32 tlen = (u32)(thi - tlo);
        const u64 tohost = L.tohost, mbase = d.mem_base, msize = d.mem_size;
        const u32 xb = (u32)(size_t)((char*)L.m.xs - (char*)wsp_lds);      // byte offset of this lane's x block
        const char* const ldsb = (const char*)wsp_lds;
        u64* const sinkx = (u64*)(ldsb + xb);
        const u32 foff = d.lpw * (WSP_XSLOTS * 8);
        const bool fsd = (L.h.flags & HF_FS_DIRTY) != 0, boxd = (d.isa & ISA_D) != 0;
        u8* const gsink = (u8*)(d.cold + (size_t)CI_COUNT * S + (blockIdx.x * 64 + ln_of(d)));
        u32 vpc = can ? (u32)off64 : 0xffffffffu;          // lanes that take no part never match a scheduled pc
        u32 vit = 0, ncv = 0, left_it = n;
#if WSP_CONV_ASM
        // the hand-written loop (wsp_conv_core.h), same contract, traced or not; it reads the program's hot records (the C++ loop
        // below remains for images too large for them)
        const u64 hrecs_a = (u64)(size_t)pd.hrecs;
        const bool use_asm = hrecs_a != 0;
        if (use_asm) {
          const bool fsd_all = FP && __all(!can || fsd);
          const u32 cfg = (rv32 ? 2u : 0u) | (fsd_all ? 4u : 0u) | (boxd ? 8u : 0u) | ((TRACE && tr) ? (1u << 17) : 0u);
          u32 vfcsr = L.h.fcsr; u64 vth = L.h.thash;
          conv_core((const void*)(size_t)hrecs_a, (const void*)recs_u, mbase, msize, tlo, thi, tlen, cfg, d.lpw_shift, foff, n, d.isa, xb, L.m.arena, tohost, vpc,
                    vfcsr, vth, vit, ncv);
          if (can) { L.h.fcsr = vfcsr; L.h.thash = vth; }
        } else {
#endif
        u32 upc = wave_min_dpp(vpc);
        SQ q = *(SQP)(recs_u + ((u64)upc << 3));
        do {
          // operand reads first (every lane reads its own block: lanes outside the group discard the values), the rest of
          // the decode runs on the scalar unit while they are in flight
          const u32 ofs = q.z;
          u64 a = *(const u64*)(ldsb + (xb + (ofs & 0x1ffu)));
          u64 b2 = *(const u64*)(ldsb + (xb + (ofs >> 16)));
          const u32 w0 = q.x; const i32 imm = (i32)q.y; const u32 raw = q.w;
          const u32 op = w0 & 0xff;
          const u32 size = 2u + ((w0 >> 27) & 2u);
          const u32 spc = upc + size;
          const u64 simm = (u64)(i64)imm;
          // The operands must have arrived before the next record is requested: LDS and scalar loads share one counter
          // (lgkmcnt) and scalar loads return out of order, so a record load issued earlier would be waited for right
          // here.  Issued now it is in flight for the rest of the step and is first needed at the top of the next one.
          // (The address of the record load is made to depend on an operand that has arrived: a zero derived from it.)
          asm volatile("" : "+v"(a), "+v"(b2));
          u32 zero;
          asm volatile("s_and_b32 %0, %1, 0" : "=s"(zero) : "s"(__builtin_amdgcn_readfirstlane((int)(u32)a)) : "scc");
          zero = (u32)__builtin_amdgcn_readfirstlane((int)zero);   // (an asm result counts as divergent: say that it is not)
          const SQ qn = *(SQP)(recs_u + ((u64)(spc + zero) << 3));
          if (vpc == upc) {
            u64* const rdp = (u64*)(ldsb + (xb + ((w0 >> 5) & 0xf8u)));
            const u64 bi = b2 + simm;
            u32 npc = spc; bool bad = false;
            u64 tr_k = 1ULL << 60, tr_v = 0;
            const bool rd_nz = (w0 & 0x1f00u) != 0;
            const u64 rd_key = rd_nz ? ((u64)((w0 >> 8) & 31) ^ (1ULL << 60)) : (1ULL << 60);
#define WSP_CW(val) do { const u64 v_ = (val); *rdp = v_; if (TRACE) { tr_k = rd_key; tr_v = rd_nz ? (v_ & am) : 0; } } while (0)
#define WSP_BAD(cond) do { bad = (cond); } while (0)
            if (op == OP_ADD) WSP_CW(a + bi);
            else if (op >= OP_LB && op <= OP_LWU) {
              const u32 k = op - OP_LB, sz = 1u << (k & 3);
              const u64 va = (a + simm) & am, off = va - mbase;
              const bool ok = ((va & (sz - 1)) == 0) & (off <= msize - sz) & (va != tohost);
              const u8* p = L.m.p(ok ? off : 0);
              u64 v;
              if (k == 2) v = (u64)(i64)*(const i32*)p; else if (k == 3) v = *(const u64*)p;
              else if (k == 6) v = *(const u32*)p; else if (k == 4) v = *p; else if (k == 0) v = (u64)(i64)*(const int8_t*)p;
              else if (k == 1) v = (u64)(i64)*(const int16_t*)p; else v = *(const u16*)p;
              *(ok ? rdp : sinkx) = v;
              WSP_BAD(!ok);
              if (TRACE) { tr_k = rd_key; tr_v = rd_nz ? (v & am) : 0; }
            } else if (op >= OP_BEQ && op <= OP_BGEU) {
              const u32 m = op - OP_BEQ;
              const bool c = m < 2 ? a == b2 : m < 4 ? (i64)a < (i64)b2 : a < b2;
              const bool take = c != (bool)(m & 1);
              const u32 t = upc + (u32)imm;
              const bool ok = zca & (t < tlen);          // wave-uniform
              npc = take ? (t & ~1u) : spc;
              if (!ok) WSP_BAD(take);
            } else if (op == OP_ADDW) WSP_CW((u64)(i64)(i32)((u32)a + (u32)bi));
            else if (op == OP_MULW) WSP_CW((u64)(i64)(i32)((u32)a * (u32)bi));
            else if (op == OP_SLL) WSP_CW(a << (bi & 63));
            else if (op == OP_SRL) WSP_CW(a >> (bi & 63));
            else if (op >= OP_SB && op <= OP_SD) {
              const u32 k = op - OP_SB, sz = 1u << k;
              const u64 va = (a + simm) & am, off = va - mbase;
              const bool ok = ((va & (sz - 1)) == 0) & (off <= msize - sz) & (va != tohost) & (va != tohost + 4) &
                              !((va + sz > tlo) & (va < thi));
              u8* p = ok ? L.m.p(off) : gsink;
              const u64 sv = k == 3 ? b2 : (b2 & ((1ULL << (8 * sz)) - 1));
              if (k == 2) *(u32*)p = (u32)sv; else if (k == 3) *(u64*)p = sv; else if (k == 0) *p = (u8)sv; else *(u16*)p = (u16)sv;
              WSP_BAD(!ok);
              if (TRACE) { tr_k = va ^ (4ULL << 60); tr_v = sv; }
            } else if (op == OP_AND) WSP_CW(a & bi);
            else if (op == OP_MUL) WSP_CW(a * bi);
            else if (FP && op >= OP_FLW && op <= OP_FSD) {
              const bool st = op >= OP_FSW, dw = (op & 1u) == (OP_FLD & 1u);
              const u32 sz = dw ? 8u : 4u;
              const u64 va = (a + simm) & am, off = va - mbase;
              bool ok = fsd & ((va & (sz - 1)) == 0) & (off <= msize - sz) & (va != tohost);
              if (st) {
                ok = ok & (va != tohost + 4) & !((va + sz > tlo) & (va < thi));
                const u64 fv = *(const u64*)(ldsb + (xb + foff + ((w0 >> 15) & 0xf8u)));
                u8* p = ok ? L.m.p(off) : gsink;
                const u64 sv = dw ? fv : (fv & 0xffffffffULL);
                if (dw) *(u64*)p = sv; else *(u32*)p = (u32)sv;
                if (TRACE) { tr_k = va ^ (4ULL << 60); tr_v = sv; }
              } else {
                const u8* p = L.m.p(ok ? off : 0);
                u64 v;
                if (dw) v = *(const u64*)p; else v = (u64)*(const u32*)p | (boxd ? 0xffffffff00000000ULL : 0);
                *(u64*)(ldsb + (ok ? (xb + foff + ((w0 >> 5) & 0xf8u)) : xb)) = v;
                if (TRACE) { tr_k = (u64)((w0 >> 8) & 31) ^ (2ULL << 60); tr_v = v; }
              }
              WSP_BAD(!ok);
            } else if (op == OP_JAL || op == OP_JALR) {
              const u64 t = op == OP_JAL ? (u64)(upc + (u32)imm) : (((a + simm) & am & ~1ULL) - tlo);
              const bool ok = zca & (t < (u64)tlen);
              const u64 link0 = tlo + upc + size, link = rv32 ? (u64)(i64)(i32)link0 : link0;
              *(ok ? rdp : sinkx) = link;
              npc = (u32)t & ~1u;
              WSP_BAD(!ok);
              if (TRACE) { tr_k = rd_key; tr_v = rd_nz ? link0 : 0; }
            } else switch (op) {
              case OP_SUB: WSP_CW(a - bi); break;
              case OP_SLT: WSP_CW((u64)((i64)a < (i64)bi)); break;
              case OP_SLTU: WSP_CW((u64)(a < bi)); break;
              case OP_XOR: WSP_CW(a ^ bi); break;
              case OP_SRA: WSP_CW((u64)((i64)a >> (bi & 63))); break;
              case OP_OR: WSP_CW(a | bi); break;
              case OP_SUBW: WSP_CW((u64)(i64)(i32)((u32)a - (u32)bi)); break;
              case OP_SLLW: WSP_CW((u64)(i64)(i32)((u32)a << (bi & 31))); break;
              case OP_SRLW: WSP_CW((u64)(i64)(i32)((u32)a >> (bi & 31))); break;
              case OP_SRAW: WSP_CW((u64)(i64)((i32)a >> (bi & 31))); break;
              case OP_AUIPC: { const u64 v0 = tlo + upc + simm; WSP_CW(rv32 ? (u64)(i64)(i32)v0 : v0); break; }
              case OP_NOP: break;
              default: bad = true; break;                 // not in the inline subset
            }
#undef WSP_CW
#undef WSP_BAD
            if (TRACE && tr && !bad) {
              u64 t = L.h.thash;
              t = fold(t, tlo + upc);
              t = fold(t, (u64)raw | ((u64)PRV_M << 32) | ((u64)size << 48));
              t = fold(t, tr_k); t = fold(t, tr_v);
              L.h.thash = t;
            }
            vpc = bad ? vpc : npc;                        // a lane that needs the general step changed nothing and stays put
            vit += bad ? 0u : 1u;
            ncv = bad ? 1u : 0u;
          }
          if (__any(ncv != 0)) break;                     // (a ballot taken inside the group's branch would not be uniform out here)
          u32 nupc = spc;
          if (op >= OP_JAL && op <= OP_BGEU) {            // control transfer: schedule the minimum pc
            nupc = wave_min_dpp(vpc);
            if (!__any(vpc == nupc)) nupc = (u32)__builtin_amdgcn_readlane((int)vpc, __ffsll((long long)__ballot(can)) - 1);   // cannot happen; guards the loop's progress
          }
          q = qn;
          if (nupc != spc) q = *(SQP)(recs_u + ((u64)nupc << 3));
          upc = nupc;
        } while (--left_it);
#if WSP_CONV_ASM
        }
#endif
        if (can) { L.h.exec += vit; L.h.pc = tlo + vpc; }
        need_cold = need_cold || ncv != 0;
      } else if (can) {
        u32 pco = (u32)off64, it = 0;
        u32 ncv = 0;                                      // "needs the general step", as a VGPR flag (no lane-mask algebra)
        const u32 xb = (u32)(size_t)((char*)L.m.xs - (char*)wsp_lds);      // byte offset of this lane's x block
        const char* const ldsb = (const char*)wsp_lds;
        const u64 tlo = L.text_lo, thi = L.text_hi, tohost = L.tohost, mbase = L.mem_base, msize = L.mem_size;
        u64* const sinkx = (u64*)(ldsb + xb);                                  // LDS slot 0 of this lane: register write sink
        const u32 foff = d.lpw * (WSP_XSLOTS * 8);                              // x block -> f block of the same lane
        const bool fsd = (L.h.flags & HF_FS_DIRTY) != 0, boxd = (d.isa & ISA_D) != 0;
        u8* const gsink = (u8*)(d.cold + (size_t)CI_COUNT * S + (blockIdx.x * 64 + ln_of(d)));   // 8-byte global store sink
        // ---- fetch.  pco <= text_len always holds (control transfers are range-checked, sequential flow stops at the two
        // sentinel records behind the image), so record loads need no guard.  The record of the sequentially next
        // instruction is requested as soon as the size of the current one is known and is in flight while the current
        // instruction executes; only a taken control transfer pays the load latency.
        typedef typename RecQ<TRACE>::T QT;
        typedef const __attribute__((address_space(1))) QT* QP;
        QT q = *(QP)((const char*)recs + ((u64)pco << 3));
        do {
          const u32 w0 = q.x; const i32 imm = (i32)q.y; const u32 ofs = q.z; const u32 raw = RecQ<TRACE>::raw(q);
          const u32 op = w0 & 0xff;
          const u64 a = *(const u64*)(ldsb + (xb + (ofs & 0x1ffu)));
          const u64 b2 = *(const u64*)(ldsb + (xb + (ofs >> 16)));         // x[rs2]; the zero slot for immediate forms
          u64* const rdp = (u64*)(ldsb + (xb + ((w0 >> 5) & 0xf8u)));      // rd * 8 (slot 0 = sink)
          const u64 simm = (u64)(i64)imm;
          const u64 bi = b2 + simm;                       // ALU second operand: register forms carry imm = 0
          const u32 size = 2u + ((w0 >> 27) & 2u);        // F_SIZE4 is bit 28
          const u32 spc = pco + size;
          const QT qn = *(QP)((const char*)recs + ((u64)spc << 3));
          u32 npc = spc;
          u64 tr_k = 1ULL << 60, tr_v = 0;                // trace: the change this instruction made (default: none)
          const u64 rd_key = (w0 & 0x1f00u) ? ((u64)((w0 >> 8) & 31) ^ (1ULL << 60)) : (1ULL << 60);   // x0: "no change"

          // One op for the lanes in EXEC.  Every case is straight-line (selects, no divergent branch and no short-circuit
          // evaluation); the most frequent ops sit at the top of the compare chain.
#define WSP_WX(val) do { const u64 v_ = (val); *rdp = v_; if (TRACE) { tr_k = rd_key; tr_v = (w0 & 0x1f00u) ? (v_ & am) : 0; } } while (0)
#define WSP_HOT_OP(uop, aa, bb, cc)                                                                                        \
          if (uop == OP_ADD) WSP_WX(aa + cc);                                                                              \
          else if (uop >= OP_LB && uop <= OP_LWU) {                                                                        \
            const u32 k = uop - OP_LB, sz = 1u << (k & 3);                                                                 \
            const u64 va = (aa + simm) & am, off = va - mbase;                                                                    \
            const bool ok = ((va & (sz - 1)) == 0) & (off <= msize - sz) & (va != tohost);                                 \
            const u8* p = L.m.p(ok ? off : 0);            /* rejected lanes read their own dword 0 and discard it */     \
            u64 v;                                                                                                         \
            if (k == 2) v = (u64)(i64)*(const i32*)p; else if (k == 3) v = *(const u64*)p;                                 \
            else if (k == 6) v = *(const u32*)p; else if (k == 4) v = *p; else if (k == 0) v = (u64)(i64)*(const int8_t*)p; \
            else if (k == 1) v = (u64)(i64)*(const int16_t*)p; else v = *(const u16*)p;                                    \
            *(ok ? rdp : sinkx) = v;                                                                                       \
            ncv |= ok ? 0u : 1u;                                                                                           \
            if (TRACE) { tr_k = rd_key; tr_v = (w0 & 0x1f00u) ? (v & am) : 0; }                                            \
          } else if (uop >= OP_BEQ && uop <= OP_BGEU) {                                                                    \
            const u32 m = uop - OP_BEQ;                                                                                    \
            const bool c = m < 2 ? aa == bb : m < 4 ? (i64)aa < (i64)bb : aa < bb;                                         \
            const bool take = c != (bool)(m & 1);                                                                          \
            const u32 t = pco + (u32)imm;                                                                                  \
            const bool ok = zca & (t < text_len);                                                                          \
            npc = (take & ok) ? (t & ~1u) : npc;                                                                           \
            ncv |= (take & !ok) ? 1u : 0u;                                                                                 \
          } else if (uop == OP_ADDW) WSP_WX((u64)(i64)(i32)((u32)aa + (u32)cc));                                           \
          else if (uop == OP_MULW) WSP_WX((u64)(i64)(i32)((u32)aa * (u32)cc));                                             \
          else if (uop == OP_SLL) WSP_WX(aa << (cc & 63));                                                                 \
          else if (uop == OP_SRL) WSP_WX(aa >> (cc & 63));                                                                 \
          else if (uop >= OP_SB && uop <= OP_SD) {                                                                         \
            const u32 k = uop - OP_SB, sz = 1u << k;                                                                       \
            const u64 va = (aa + simm) & am, off = va - mbase;                                                                    \
            const bool ok = ((va & (sz - 1)) == 0) & (off <= msize - sz) & (va != tohost) & (va != tohost + 4) &           \
                            !((va + sz > tlo) & (va < thi));                                                               \
            u8* p = ok ? L.m.p(off) : gsink;              /* rejected lanes store into their private sink */              \
            const u64 sv = k == 3 ? bb : (bb & ((1ULL << (8 * sz)) - 1));                                                  \
            if (k == 2) *(u32*)p = (u32)sv; else if (k == 3) *(u64*)p = sv; else if (k == 0) *p = (u8)sv; else *(u16*)p = (u16)sv; \
            ncv |= ok ? 0u : 1u;                                                                                           \
            if (TRACE) { tr_k = va ^ (4ULL << 60); tr_v = sv; }                                                            \
          } else if (uop == OP_AND) WSP_WX(aa & cc);                                                                       \
          else if (uop == OP_MUL) WSP_WX(aa * cc);                                                                         \
          else if (FP && uop >= OP_FLW && uop <= OP_FSD) {                                                                 \
            /* flw fld fsw fsd with FS already Dirty (float.cpp:281-321, 1019-1057): f registers in LDS, no CSR change */  \
            const bool st = uop >= OP_FSW, dw = (uop & 1u) == (OP_FLD & 1u);                                               \
            const u32 sz = dw ? 8u : 4u;                                                                                   \
            const u64 va = (aa + simm) & am, off = va - mbase;                                                                    \
            bool ok = fsd & ((va & (sz - 1)) == 0) & (off <= msize - sz) & (va != tohost);                                 \
            if (st) {                                                                                                      \
              ok = ok & (va != tohost + 4) & !((va + sz > tlo) & (va < thi));                                              \
              const u64 fv = *(const u64*)(ldsb + (xb + foff + ((w0 >> 15) & 0xf8u)));     /* f[rs2], raw bits */         \
              u8* p = ok ? L.m.p(off) : gsink;                                                                             \
              const u64 sv = dw ? fv : (fv & 0xffffffffULL);                                                               \
              if (dw) *(u64*)p = sv; else *(u32*)p = (u32)sv;                                                              \
              if (TRACE) { tr_k = va ^ (4ULL << 60); tr_v = sv; }                                                          \
            } else {                                                                                                       \
              const u8* p = L.m.p(ok ? off : 0);                                                                           \
              u64 v;                                                                                                       \
              if (dw) v = *(const u64*)p; else v = (u64)*(const u32*)p | (boxd ? 0xffffffff00000000ULL : 0);               \
              *(u64*)(ldsb + (ok ? (xb + foff + ((w0 >> 5) & 0xf8u)) : xb)) = v;                                          \
              if (TRACE) { tr_k = (u64)((w0 >> 8) & 31) ^ (2ULL << 60); tr_v = v; }                                        \
            }                                                                                                              \
            ncv |= ok ? 0u : 1u;                                                                                           \
          } else if (uop == OP_JAL || uop == OP_JALR) {                                                                      \
            const u64 t = uop == OP_JAL ? (u64)(pco + (u32)imm) : (((aa + simm) & am & ~1ULL) - tlo);                      \
            const bool ok = zca & (t < (u64)text_len);                                                                     \
            const u64 link0 = tlo + pco + size, link = rv32 ? (u64)(i64)(i32)link0 : link0;                                \
            *(ok ? rdp : sinkx) = link;                                                                                    \
            npc = ok ? ((u32)t & ~1u) : npc;                                                                               \
            ncv |= ok ? 0u : 1u;                                                                                           \
            if (TRACE) { tr_k = rd_key; tr_v = (w0 & 0x1f00u) ? link0 : 0; }                                               \
          } else switch (uop) {                                                                                            \
            case OP_SUB: WSP_WX(aa - cc); break;                                                                           \
            case OP_SLT: WSP_WX((u64)((i64)aa < (i64)cc)); break;                                                          \
            case OP_SLTU: WSP_WX((u64)(aa < cc)); break;                                                                   \
            case OP_XOR: WSP_WX(aa ^ cc); break;                                                                           \
            case OP_SRA: WSP_WX((u64)((i64)aa >> (cc & 63))); break;                                                       \
            case OP_OR: WSP_WX(aa | cc); break;                                                                            \
            case OP_SUBW: WSP_WX((u64)(i64)(i32)((u32)aa - (u32)cc)); break;                                               \
            case OP_SLLW: WSP_WX((u64)(i64)(i32)((u32)aa << (cc & 31))); break;                                            \
            case OP_SRLW: WSP_WX((u64)(i64)(i32)((u32)aa >> (cc & 31))); break;                                            \
            case OP_SRAW: WSP_WX((u64)(i64)((i32)aa >> (cc & 31))); break;                                                 \
            case OP_AUIPC: { const u64 v0 = tlo + pco + simm; WSP_WX(rv32 ? (u64)(i64)(i32)v0 : v0); break; }            \
            case OP_NOP: break;                                                                                            \
            default: ncv = 1u; break;                     /* not in the inline subset */                                  \
          }

          // ---- dispatch.  Converged wave (one op in all lanes, always so with one hart per wave): a single pass with the
          // op in an SGPR.  Otherwise loop over the distinct ops present.
          const u32 op0 = (u32)__builtin_amdgcn_readfirstlane((int)op);
          if (__all(op == op0)) {
            WSP_HOT_OP(op0, a, b2, bi)
          } else {
            bool todo = true;
            unsigned long long pend = __ballot(true);
            while (pend) {
              const int leader = __ffsll((long long)pend) - 1;
              const u32 lop = (u32)__builtin_amdgcn_readlane((int)op, leader);
              const bool mine = todo && op == lop;
              if (mine) {
                // Launder the uniform op and the operands: without this the compiler (a) proves lop == op inside this
                // branch and switches on the per-lane VGPR copy (a divergent, exec-masked switch), and (b) hoists every
                // case body out of the loop because the operands are invariant in it.
                u32 uop = lop; u64 aa = a, bb = b2, cc = bi;
                asm volatile("" : "+v"(uop), "+v"(aa), "+v"(bb), "+v"(cc));
                uop = (u32)__builtin_amdgcn_readfirstlane((int)uop);
                WSP_HOT_OP(uop, aa, bb, cc)
                todo = false;
              }
              pend &= ~__ballot(mine);
            }
          }
#undef WSP_HOT_OP
#undef WSP_WX
          if (TRACE && tr && !ncv) {                      // commit record of an M-mode instruction without trap or CSR change
            u64 t = L.h.thash;
            t = fold(t, tlo + pco);
            t = fold(t, (u64)raw | ((u64)PRV_M << 32) | ((u64)size << 48));
            t = fold(t, tr_k); t = fold(t, tr_v);
            L.h.thash = t;
          }
          pco = ncv ? pco : npc;
          q = qn;
          if (pco != spc) q = *(QP)((const char*)recs + ((u64)pco << 3));
          ++it;
        } while (!__any(ncv != 0) && it < n);
        L.h.exec += ncv ? it - 1 : it;                    // the lane that raised the flag did not execute in the last iteration
        L.h.pc = tlo + pco;
        need_cold = ncv != 0;
      }
    }
    bool wtake = false; Rec wrec; wrec.w0 = 0; wrec.imm = 0; wrec.raw = 0; wrec.pad = 0;
    if (need_cold && (L.h.flags & HF_HOT)) {
      // The warm path (wsp_warm: LDS register files, fflags, plain-storage CSR rows; nothing else of the hart) for a hot hart at
      // an F / D computational instruction with FS already Dirty, a bit-manipulation instruction, mulh* / div* / rem*, or a
      // Zicsr access the path can take
      Rec rec; rec.w0 = 0; rec.imm = 0; rec.raw = 0; rec.pad = 0;
      if (L.hot_fetch(recs, rec)) {
        wrec = rec;
        const u32 wop = rec.w0 & 0xff;
        const bool fsd = (L.h.flags & HF_FS_DIRTY) != 0;
        const bool take = (FP && fsd && wop >= OP_FADD && wop <= OP_FMV_X2F) || wop == OP_ZB || (wop >= OP_MULH && wop <= OP_REMUW && wop != OP_MULW) ||
                          (wop >= OP_CSRRW && wop <= OP_CSRRC) || wop == OP_MRET;
        wtake = take;
      }
    }
    WSP_PROF_ADD(15, WSP_PROF_T() - ph0);     // hot loops + warm-path candidate scan
    const unsigned long long pw0 = WSP_PROF_T();
    if (wtake) {
      const WarmRet w = (TRACE && tr) ? wsp_warm<true>(dp, wrec.w0, wrec.imm, wrec.raw, L.h.pc, L.h.thash, L.h.fcsr, L.h.flags)
                                      : wsp_warm<false>(dp, wrec.w0, wrec.imm, wrec.raw, L.h.pc, L.h.thash, L.h.fcsr, L.h.flags);
      if (w.handled) {
        L.h.pc = w.pc; L.h.exec++; L.h.thash = w.thash; L.h.fcsr = w.fcsr; L.h.flags = w.flags;
        need_cold = false;
      }
    }
    WSP_PROF_ADD(14, WSP_PROF_T() - pw0);     // warm path
    const unsigned long long ps0 = WSP_PROF_T();
    const bool swarm_can = WSP_SWARM && need_cold && live && d.tlb_n && recs &&
                           !(L.h.flags & (HF_HOT | HF_RING | HF_CODE_DIRTY | HF_INH_CY | HF_INH_IR)) && L.h.pc != L.stop_pc;
    if (swarm_can && swarm_skip) --swarm_skip;
    else if (swarm_can) {
      // S / U-mode code under Sv39: the translated warm path (wsp_swarm); a hart it does not take is back at once
      VWarmArgs sa; sa.recs = recs; sa.text_lo = L.text_lo; sa.text_hi = L.text_hi; sa.tohost = L.tohost; sa.stop_pc = L.stop_pc;
      const u64 left = stop_at - L.h.exec;
      const u32 smax = left > 1024 ? 1024u : (u32)left;
      const SWarmRet w = (TRACE && tr) ? wsp_swarm<true>(dp, sa, L.h.pc, L.h.thash, L.h.flags, L.h.fcsr, smax)
                                       : wsp_swarm<false>(dp, sa, L.h.pc, L.h.thash, L.h.flags, L.h.fcsr, smax);
      if (w.n) {
        // a run that ended before its budget ended on an instruction for the general path: it goes there now.  After a long
        // run (code that mostly qualifies: the stop was a TLB refill or a lone CSR access) for that one step; after a short
        // one (trap-heavy code) for a whole burst, whose start-up cost one step would not repay
        L.h.pc = w.pc; L.h.exec += w.n; L.h.thash = w.thash; L.h.fcsr = w.fcsr; need_cold = w.n < smax;
        swarm_credit = w.n >= 16 ? 2u : 0u; swarm_backoff = 0;
      } else if (swarm_credit) --swarm_credit;
      else { swarm_backoff = swarm_backoff < 15 ? 2 * swarm_backoff + 1 : 15u; swarm_skip = swarm_backoff; }   // not its kind of code: try again later
    } else if (need_cold) swarm_credit = 0;
    WSP_PROF_ADD(12, WSP_PROF_T() - ps0);     // translated warm path
    if (d.vreg) {
      // Vector warm path (wsp_vwarm): hot harts with VS Dirty at a vector instruction; the idle lanes l + lpw, l + 2 lpw, ... join
      // hart l as its sub-group and get the instruction and the hart's live registers from its lane.
      Rec rec; rec.w0 = 0; rec.imm = 0; rec.raw = 0; rec.pad = 0;
      bool cand = need_cold && active && (L.h.flags & (HF_HOT | HF_VS_DIRTY | HF_RING)) == (HF_HOT | HF_VS_DIRTY) && L.hot_fetch(recs, rec);
      cand = cand && (rec.w0 & 0xff) >= OP_VEC && (rec.w0 & 0xff) <= OP_VST;
      const unsigned long long cm = __ballot(cand);
      if (cm) {
        const int owner = (int)(lane & (d.lpw - 1));
        const bool helper = lane >= d.lpw && ((cm >> owner) & 1);
        u64 vpc = L.h.pc, vth = L.h.thash; u32 vfcsr = L.h.fcsr, vfl = L.h.flags, vw0 = rec.w0, vraw = rec.raw; i32 vimm = rec.imm;
        const u64 left = stop_at - L.h.exec;
        u32 vmax = left > 64 ? 64u : (u32)left;
#define WSP_FROM_OWNER64(x) do { const u32 lo_ = (u32)__shfl((int)(u32)x, owner, 64), hi_ = (u32)__shfl((int)(u32)(x >> 32), owner, 64); if (helper) x = (u64)lo_ | ((u64)hi_ << 32); } while (0)
#define WSP_FROM_OWNER32(x) do { const u32 v_ = (u32)__shfl((int)x, owner, 64); if (helper) x = v_; } while (0)
        WSP_FROM_OWNER64(vpc); WSP_FROM_OWNER64(vth); WSP_FROM_OWNER32(vfcsr); WSP_FROM_OWNER32(vfl); WSP_FROM_OWNER32(vw0); WSP_FROM_OWNER32(vraw); WSP_FROM_OWNER32(vmax);
        { u32 t_ = (u32)vimm; WSP_FROM_OWNER32(t_); vimm = (i32)t_; }
#undef WSP_FROM_OWNER64
#undef WSP_FROM_OWNER32
        if (cand || helper) {
          // one call site for the hart's lane and its helpers: they execute the call together.  (A helper lane was set up with
          // its hart's slot: program image, text bounds and tohost are the owner's.)
          VWarmArgs va; va.recs = recs; va.text_lo = L.text_lo; va.text_hi = L.text_hi; va.tohost = L.tohost; va.stop_pc = ~0ULL;
          const unsigned long long pv0 = WSP_PROF_T();
          const WarmRet w = wsp_vwarm(dp, va, vw0, vimm, vraw, vpc, vth, vfcsr, vfl, vmax);
          WSP_PROF_ADD(10, WSP_PROF_T() - pv0); WSP_PROF_ADD(9, 1); WSP_PROF_ADD(11, __popcll(__ballot(cand && w.handled)));
          if (cand && w.handled) { L.h.pc += 4 * w.handled; L.h.exec += w.handled; L.h.thash = w.thash; L.h.fcsr = w.fcsr; need_cold = false; }
        }
      }
    }
    if (need_cold) {
      // A cold call is a context switch: every per-lane value of the hot loop is either passed through HotRegs or
      // rebuilt from the kernel arguments afterwards.
      HotRegs hr; hr.pc = L.h.pc; hr.exec = L.h.exec; hr.thash = L.h.thash; hr.stop_at = stop_at; hr.slice_end = slice_end;
      hr.flags = L.h.flags | (swarm_credit ? HR_SWARM_LIVE : 0u); hr.status = L.h.status; hr.fcsr = L.h.fcsr;
      hr = wsp_cold(dp, hr);
      const u32 ln2 = ln_of(d);
      wsp_lane_setup(L, d, blockIdx.x, ln2, blockIdx.x * 64 + ln2, recs);
      L.h.pc = hr.pc; L.h.exec = hr.exec; L.h.thash = hr.thash; stop_at = hr.stop_at; slice_end = hr.slice_end;
      L.h.flags = hr.flags; L.h.status = hr.status; L.h.fcsr = hr.fcsr;
    }
  }

  __syncthreads();
  for (u32 i = lane; i < vwords; i += 64) vhbm[i] = vlds[i];
  if (WSP_VCSR_LDS && d.vreg && active) for (u32 k = 0; k < 4; ++k) d.cold[(size_t)(CI_VSTART + k) * S + slot] = L.m.vcsr_[k];
  if (active) {
    for (u32 r = 1; r < 32; ++r) d.x[(size_t)r * S + slot] = L.m.rx(r);
    if (FP) for (u32 r = 0; r < 32; ++r) d.f[(size_t)r * S + slot] = L.m.rf(r);
    // bring the lazily synchronised counters up to date
    const u64 dlt = L.h.exec - d.hot64[H64_EXEC * S + slot];
    const u64 retired = d.hot64[H64_RETIRED * S + slot] + dlt;
    if (L.h.status == ST_RUNNING && (L.h.exec >= d.max_exec[slot] || retired >= d.max_ret[slot])) L.h.status = ST_STOP_LIMIT;   // Hart.cpp:6025-6026
    d.hot64[H64_CYCLE * S + slot] += dlt; d.hot64[H64_INSTRET * S + slot] += dlt; d.hot64[H64_RETIRED * S + slot] = retired;
    d.hot64[H64_EXEC * S + slot] = L.h.exec; d.hot64[H64_PC * S + slot] = L.h.pc; d.hot64[H64_THASH * S + slot] = L.h.thash;
    d.hot32[H32_FLAGS * S + slot] = L.h.flags; d.hot32[H32_STATUS * S + slot] = L.h.status; d.hot32[H32_FCSR * S + slot] = L.h.fcsr;
  }
  WSP_PROF_ADD(0, WSP_PROF_T() - pk0);
  const unsigned long long still = __ballot(active && L.h.status == ST_RUNNING);
  if (lane == 0 && still) atomicAdd(d.running, (u32)__popcll(still));
}

// ------------------------------------------------------------------------------------------ support kernels
// Pre-decode one program image: thread i decodes the (up to) 4 bytes at byte offset 2*i.
__global__ void wsp_predecode_kernel(const u8* img, u64 n_bytes, u32 isa, Rec* out)
{
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 off = i * 2;
  if (off + 2 > n_bytes) {
    // two sentinel records behind the image: a hart that runs off the end of its text fetches "not pre-decoded"
    if (off < n_bytes + 4) { Rec r = mk(OP_SLOWFETCH, 0, 0, 0, 0); finish_rec(r); out[i] = r; }
    return;
  }
  u32 lo = img[off] | ((u32)img[off + 1] << 8);
  Rec r;
  if ((lo & 3) != 3) r = decode16(lo, isa);
  else if (off + 4 <= n_bytes) r = decode32(lo | ((u32)img[off + 2] << 16) | ((u32)img[off + 3] << 24), isa);
  else { r = mk(OP_SLOWFETCH, 0, 0, 0, 0); r.raw = lo; }   // upper half lies outside the decoded range
  finish_rec(r);
  out[i] = r;
}

// Hot records of one program image from its pre-decoded records (two sentinels included: class 0 = not inline).
__global__ void wsp_hrec_kernel(const Rec* recs, u64 nslots, u32 isa, u64 text_lo, u32 text_len, HRec* out)
{
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nslots + 2) return;
  if (i >= nslots) { HRec h; for (u32 k = 0; k < 8; ++k) h.w[k] = 0; h.w[1] = 2; h.w[4] = 64; h.w[5] = 64; out[i] = h; return; }
  out[i] = mk_hrec(recs[i], isa, text_lo, text_len, (u32)(i * 2));
}

__global__ void wsp_decode_words_kernel(const u32* words, u64 n, u32 isa, Rec* out)
{
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = decode(words[i], isa);
}

// Copy `n` bytes at arena offset `off` from a linear staging buffer into harts [lo, hi).
__global__ void wsp_scatter_kernel(DevState d, u32 lo, u32 hi, u64 off, const u8* src, u64 n)
{
  u64 w0 = off >> 3, w1 = (off + n + 7) >> 3, nw = w1 - w0;
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 total = nw * (hi - lo);
  if (i >= total) return;
  u32 hart = lo + (u32)(i % (hi - lo));            // consecutive threads -> consecutive harts: coalesced stores
  u64 w = w0 + i / (hi - lo);
  u64 bo = w << 3;
  u8* dst = d.arena + arena_off(d, hart, bo);
  if (bo >= off && bo + 8 <= off + n) { u64 v; memcpy(&v, src + (bo - off), 8); *(u64*)dst = v; return; }
  for (u32 k = 0; k < 8; ++k) { u64 a = bo + k; if (a >= off && a < off + n) dst[k] = src[a - off]; }
}

// Gather `n` bytes at arena offset `off` of one hart into a linear buffer.
__global__ void wsp_gather_kernel(DevState d, u32 hart, u64 off, u8* dst, u64 n)
{
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = d.arena[arena_off(d, hart, off + i)];
}

// vals[i] -> x[idx] (res==X) or the dword at arena offset idx (res==MEM) of hart lo+i.
__global__ void wsp_poke_each_kernel(DevState d, u32 lo, u32 hi, u32 res, u64 idx, const u64* vals)
{
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (lo + i >= hi) return;
  u32 hart = lo + i;
  if (res == WSP_RES_X) d.x[(size_t)idx * d.S + slot_of(d, hart)] = (d.isa & ISA_RV32) ? (u64)(i64)(i32)vals[i] : vals[i];
  else *(u64*)(d.arena + arena_off(d, hart, idx)) = vals[i];
}

// Memory digest: per hart, sum over its 8-byte words of mix(word ^ GOLDEN*(index+1)).  A wave's arena block is
// lpw*mem_size contiguous bytes in which word j belongs to lane j % lpw, dword j / lpw (arena_off), so the block is
// streamed linearly - 256 threads x 8 B = 2 KiB per load instruction whatever lpw is - and the partial sums are
// combined per lane.  grid = (waves, chunks), block = 256; out[] must be zeroed first.
#define WSP_MH_THREADS 256
__global__ void __launch_bounds__(WSP_MH_THREADS) wsp_memhash_kernel(DevState d, u64 words_per_chunk, u64* out /*[S]*/)
{
  __shared__ u64 red[WSP_MH_THREADS];
  const u32 t = threadIdx.x, wave = blockIdx.x;
  const u64 total = (d.mem_size >> 3) << d.lpw_shift;                 // words in this wave's block
  u64 j0 = (u64)blockIdx.y * words_per_chunk, j1 = j0 + words_per_chunk;   // words_per_chunk is a multiple of 256
  if (j1 > total) j1 = total;
  const u64* base = (const u64*)(d.arena + (size_t)wave * d.lpw * d.mem_size);
  const u32 sh = d.lpw_shift;
  u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  u64 j = j0 + t;
  for (; j + 3 * WSP_MH_THREADS < j1; j += 4 * WSP_MH_THREADS) {
    const u64 w0 = base[j], w1 = base[j + WSP_MH_THREADS], w2 = base[j + 2 * WSP_MH_THREADS], w3 = base[j + 3 * WSP_MH_THREADS];
    s0 += mix(w0 ^ (GOLDEN * ((j >> sh) + 1)));
    s1 += mix(w1 ^ (GOLDEN * (((j + WSP_MH_THREADS) >> sh) + 1)));
    s2 += mix(w2 ^ (GOLDEN * (((j + 2 * WSP_MH_THREADS) >> sh) + 1)));
    s3 += mix(w3 ^ (GOLDEN * (((j + 3 * WSP_MH_THREADS) >> sh) + 1)));
  }
  for (; j < j1; j += WSP_MH_THREADS) s0 += mix(base[j] ^ (GOLDEN * ((j >> sh) + 1)));
  red[t] = s0 + s1 + s2 + s3;
  __syncthreads();
  // thread t always handled lane t % lpw (chunk starts and the stride are multiples of lpw): fold the halves down to lpw
  for (u32 k = WSP_MH_THREADS / 2; k >= d.lpw; k >>= 1) {
    if (t < k) red[t] += red[t + k];
    __syncthreads();
  }
  if (t < d.lpw && wave * d.lpw + t < d.n_harts) atomicAdd((unsigned long long*)&out[wave * 64 + t], (unsigned long long)red[t]);
}

__global__ void wsp_results_kernel(DevState d, const u64* memhash, wsp_result* out)
{
  u32 hart = blockIdx.x * blockDim.x + threadIdx.x;
  if (hart >= d.n_harts) return;
  const u32 S = d.S, slot = slot_of(d, hart);
  u64 s = 0;
  s = fold(s, d.hot64[H64_PC * S + slot]);
  for (u32 r = 1; r < 32; ++r) s = fold(s, d.x[(size_t)r * S + slot]);
  for (u32 r = 0; r < 32; ++r) s = fold(s, d.f[(size_t)r * S + slot]);
  s = fold(s, d.hot32[H32_PRIV * S + slot]);
  s = fold(s, d.hot32[H32_FCSR * S + slot]);
  const int c1[] = { CI_STVEC, CI_SCOUNTEREN, CI_SENVCFG, CI_SSCRATCH, CI_SEPC, CI_SCAUSE, CI_STVAL, CI_SATP };
  for (int i = 0; i < 8; ++i) s = fold(s, d.cold[(size_t)c1[i] * S + slot]);
  s = fold(s, d.hot64[H64_MSTATUS * S + slot]);
  const int c2[] = { CI_MEDELEG, CI_MIDELEG, CI_MIE, CI_MTVEC, CI_MCOUNTEREN, CI_MENVCFG, CI_MCOUNTINHIBIT, CI_MSCRATCH, CI_MEPC, CI_MCAUSE, CI_MTVAL, CI_MIP };
  for (int i = 0; i < 12; ++i) s = fold(s, d.cold[(size_t)c2[i] * S + slot]);
  s = fold(s, d.hot64[H64_CYCLE * S + slot]);
  s = fold(s, d.hot64[H64_INSTRET * S + slot]);
  s = fold(s, d.hot64[H64_EXEC * S + slot]);          // time CSR == loop iterations (Hart.hpp:3101)
  for (u32 i = 0; i < 72; ++i) s = fold(s, d.cold[(size_t)(CI_PMPCFG0 + i) * S + slot]);   // pmpcfg0,2,..14 then pmpaddr0..63
  if (d.isa & ISA_S) {                                  // TLB contents are architecturally visible (stale entries): part of the state
    const u64* t = d.tlb + (size_t)hart * 2 * d.tlb_n; u64 th = 0;
    for (u32 i = 0; i < 2 * d.tlb_n; ++i) th = fold(th, t[i]);
    s = fold(s, th);
  }
  if (d.isa & ISA_V) {
    s = fold(s, d.cold[(size_t)CI_VSTART * S + slot]); s = fold(s, d.cold[(size_t)CI_VCSR * S + slot]);
    s = fold(s, d.cold[(size_t)CI_VL * S + slot]); s = fold(s, d.cold[(size_t)CI_VTYPE * S + slot]);
    const u64* v = (const u64*)(d.vreg + (size_t)hart * 32 * d.vlenb);
    for (u32 i = 0; i < 32 * d.vlenb / 8; ++i) s = fold(s, v[i]);
  }
  wsp_result r;
  r.status = d.hot32[H32_STATUS * S + slot]; r.priv = d.hot32[H32_PRIV * S + slot];
  r.stop_value = d.hot64[H64_STOP_VALUE * S + slot]; r.pc = d.hot64[H64_PC * S + slot];
  r.exec_count = d.hot64[H64_EXEC * S + slot]; r.retired_count = d.hot64[H64_RETIRED * S + slot];
  r.state_hash = s; r.mem_hash = memhash[slot]; r.trace_hash = d.hot64[H64_THASH * S + slot];
  out[hart] = r;
}

// Device softfloat sweep (diagnostic surface, wsp_sf_eval): the gfx950 build of wsp_softfloat.h - the schoolbook 128 / 64
// division, __umul64hi products and clz paths that a host build of the header replaces with compiler builtins.
__global__ void wsp_sf_eval_kernel(u32 op, int rm, const u64* a, const u64* b, const u64* c, u64 n, u64* out, u32* flags)
{
  using namespace wsf;
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool dbl = (op & 0x100) != 0; const u32 k = op & 0xff;
  const u64 x = a[i], y = b[i], z = c[i];
  unsigned fl = 0; u64 r = 0;
  if ((op & 0x200) && dbl && rm == RM_RNE) {
    // the FP unit's paths (hw_* in wsp_softfloat.h) where they apply; flags bit 31 tells the caller that one did
    bool hw = false;
    switch (k) {
      case 0: hw = hw_add64(x, y, r, fl); break;
      case 1: hw = hw_add64(x, y ^ F64::SIGN, r, fl); break;
      case 2: hw = hw_mul64(x, y, r, fl); break;
      case 3: hw = hw_div64(x, y, r, fl); break;
      case 5: hw = hw_fma64(x, y, z, r, fl); break;
      case 10: case 11: case 12: case 13: r = hw_int_to_f64(z, (k - 10) & 1, ((k - 10) & 2) != 0, fl); hw = true; break;
      default: break;
    }
    if (hw) { out[i] = r; flags[i] = fl | 0x80000000u; return; }
  }
  if ((op & 0x200) && !dbl && rm == RM_RNE) {                // the same for binary32
    const u64 x32 = x & 0xffffffffULL, y32 = y & 0xffffffffULL, z32 = z & 0xffffffffULL;
    bool hw = false;
    switch (k) {
      case 0: hw = hw_add32(x32, y32, r, fl); break;
      case 1: hw = hw_add32(x32, y32 ^ F32::SIGN, r, fl); break;
      case 2: hw = hw_mul32(x32, y32, r, fl); break;
      case 3: hw = hw_div32(x32, y32, r, fl); break;
      case 5: hw = hw_fma32(x32, y32, z32, r, fl); break;
      case 10: case 11: r = hw_int_to_f32(z, (k - 10) & 1, fl); hw = true; break;
      default: break;
    }
    if (hw) { out[i] = r; flags[i] = fl | 0x80000000u; return; }
  }
  switch (k) {
    case 0: r = dbl ? f_add<F64>(x, y, rm, fl) : f_add<F32>(x & 0xffffffffULL, y & 0xffffffffULL, rm, fl); break;
    case 1: r = dbl ? f_sub<F64>(x, y, rm, fl) : f_sub<F32>(x & 0xffffffffULL, y & 0xffffffffULL, rm, fl); break;
    case 2: r = dbl ? f_mul<F64>(x, y, rm, fl) : f_mul<F32>(x & 0xffffffffULL, y & 0xffffffffULL, rm, fl); break;
    case 3: r = dbl ? f_div<F64>(x, y, rm, fl) : f_div<F32>(x & 0xffffffffULL, y & 0xffffffffULL, rm, fl); break;
    case 4: r = dbl ? f_sqrt<F64>(x, rm, fl) : f_sqrt<F32>(x & 0xffffffffULL, rm, fl); break;
    case 5: r = dbl ? f_fma<F64>(x, y, z, rm, fl) : f_fma<F32>(x & 0xffffffffULL, y & 0xffffffffULL, z & 0xffffffffULL, rm, fl); break;
    case 6: case 7: case 8: case 9: {
      const bool sg = (k - 6) & 1, w = ((k - 6) & 2) != 0;
      r = dbl ? f_to_int<F64>(x, rm, sg, w, fl) : f_to_int<F32>(x & 0xffffffffULL, rm, sg, w, fl); break; }
    case 10: case 11: case 12: case 13: {
      const bool sg = (k - 10) & 1, w = ((k - 10) & 2) != 0;
      r = dbl ? int_to_f<F64>(z, sg, w, rm, fl) : int_to_f<F32>(z, sg, w, rm, fl); break; }
    default: r = dbl ? f64_to_f32(x, rm, fl) : f32_to_f64(x & 0xffffffffULL, fl); break;
  }
  out[i] = r; flags[i] = fl;
}

// Scratch of wsp_batch_results_noalloc: the per-slot partial sums of the memory digest.  One copy per GPU the library runs on,
// held until the process ends (8 MiB); its device address is looked up once per device (scratch_of) and only that pointer is
// ever handed to a kernel or a memset - the array's name is a host-side stand-in, not a device address.
#define WSP_SCRATCH_SLOTS (1u << 20)
static __device__ __align__(256) unsigned long long wsp_scratch[WSP_SCRATCH_SLOTS];

// ------------------------------------------------------------------------------------------ host side
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPOK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(WSP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" const char* wsp_last_error(void) { return g_err.c_str(); }
extern "C" const char* wsp_version(void) { return "whisper_amd 0.1 (gfx950)"; }

// Isa::applyIsaString (Isa.cpp:561) for the letters that matter to the hot path.
static int parse_isa(const char* isa, u32* out, std::string* why)
{
  std::string s = isa ? isa : "";
  for (auto& c : s) c = (char)tolower(c);
  if (s.rfind("rv64", 0) == 0) s = s.substr(4);
  u32 f = 0;
  if (s.rfind("rv32", 0) == 0) { s = s.substr(4); f |= ISA_RV32; }      // (wsp_batch_create decides the width: descriptor, config, then this prefix)
  size_t us = s.find('_');
  std::string base = s.substr(0, us);
  for (char c : base) {
    switch (c) {
      case 'i': break;
      case 'm': f |= ISA_M; break; case 'a': f |= ISA_A; break; case 'f': f |= ISA_F; break; case 'd': f |= ISA_D; break;
      case 'c': f |= ISA_C; break; case 's': f |= ISA_S; break; case 'u': f |= ISA_U; break; case 'v': f |= ISA_V; break;
      case 'g': f |= ISA_M | ISA_A | ISA_F | ISA_D; break;
      default: if (why) *why = std::string("ISA letter '") + c + "' is not supported"; return WSP_EUNSUPPORTED;
    }
  }
  while (us != std::string::npos) {
    size_t nx = s.find('_', us + 1);
    std::string e = s.substr(us + 1, nx == std::string::npos ? std::string::npos : nx - us - 1);
    if (e == "zicsr" || e == "zifencei" || e.empty()) {}
    else if (e == "zicntr") f |= ISA_ZICNTR;
    else if (e == "zca") f |= ISA_C;
    else if (e == "zba") f |= ISA_ZBA; else if (e == "zbb") f |= ISA_ZBB; else if (e == "zbs") f |= ISA_ZBS;
    else if (e == "zbc") f |= ISA_ZBC; else if (e == "zicond") f |= ISA_ZICOND;
    else { if (why) *why = "ISA extension '" + e + "' is not supported"; return WSP_EUNSUPPORTED; }
    us = nx;
  }
  if ((f & ISA_D) && !(f & ISA_F)) f &= ~(u32)ISA_D;              // Hart.cpp:585-595
  if ((f & ISA_S) && !(f & ISA_U)) { if (why) *why = "S without U in the ISA string"; return WSP_EINVAL; }   // Isa.cpp:655-660
  *out = f;
  return WSP_OK;
}

extern "C" uint32_t wsp_isa_flags(const char* isa) { u32 f = 0; std::string w; if (parse_isa(isa, &f, &w) != WSP_OK) { g_err = w; return 0xffffffffu; } return f; }

// Device address of wsp_scratch on each GPU, looked up with that GPU current (each GPU holds its own copy at its own address).
static u64* g_scratch_ptr[64];
static std::mutex g_scratch_mu;                        // batches may be created from several host threads
static int scratch_of(int dev, u64** out)
{
  if (dev < 0 || dev >= 64) return fail(WSP_ENODEV, "device ordinal out of range for the static scratch table");
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    HIPOK(hipSetDevice(dev));
    HIPOK(hipGetSymbolAddress(&p, HIP_SYMBOL(wsp_scratch)));
    g_scratch_ptr[dev] = (u64*)p;
  }
  *out = g_scratch_ptr[dev];
  return WSP_OK;
}

struct Program {
  u64 text_lo = 0, text_hi = 0;
  Rec* d_recs = nullptr;
  HRec* d_hrecs = nullptr;             // hot records (wsp_conv_core.h); null for single-hart programs loaded in bulk
  bool pooled = false;                 // d_recs points into one of the batch's rec_pools (wsp_batch_load_elf_many)
  std::map<std::string, u64> symbols;
};

struct wsp_batch {
  DevState d{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<Program> progs;
  std::vector<u32> hart_prog;          // host copy, index = hart
  std::vector<u64> hart_stop_pc;       // host copy of the stop addresses (~0: none)
  std::vector<u32> snap_hart_prog; std::vector<u64> snap_hart_stop_pc;   // the two host copies as of wsp_batch_snapshot
  std::vector<Rec*> rec_pools;         // record allocations shared by many programs
  ProgDev* d_progs = nullptr; size_t d_progs_cap = 0;
  DevState* d_state = nullptr;
  u64* d_memhash = nullptr; wsp_result* d_results = nullptr;
  double last_kernel_s = 0, last_hash_s = 0; u32 last_launches = 0; u64 last_hash_bytes = 0;
  bool has_fp = false;
  bool abi_names = false;              // config tag abi_names: the text log prints ABI register names
#define WSP_SNAP_PARTS 13
  void* snap[WSP_SNAP_PARTS] = {};
};

struct SnapPart { void* live; size_t bytes; };
static void snap_parts(wsp_batch* b, SnapPart out[WSP_SNAP_PARTS])
{
  DevState& d = b->d; size_t S = d.S;
  out[0] = { d.x, 32 * S * 8 }; out[1] = { d.f, 32 * S * 8 }; out[2] = { d.hot64, H64_COUNT * S * 8 }; out[3] = { d.hot32, H32_COUNT * S * 4 };
  out[4] = { d.cold, CI_COUNT * S * 8 }; out[5] = { d.tohost, S * 8 }; out[6] = { d.max_exec, S * 8 }; out[7] = { d.prog, S * 4 };
  out[8] = { d.arena, (size_t)d.n_waves * d.lpw * d.mem_size };
  out[9] = { d.tlb, (size_t)d.n_harts * 2 * d.tlb_n * 8 };
  out[10] = { d.vreg, (size_t)d.n_harts * 32 * d.vlenb };
  out[11] = { d.max_ret, S * 8 }; out[12] = { d.stop_pc, S * 8 };
}

static int upload_progs(wsp_batch* b)
{
  std::vector<ProgDev> v(b->progs.size());
  for (size_t i = 0; i < v.size(); ++i) { v[i].text_lo = b->progs[i].text_lo; v[i].text_hi = b->progs[i].text_hi; v[i].recs = b->progs[i].d_recs; v[i].hrecs = b->progs[i].d_hrecs; }
  if (v.size() > b->d_progs_cap) {
    if (b->d_progs) HIPOK(hipFree(b->d_progs));
    b->d_progs_cap = std::max<size_t>(16, v.size() * 2);
    HIPOK(hipMalloc(&b->d_progs, b->d_progs_cap * sizeof(ProgDev)));
    b->d.progs = b->d_progs;
  }
  if (!v.empty()) HIPOK(hipMemcpy(b->d_progs, v.data(), v.size() * sizeof(ProgDev), hipMemcpyHostToDevice));
  return WSP_OK;
}

static int csr_loc(u32 num, u32* ix);

// Reset value of a modelled CSR for every hart of a fresh batch (all harts are in the reset state: legalisation against
// the previous value uses the built-in reset values).
static int reset_csr_all(wsp_batch* b, u32 num, u64 v)
{
  DevState& d = b->d; const size_t S = d.S;
  u32 ix = 0; const int loc = csr_loc(num, &ix);
  if (loc < 0) return fail(WSP_EUNSUPPORTED, "config csr reset: only CSRs with storage of their own can be given a reset value (not sstatus / sie / sip / fflags / frm / vxsat / vxrm views)");
  if (loc == 1) { std::vector<u32> r(S, (u32)v & 0xff); HIPOK(hipMemcpy(d.hot32 + ix * S, r.data(), S * 4, hipMemcpyHostToDevice)); return WSP_OK; }
  if (num == C_MSTATUS) v = (MSTATUS_RESET & ~MSTATUS_WMASK) | (v & (MSTATUS_WMASK | MST_SD));
  if (loc == 2) {
    if (ix >= CI_PMPCFG0 && ix < CI_PMPADDR0) v = Lane<DevMem>::legalize_pmpcfg(0, v) & 0x9f9f9f9f9f9f9f9fULL;
    else if (ix >= CI_PMPADDR0 && ix < CI_PMPADDR0 + 64) v &= 0x003fffffffffffffULL;
    if (num == C_SATP && (v >> 60) != 0 && (v >> 60) != 8) return fail(WSP_EUNSUPPORTED, "config csr reset: satp mode other than bare / sv39");
    if (num == C_MCOUNTINHIBIT) {
      v &= 0xfffffffdULL;
      std::vector<u32> fl(S, ((v & 1) ? (u32)HF_INH_CY : 0) | ((v & 4) ? (u32)HF_INH_IR : 0));
      HIPOK(hipMemcpy(d.hot32 + (size_t)H32_FLAGS * S, fl.data(), S * 4, hipMemcpyHostToDevice));
    }
  }
  std::vector<u64> r(S, v);
  HIPOK(hipMemcpy((loc == 0 ? d.hot64 : d.cold) + ix * S, r.data(), S * 8, hipMemcpyHostToDevice));
  return WSP_OK;
}

extern "C" void wsp_batch_destroy(wsp_batch* b)
{
  if (!b) return;
  hipSetDevice(b->device);
  for (auto& p : b->progs) { if (p.d_recs && !p.pooled) hipFree(p.d_recs); if (p.d_hrecs) hipFree(p.d_hrecs); }
  for (Rec* p : b->rec_pools) hipFree(p);
  void* ptrs[] = { b->d.x, b->d.f, b->d.hot64, b->d.hot32, b->d.cold, b->d.tlb, b->d.vreg, b->d.tohost, b->d.max_exec, b->d.prog, b->d.arena, b->d.running, b->d_progs, b->d_memhash, b->d_results, b->d_state, b->d.ring, b->d.ring_pos, b->d.max_ret, b->d.stop_pc };
  for (void* p : ptrs) if (p) hipFree(p);
  for (void* p : b->snap) if (p) hipFree(p);
  if (b->ev0) hipEventDestroy(b->ev0);
  if (b->ev1) hipEventDestroy(b->ev1);
  if (b->stream) hipStreamDestroy(b->stream);
  delete b;
}

extern "C" int wsp_batch_create(const wsp_batch_desc* desc, wsp_batch** out)
{
  if (!desc || !out) return fail(WSP_EINVAL, "null argument");
  *out = nullptr;
  // Configuration file text first; explicit descriptor fields override it (as command line options override the file,
  // Session.cpp:1373-1424, 1440-1520).
  wspcfg::Parsed cfg;
  if (desc->config_json) {
    wspcfg::JVal root; std::string perr;
    if (!wspcfg::parse_json(desc->config_json, root, perr)) return fail(WSP_EFORMAT, perr);
    int crc = wspcfg::apply(root, cfg, perr);
    if (crc != WSP_OK) return fail(crc, perr);
  }
  const char* isa_str = desc->isa ? desc->isa : (cfg.isa.empty() ? nullptr : cfg.isa.c_str());
  if (!isa_str) return fail(WSP_EINVAL, "no ISA string (descriptor isa or config tag \"isa\")");
  u32 xlen = desc->xlen ? desc->xlen : cfg.xlen;
  if (!xlen) { std::string t = isa_str; xlen = (t.rfind("rv32", 0) == 0 || t.rfind("RV32", 0) == 0) ? 32 : 64; }   // HartConfig::getXlen (HartConfig.cpp:3431-3459)
  if (xlen != 64 && xlen != 32) return fail(WSP_EUNSUPPORTED, "xlen must be 32 or 64");
  const u64 mem_size = desc->mem_size ? desc->mem_size : cfg.mem_size;
  if (desc->n_harts == 0) return fail(WSP_EINVAL, "n_harts must be > 0");
  if (mem_size == 0 || (mem_size & 4095)) return fail(WSP_EINVAL, "mem_size must be a non-zero multiple of 4096");
  if (desc->mem_base + mem_size < desc->mem_base) return fail(WSP_EINVAL, "arena wraps the address space");
  u32 isa = 0; std::string why;
  int rc = parse_isa(isa_str, &isa, &why);
  if (rc != WSP_OK) return fail(rc, why);
  isa |= cfg.features;
  isa = xlen == 32 ? (isa | ISA_RV32) : (isa & ~(u32)ISA_RV32);
  if (xlen == 32) {     // System<uint32_t>: modelled for M / U mode integer, A, F, D, C code
    if (isa & (ISA_S | ISA_V)) return fail(WSP_EUNSUPPORTED, "xlen=32 is modelled without S mode (Sv32) and without V");
    if (desc->mem_base + mem_size > (1ULL << 32)) return fail(WSP_EINVAL, "an RV32 arena must lie below 4 GiB");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(WSP_ENODEV, "no HIP device: whisper_amd has no CPU execution path");
  int dev = desc->device;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= ndev) return fail(WSP_ENODEV, "device ordinal out of range");
  HIPOK(hipSetDevice(dev));
  u32 lpw = desc->lanes_per_wave;
  if (lpw == 0) {
    // One wave's step takes about the same time however many of its lanes carry harts (harts of one program execute
    // together in the converged loop), and one wave per SIMD already hides most of what can be hidden: the fewest waves that
    // still give every SIMD one - 1024 - is the fastest layout measured (profiles/r02/: 4096 harts lpw 1/2/4/8 -> 9.5 / 11.9 /
    // 13.5 / 13.7 GIPS, 16384 harts lpw 4/16/64 -> 36 / 55 / 57, 65536 harts lpw 16/64 -> 140 / 207).
    lpw = 1;
    while (lpw < 64 && (desc->n_harts + lpw - 1) / lpw > 1024u * WSP_WAVES_PER_EU) lpw <<= 1;
    // V batches keep lanes free for the vector unit's element loops (the hart's sub-group, wsp_vwarm): at least 8 lanes per hart at
    // VLEN 512 and above, 4 below (measured at 16384 harts, one wave per SIMD: vector kernels at VLEN 512 0.42 / 0.63 / 0.43 GIPS
    // with 4 / 8 / 16 harts per wave, at VLEN 256 0.69 / 1.07 / 1.46 / 0.91 with 4 / 8 / 16 / 32)
    const u32 vb0 = desc->vlen_bytes ? desc->vlen_bytes : cfg.vlen_bytes ? cfg.vlen_bytes : 64u;
    const u32 vcap = vb0 >= 64 ? 8u : 16u;
    if ((isa & ISA_V) && lpw > vcap) lpw = vcap;
  }
  if (lpw == 0 || lpw > 64 || (lpw & (lpw - 1))) return fail(WSP_EINVAL, "lanes_per_wave must be 0 or a power of two <= 64");
  if (isa & ISA_V) {
    // the vector register files of a wave's harts are staged in LDS (wsp_vreg_lds): the wave's block must fit the 64 KiB a
    // workgroup gets without asking for more; a larger request is lowered (16 at VLEN 512, 32 at VLEN 256 / 128)
    const u32 vb = desc->vlen_bytes ? desc->vlen_bytes : cfg.vlen_bytes ? cfg.vlen_bytes : 64u;
    while (lpw > 1 && (size_t)lpw * (2 * WSP_XSLOTS * 8 + 32 * vb + 32) > 65536) lpw >>= 1;
  }
  wsp_batch* b = new wsp_batch();
  b->device = dev;
  DevState& d = b->d;
  d.n_harts = desc->n_harts; d.lpw = lpw; d.lpw_shift = (u32)__builtin_ctz(lpw);
  d.n_waves = (desc->n_harts + lpw - 1) / lpw; d.S = d.n_waves * 64; d.isa = isa; d.hart_id_base = desc->hart_id_base;
  d.mem_base = desc->mem_base; d.mem_size = mem_size;
  b->abi_names = cfg.abi_names;
  const u32 tlb_req = desc->tlb_entries ? desc->tlb_entries : cfg.tlb_entries, vlen_req = desc->vlen_bytes ? desc->vlen_bytes : cfg.vlen_bytes;
  d.tlb_n = (isa & ISA_S) ? (tlb_req ? tlb_req : 2048u) : 0u;        // Hart.cpp:106
  if (d.tlb_n & (d.tlb_n - 1)) { delete b; return fail(WSP_EINVAL, "tlb_entries must be a power of two"); }   // Tlb.cpp:11
  d.vlenb = (isa & ISA_V) ? (vlen_req ? vlen_req : 64u) : 0u;           // Hart.cpp:1081
  if ((isa & ISA_V) && d.vlenb != 16 && d.vlenb != 32 && d.vlenb != 64) { delete b; return fail(WSP_EINVAL, "vlen_bytes must be 16, 32 or 64"); }
  b->has_fp = (isa & ISA_F) != 0;
  size_t vreg_bytes = (size_t)d.n_harts * 32 * d.vlenb;
  size_t S = d.S, arena_bytes = (size_t)d.n_waves * lpw * d.mem_size, tlb_bytes = (size_t)d.n_harts * 2 * d.tlb_n * 8;
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  if (arena_bytes + tlb_bytes + vreg_bytes + S * 1024 > free_b) { delete b; return fail(WSP_ENOMEM, "batch does not fit device memory"); }
#define ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void**)&(ptr), (bytes)); if (e_ != hipSuccess) { std::string m_ = std::string("hipMalloc ") + #ptr + ": " + hipGetErrorString(e_); wsp_batch_destroy(b); return fail(WSP_ENOMEM, m_); } \
    e_ = hipMemset((ptr), 0, (bytes)); if (e_ != hipSuccess) { wsp_batch_destroy(b); return fail(WSP_EHIP, "hipMemset failed"); } } while (0)
  ALLOC(d.x, 32 * S * 8); ALLOC(d.f, 32 * S * 8); ALLOC(d.hot64, H64_COUNT * S * 8); ALLOC(d.hot32, H32_COUNT * S * 4);
  ALLOC(d.cold, (CI_COUNT + 1) * S * 8);   /* + one row: per-slot store sink of the hot loop */ ALLOC(d.tohost, S * 8); ALLOC(d.max_exec, S * 8); ALLOC(d.prog, S * 4); ALLOC(d.max_ret, S * 8); ALLOC(d.stop_pc, S * 8);
  ALLOC(d.arena, arena_bytes); if (tlb_bytes) ALLOC(d.tlb, tlb_bytes); if (vreg_bytes) ALLOC(d.vreg, vreg_bytes); ALLOC(d.running, 4); ALLOC(b->d_state, sizeof(DevState)); ALLOC(b->d_memhash, S * 8); ALLOC(b->d_results, (size_t)d.n_harts * sizeof(wsp_result));
#undef ALLOC
  // Hart::reset (Hart.cpp:942-1070): pc = reset pc (arena base here), M mode, mstatus reset value.
  std::vector<u64> v64(S);
  std::fill(v64.begin(), v64.end(), cfg.has_reset_pc ? (cfg.reset_pc & ~1ULL) : d.mem_base);   // defineResetPc (config tag reset_vec), else the arena base
  hipMemcpy(d.hot64 + (size_t)H64_PC * S, v64.data(), S * 8, hipMemcpyHostToDevice);
  std::fill(v64.begin(), v64.end(), xlen == 32 ? (3ULL << MST_MPP_SH) : MSTATUS_RESET);   // RV32: no SXL / UXL fields (mstatus = 0x1800)
  hipMemcpy(d.hot64 + (size_t)H64_MSTATUS * S, v64.data(), S * 8, hipMemcpyHostToDevice);
  std::fill(v64.begin(), v64.end(), 1ULL << 63);      hipMemcpy(d.cold + (size_t)CI_VTYPE * S, v64.data(), S * 8, hipMemcpyHostToDevice);   // vtype resets to vill (CsRegs.cpp:5363)
  std::fill(v64.begin(), v64.end(), ~0ULL);           hipMemcpy(d.tohost, v64.data(), S * 8, hipMemcpyHostToDevice);
  hipMemcpy(d.max_exec, v64.data(), S * 8, hipMemcpyHostToDevice);
  hipMemcpy(d.max_ret, v64.data(), S * 8, hipMemcpyHostToDevice); hipMemcpy(d.stop_pc, v64.data(), S * 8, hipMemcpyHostToDevice);
  std::vector<u32> v32(S, (u32)PRV_M);                hipMemcpy(d.hot32 + (size_t)H32_PRIV * S, v32.data(), S * 4, hipMemcpyHostToDevice);
  std::fill(v32.begin(), v32.end(), ~0u);             hipMemcpy(d.prog, v32.data(), S * 4, hipMemcpyHostToDevice);
  b->hart_prog.assign(d.n_harts, ~0u); b->hart_stop_pc.assign(d.n_harts, ~0ULL);
  { u64* sp = nullptr; if (scratch_of(dev, &sp) != WSP_OK) { std::string m_ = g_err; wsp_batch_destroy(b); return fail(WSP_EHIP, m_); } }   // set-up time: wsp_batch_results_noalloc must not look anything up
  if (hipStreamCreate(&b->stream) != hipSuccess || hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess) {
    wsp_batch_destroy(b); return fail(WSP_EHIP, "stream/event creation failed");
  }
  for (auto& cr : cfg.csr) {                            // config "csr": { name: { "reset": v } } (applyCsrConfig, HartConfig.cpp:243-330)
    int crc = reset_csr_all(b, cr.num, cr.value);
    if (crc != WSP_OK) { std::string m_ = g_err; wsp_batch_destroy(b); return fail(crc, m_); }
  }
  *out = b;
  return WSP_OK;
}

extern "C" int wsp_config_parse(const char* json_text, wsp_config_info* out)
{
  if (!json_text || !out) return fail(WSP_EINVAL, "null argument");
  memset(out, 0, sizeof *out);
  wspcfg::JVal root; wspcfg::Parsed cfg; std::string perr;
  if (!wspcfg::parse_json(json_text, root, perr)) return fail(WSP_EFORMAT, perr);
  int rc = wspcfg::apply(root, cfg, perr);
  if (rc != WSP_OK) return fail(rc, perr);
  out->xlen = cfg.xlen; out->features = cfg.features; out->mem_size = cfg.mem_size; out->reset_pc = cfg.reset_pc; out->has_reset_pc = cfg.has_reset_pc;
  out->tlb_entries = cfg.tlb_entries; out->vlen_bytes = cfg.vlen_bytes; out->abi_names = cfg.abi_names;
  out->n_csr_resets = (u32)std::min<size_t>(cfg.csr.size(), 32);
  for (u32 i = 0; i < out->n_csr_resets; ++i) { out->csr_num[i] = cfg.csr[i].num; out->csr_value[i] = cfg.csr[i].value; }
  snprintf(out->isa, sizeof out->isa, "%s", cfg.isa.c_str());
  return WSP_OK;
}

// Device staging buffer that is released on every return path.
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
  template <class T> T* as() const { return (T*)p; }
};

static int check_range(wsp_batch* b, u32 lo, u32 hi)
{
  if (!b) return fail(WSP_EINVAL, "null batch");
  if (lo >= hi || hi > b->d.n_harts) return fail(WSP_ERANGE, "hart range out of bounds");
  return WSP_OK;
}

static int write_arena(wsp_batch* b, u32 lo, u32 hi, u64 addr, const void* bytes, size_t n)
{
  DevState& d = b->d;
  if (n == 0) return WSP_OK;
  if (addr < d.mem_base || addr - d.mem_base > d.mem_size - n || n > d.mem_size) return fail(WSP_ERANGE, "image outside the hart arena");
  DevBuf stage;
  HIPOK(stage.alloc(n));
  HIPOK(hipMemcpy(stage.p, bytes, n, hipMemcpyHostToDevice));
  u64 off = addr - d.mem_base, nw = ((off + n + 7) >> 3) - (off >> 3), total = nw * (hi - lo);
  hipLaunchKernelGGL(wsp_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, b->stream, d, lo, hi, off, stage.as<u8>(), (u64)n);
  HIPOK(hipGetLastError());
  HIPOK(hipStreamSynchronize(b->stream));
  return WSP_OK;
}

static int set_u32_range(wsp_batch* b, u32* base, u32 lo, u32 hi, u32 val, u32 or_mask = 0, bool do_or = false)
{
  // per-hart scalars are few: update through the host (setup path, not timed)
  DevState& d = b->d;
  for (u32 h = lo; h < hi; ++h) {
    u32 s = slot_of(d, h), v = val;
    if (do_or) { HIPOK(hipMemcpy(&v, base + s, 4, hipMemcpyDeviceToHost)); v |= or_mask; }
    HIPOK(hipMemcpy(base + s, &v, 4, hipMemcpyHostToDevice));
  }
  return WSP_OK;
}

static int set_u64_slots(wsp_batch* b, u64* base, u32 lo, u32 hi, u64 val)
{
  DevState& d = b->d;
  // contiguous runs of slots inside a wave
  u32 h = lo;
  std::vector<u64> buf(64, val);
  while (h < hi) {
    u32 w = h >> d.lpw_shift, l = h & (d.lpw - 1);
    u32 cnt = std::min(hi - h, d.lpw - l);
    HIPOK(hipMemcpy(base + (size_t)w * 64 + l, buf.data(), (size_t)cnt * 8, hipMemcpyHostToDevice));
    h += cnt;
  }
  return WSP_OK;
}

static int set_u32_slots(wsp_batch* b, u32* base, u32 lo, u32 hi, u32 val)
{
  DevState& d = b->d;
  u32 h = lo;
  std::vector<u32> buf(64, val);
  while (h < hi) {
    u32 w = h >> d.lpw_shift, l = h & (d.lpw - 1);
    u32 cnt = std::min(hi - h, d.lpw - l);
    HIPOK(hipMemcpy(base + (size_t)w * 64 + l, buf.data(), (size_t)cnt * 4, hipMemcpyHostToDevice));
    h += cnt;
  }
  return WSP_OK;
}

// Hot records for a program whose harts may share waves (the hand-written converged loop reads them).
static int build_hrecs(wsp_batch* b, Program& p)
{
  if (!p.d_recs || p.d_hrecs) return WSP_OK;
  const u64 nslots = (p.text_hi - p.text_lo) / 2;
  if (nslots >= (1u << 27)) return WSP_OK;             // the loop's record offsets are 32-bit: larger images run the C++ loop
  HRec* h = nullptr;
  HIPOK(hipMalloc(&h, (nslots + 2) * sizeof(HRec)));
  hipLaunchKernelGGL(wsp_hrec_kernel, dim3((unsigned)((nslots + 2 + 255) / 256)), dim3(256), 0, b->stream, (const Rec*)p.d_recs, nslots, b->d.isa,
                     p.text_lo, (u32)(p.text_hi - p.text_lo), h);
  hipError_t e = hipGetLastError(); if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e != hipSuccess) { hipFree(h); return fail(WSP_EHIP, std::string("hot records: ") + hipGetErrorString(e)); }
  p.d_hrecs = h;
  return WSP_OK;
}

// Turn the pre-decoded record at `stop_pc` of program `pg` into OP_STOPCHK (see wsp_batch_set_stop).
static int patch_stop(wsp_batch* b, u32 pg, u64 stop_pc)
{
  Program& p = b->progs[pg];
  if (!p.d_recs || stop_pc < p.text_lo || stop_pc >= p.text_hi) return WSP_OK;
  u32* w0p = &p.d_recs[(stop_pc - p.text_lo) / 2].w0; u32 w0 = 0;
  HIPOK(hipMemcpy(&w0, w0p, 4, hipMemcpyDeviceToHost));
  if ((w0 & 0xff) == OP_SLOWFETCH || (w0 & 0xff) == OP_STOPCHK) return WSP_OK;
  w0 = (w0 & ~0xffu) | OP_STOPCHK;
  HIPOK(hipMemcpy(w0p, &w0, 4, hipMemcpyHostToDevice));
  // ... and out of the hand-written loop's inline subset too (handler offset 0 = not inline)
  if (p.d_hrecs) { const u32 zero = 0; HIPOK(hipMemcpy(&p.d_hrecs[(stop_pc - p.text_lo) / 2].w[0], &zero, 4, hipMemcpyHostToDevice)); }
  return WSP_OK;
}

// Register a code region: upload, pre-decode on the device, bind to harts [lo, hi).
static int add_program(wsp_batch* b, u32 lo, u32 hi, u64 text_lo, const std::vector<u8>& text, const std::map<std::string, u64>& syms)
{
  DevState& d = b->d;
  Program p; p.symbols = syms;
  if (!text.empty()) {
    p.text_lo = text_lo; p.text_hi = text_lo + (text.size() & ~(size_t)1);
    u64 nslots = (p.text_hi - p.text_lo) / 2;
    if (nslots >= (1u << 29)) return fail(WSP_ERANGE, "code image of 1 GiB or more");
    DevBuf dimg;
    HIPOK(dimg.alloc(text.size()));
    HIPOK(hipMemcpy(dimg.p, text.data(), text.size(), hipMemcpyHostToDevice));
    DevBuf recs;
    HIPOK(recs.alloc((nslots + 2) * sizeof(Rec)));
    hipLaunchKernelGGL(wsp_predecode_kernel, dim3((unsigned)((nslots + 2 + 255) / 256)), dim3(256), 0, b->stream, dimg.as<u8>(), (u64)(p.text_hi - p.text_lo), d.isa, recs.as<Rec>());
    HIPOK(hipGetLastError());
    HIPOK(hipStreamSynchronize(b->stream));
    p.d_recs = recs.as<Rec>(); recs.p = nullptr;       // owned by the batch from here on
    if (hi - lo > 1) { int rc = build_hrecs(b, p); if (rc) return rc; }
  }
  b->progs.push_back(p);
  u32 id = (u32)b->progs.size() - 1;
  int rc = upload_progs(b); if (rc) return rc;
  rc = set_u32_slots(b, d.prog, lo, hi, id); if (rc) return rc;
  for (u32 h = lo; h < hi; ++h) b->hart_prog[h] = id;
  u64 last = ~0ULL;                                    // stop addresses set before the code was loaded
  for (u32 h = lo; h < hi; ++h) if (b->hart_stop_pc[h] != last) { last = b->hart_stop_pc[h]; if (last != ~0ULL) { rc = patch_stop(b, id, last); if (rc) return rc; } }
  return WSP_OK;
}

extern "C" int wsp_batch_load_image(wsp_batch* b, uint32_t lo, uint32_t hi, uint64_t addr, const void* bytes, size_t n, uint32_t flags)
{
  int rc = check_range(b, lo, hi); if (rc) return rc;
  if (!bytes && n) return fail(WSP_EINVAL, "null image");
  HIPOK(hipSetDevice(b->device));
  rc = write_arena(b, lo, hi, addr, bytes, n); if (rc) return rc;
  if (flags & WSP_LOAD_EXEC) {
    if (addr & 1) return fail(WSP_EINVAL, "code must be 2-byte aligned");
    std::vector<u8> t((const u8*)bytes, (const u8*)bytes + n);
    return add_program(b, lo, hi, addr, t, {});
  }
  // data written over a pre-decoded region: those harts must decode from RAM from now on
  for (u32 h = lo; h < hi; ++h) {
    u32 pg = b->hart_prog[h];
    if (pg != ~0u && addr + n > b->progs[pg].text_lo && addr < b->progs[pg].text_hi) {
      rc = set_u32_range(b, b->d.hot32 + (size_t)H32_FLAGS * b->d.S, h, h + 1, 0, HF_CODE_DIRTY, true); if (rc) return rc;
    }
  }
  return WSP_OK;
}

// ---- ELF64 little-endian reader (stands in for ELFIO use in Memory::loadElfFile, Memory.cpp:687)
struct Elf64_Ehdr_ { u8 ident[16]; u16 type, machine; u32 version; u64 entry, phoff, shoff; u32 flags; u16 ehsize, phentsize, phnum, shentsize, shnum, shstrndx; };
struct Elf64_Phdr_ { u32 type, flags; u64 offset, vaddr, paddr, filesz, memsz, align; };
struct Elf64_Shdr_ { u32 name, type; u64 flags, addr, offset, size; u32 link, info; u64 addralign, entsize; };
struct Elf64_Sym_ { u32 name; u8 info, other; u16 shndx; u64 value, size; };
// ELF32 (an RV32 program, README.md:167-185): the same records with 32-bit fields, widened on the way in
struct Elf32_Ehdr_ { u8 ident[16]; u16 type, machine; u32 version; u32 entry, phoff, shoff; u32 flags; u16 ehsize, phentsize, phnum, shentsize, shnum, shstrndx; };
struct Elf32_Phdr_ { u32 type, offset, vaddr, paddr, filesz, memsz, flags, align; };
struct Elf32_Shdr_ { u32 name, type, flags, addr, offset, size, link, info, addralign, entsize; };
struct Elf32_Sym_ { u32 name, value, size; u8 info, other; u16 shndx; };

// What one ELF file asks of a hart (Memory::loadElfFile, Memory.cpp:450-489, 687; System::loadElfFiles, System.cpp:288-354).
struct ElfSeg { u64 paddr, file_off, size; };
struct ParsedElf {
  std::vector<ElfSeg> loads;            // PT_LOAD segments, p_filesz bytes each, by physical address
  std::vector<u8> text; u64 text_lo = 0;   // the executable span [text_lo, text_lo + text.size()), gaps zero-filled
  std::map<std::string, u64> syms;
  u64 entry = 0, gp = 0, tdata = 0, tohost = ~0ULL;
};

static int parse_elf(const DevState& d, const void* elf, size_t n, ParsedElf& out)
{
  const u8* e = (const u8*)elf;
  if (!e || n < sizeof(Elf32_Ehdr_) || memcmp(e, "\177ELF", 4) != 0 || (e[4] == 2 && n < sizeof(Elf64_Ehdr_))) return fail(WSP_EFORMAT, "not an ELF file");
  const bool c32 = e[4] == 1;
  if (e[4] != 1 && e[4] != 2) return fail(WSP_EFORMAT, "ELF class is neither 32- nor 64-bit");
  if (c32 != ((d.isa & ISA_RV32) != 0)) return fail(WSP_EFORMAT, c32 ? "ELF32 file for an xlen=64 batch" : "ELF64 file for an xlen=32 batch");   // Session.cpp:1373-1424: the ELF class decides the width
  if (e[5] != 1) return fail(WSP_EFORMAT, "ELF is not little-endian");
  Elf64_Ehdr_ eh;
  if (c32) {
    Elf32_Ehdr_ h32; memcpy(&h32, e, sizeof h32);
    memcpy(eh.ident, h32.ident, 16); eh.type = h32.type; eh.machine = h32.machine; eh.version = h32.version; eh.entry = h32.entry; eh.phoff = h32.phoff; eh.shoff = h32.shoff;
    eh.flags = h32.flags; eh.ehsize = h32.ehsize; eh.phentsize = h32.phentsize; eh.phnum = h32.phnum; eh.shentsize = h32.shentsize; eh.shnum = h32.shnum; eh.shstrndx = h32.shstrndx;
  } else memcpy(&eh, e, sizeof eh);
  const size_t ph_sz = c32 ? sizeof(Elf32_Phdr_) : sizeof(Elf64_Phdr_), sh_sz = c32 ? sizeof(Elf32_Shdr_) : sizeof(Elf64_Shdr_), sym_sz = c32 ? sizeof(Elf32_Sym_) : sizeof(Elf64_Sym_);
  auto read_ph = [&](u32 i) { Elf64_Phdr_ ph; const u8* q = e + eh.phoff + (u64)i * eh.phentsize;
    if (c32) { Elf32_Phdr_ p; memcpy(&p, q, sizeof p); ph.type = p.type; ph.flags = p.flags; ph.offset = p.offset; ph.vaddr = p.vaddr; ph.paddr = p.paddr; ph.filesz = p.filesz; ph.memsz = p.memsz; ph.align = p.align; }
    else memcpy(&ph, q, sizeof ph);
    return ph; };
  auto read_sh = [&](u32 i) { Elf64_Shdr_ sh; const u8* q = e + eh.shoff + (u64)i * eh.shentsize;
    if (c32) { Elf32_Shdr_ p; memcpy(&p, q, sizeof p); sh.name = p.name; sh.type = p.type; sh.flags = p.flags; sh.addr = p.addr; sh.offset = p.offset; sh.size = p.size; sh.link = p.link; sh.info = p.info; sh.addralign = p.addralign; sh.entsize = p.entsize; }
    else memcpy(&sh, q, sizeof sh);
    return sh; };
  // every range check in the form "off > n || len > n - off": no sum can wrap
  auto in_file = [n](u64 off, u64 len) { return off <= n && len <= n - off; };
  if (eh.phnum && eh.phentsize != ph_sz) return fail(WSP_EFORMAT, "unexpected program header entry size");
  if (eh.shnum && eh.shentsize != sh_sz) return fail(WSP_EFORMAT, "unexpected section header entry size");
  if (!in_file(eh.phoff, (u64)eh.phnum * ph_sz)) return fail(WSP_EFORMAT, "program headers out of file");
  u64 xlo = ~0ULL, xhi = 0;
  std::vector<Elf64_Phdr_> xs;
  for (u32 i = 0; i < eh.phnum; ++i) {
    const Elf64_Phdr_ ph = read_ph(i);
    if (ph.type != 1 || ph.filesz == 0) continue;                            // PT_LOAD, p_filesz bytes only (Memory.cpp:450-489)
    if (!in_file(ph.offset, ph.filesz)) return fail(WSP_EFORMAT, "segment out of file");
    if (ph.paddr < d.mem_base || ph.paddr - d.mem_base > d.mem_size - ph.filesz || ph.filesz > d.mem_size)
      return fail(WSP_ERANGE, "ELF segment outside the hart arena");
    out.loads.push_back({ ph.paddr, ph.offset, ph.filesz });
    if (ph.flags & 1) { xlo = std::min(xlo, ph.paddr); xhi = std::max(xhi, ph.paddr + ph.filesz); xs.push_back(ph); }
  }
  // symbols + .tdata
  if (eh.shoff && in_file(eh.shoff, (u64)eh.shnum * sh_sz)) {
    std::vector<Elf64_Shdr_> sh(eh.shnum);
    for (u32 i = 0; i < eh.shnum; ++i) sh[i] = read_sh(i);
    // a name is taken only when its terminating NUL lies inside its string table (and the table inside the file)
    auto name_at = [&](const Elf64_Shdr_& tab, u64 off) -> const char* {
      if (!in_file(tab.offset, tab.size) || off >= tab.size) return nullptr;
      const char* p0 = (const char*)e + tab.offset + off;
      return memchr(p0, 0, tab.size - off) ? p0 : nullptr;
    };
    const Elf64_Shdr_* shstr = eh.shstrndx < eh.shnum ? &sh[eh.shstrndx] : nullptr;
    for (u32 i = 0; i < eh.shnum; ++i) {
      const char* sn = shstr ? name_at(*shstr, sh[i].name) : nullptr;
      if (sn && strcmp(sn, ".tdata") == 0) out.tdata = sh[i].addr;
      if (sh[i].type != 2 || sh[i].link >= eh.shnum || !in_file(sh[i].offset, sh[i].size)) continue;   // SHT_SYMTAB
      const Elf64_Shdr_& st = sh[sh[i].link];
      size_t cnt = sh[i].size / sym_sz;
      for (size_t k = 0; k < cnt; ++k) {
        Elf64_Sym_ sy;
        if (c32) { Elf32_Sym_ p; memcpy(&p, e + sh[i].offset + k * sym_sz, sizeof p); sy.name = p.name; sy.info = p.info; sy.other = p.other; sy.shndx = p.shndx; sy.value = p.value; sy.size = p.size; }
        else memcpy(&sy, e + sh[i].offset + k * sym_sz, sizeof sy);
        const char* nm = name_at(st, sy.name);
        if (nm && *nm) out.syms[nm] = sy.value;
      }
    }
  }
  if (xlo < xhi) {
    if (xhi - xlo > (64u << 20)) return fail(WSP_EUNSUPPORTED, "executable span larger than 64 MiB");
    out.text.assign(xhi - xlo, 0); out.text_lo = xlo;
    for (auto& ph : xs) memcpy(out.text.data() + (ph.paddr - xlo), e + ph.offset, ph.filesz);
  }
  out.entry = eh.entry;
  auto gp = out.syms.find("__global_pointer$"); if (gp != out.syms.end()) out.gp = gp->second;
  auto th = out.syms.find("tohost"); if (th != out.syms.end()) out.tohost = th->second;
  return WSP_OK;
}

// Fill x[reg] of harts [lo, hi) with v where the register is still zero (System.cpp:334-354: gp / tp are set only when unset).
__global__ void wsp_set_if_zero_kernel(DevState d, u32 lo, u32 hi, u32 reg, u64 v)
{
  const u32 h = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= hi) return;
  u64* p = d.x + (size_t)reg * d.S + slot_of(d, h);
  if (*p == 0) *p = v;
}

extern "C" int wsp_batch_load_elf_mem(wsp_batch* b, uint32_t lo, uint32_t hi, const void* elf, size_t n)
{
  int rc = check_range(b, lo, hi); if (rc) return rc;
  HIPOK(hipSetDevice(b->device));
  DevState& d = b->d;
  ParsedElf pe;
  rc = parse_elf(d, elf, n, pe); if (rc) return rc;
  for (auto& sg : pe.loads) { rc = write_arena(b, lo, hi, sg.paddr, (const u8*)elf + sg.file_off, sg.size); if (rc) return rc; }
  rc = add_program(b, lo, hi, pe.text.empty() ? 0 : pe.text_lo, pe.text, pe.syms); if (rc) return rc;
  // pc = entry, gp, tp (System.cpp:334-354); tohost (System.cpp:323)
  if (pe.entry) { rc = set_u64_slots(b, d.hot64 + (size_t)H64_PC * d.S, lo, hi, pe.entry & ~1ULL); if (rc) return rc; }
  const unsigned grid = (hi - lo + 255) / 256;
  if (pe.gp) hipLaunchKernelGGL(wsp_set_if_zero_kernel, dim3(grid), dim3(256), 0, b->stream, d, lo, hi, 3u, pe.gp);
  if (pe.tdata) hipLaunchKernelGGL(wsp_set_if_zero_kernel, dim3(grid), dim3(256), 0, b->stream, d, lo, hi, 4u, pe.tdata);
  if (pe.gp || pe.tdata) { HIPOK(hipGetLastError()); HIPOK(hipStreamSynchronize(b->stream)); }
  if (pe.tohost != ~0ULL) { rc = set_u64_slots(b, d.tohost, lo, hi, pe.tohost); if (rc) return rc; }
  return WSP_OK;
}

// ---- many ELF files in one call (BASELINE configs[4]: 125 000 distinct programs per GPU).  One staging upload, one scatter
// launch, one pre-decode launch, one upload each of the program table and the per-hart rows; nothing per program on the device.
struct ManySeg { u64 src_off, arena_off, n; u32 hart_lo, hart_hi; };
struct ManyProg { u64 img_off, n_bytes, rec_off; };

// block = one segment: its bytes go to every hart of [hart_lo, hart_hi) (consecutive threads -> consecutive harts)
__global__ void wsp_scatter_many_kernel(DevState d, const ManySeg* segs, const u8* src)
{
  const ManySeg sg = segs[blockIdx.x];
  const u64 w0 = sg.arena_off >> 3, nw = ((sg.arena_off + sg.n + 7) >> 3) - w0, nh = sg.hart_hi - sg.hart_lo;
  for (u64 i = threadIdx.x; i < nw * nh; i += blockDim.x) {
    const u32 hart = sg.hart_lo + (u32)(i % nh);
    const u64 bo = (w0 + i / nh) << 3;
    u8* dst = d.arena + arena_off(d, hart, bo);
    for (u32 k = 0; k < 8; ++k) { const u64 a = bo + k; if (a >= sg.arena_off && a < sg.arena_off + sg.n) dst[k] = src[sg.src_off + (a - sg.arena_off)]; }
  }
}

// block = one program: pre-decode its image exactly as wsp_predecode_kernel does
__global__ void wsp_predecode_many_kernel(const ManyProg* progs, const u8* imgs, u32 isa, Rec* recs)
{
  const ManyProg p = progs[blockIdx.x];
  const u8* img = imgs + p.img_off; Rec* out = recs + p.rec_off;
  const u64 nslots = p.n_bytes / 2;
  for (u64 i = threadIdx.x; i < nslots + 2; i += blockDim.x) {
    const u64 off = i * 2; Rec r;
    if (off + 2 > p.n_bytes) r = mk(OP_SLOWFETCH, 0, 0, 0, 0);
    else {
      const u32 lo = img[off] | ((u32)img[off + 1] << 8);
      if ((lo & 3) != 3) r = decode16(lo, isa);
      else if (off + 4 <= p.n_bytes) r = decode32(lo | ((u32)img[off + 2] << 16) | ((u32)img[off + 3] << 24), isa);
      else { r = mk(OP_SLOWFETCH, 0, 0, 0, 0); r.raw = lo; }
    }
    finish_rec(r);
    out[i] = r;
  }
}

extern "C" int wsp_batch_load_elf_many(wsp_batch* b, uint32_t hart_lo, uint32_t count, const void* const* elfs, const size_t* sizes, uint32_t harts_per_elf)
{
  if (!b || !elfs || !sizes) return fail(WSP_EINVAL, "null argument");
  if (count == 0 || harts_per_elf == 0) return fail(WSP_EINVAL, "count and harts_per_elf must be > 0");
  if ((u64)count * harts_per_elf > 0xffffffffULL - hart_lo) return fail(WSP_ERANGE, "hart range out of bounds");
  const u32 hart_hi = hart_lo + count * harts_per_elf;
  int rc = check_range(b, hart_lo, hart_hi); if (rc) return rc;
  HIPOK(hipSetDevice(b->device));
  DevState& d = b->d; const size_t S = d.S;
  std::vector<ParsedElf> pes(count);
  std::vector<ManySeg> segs; std::vector<ManyProg> mps(count);
  std::vector<u8> stage, imgs;
  u64 rec_total = 0;
  for (u32 i = 0; i < count; ++i) {
    rc = parse_elf(d, elfs[i], sizes[i], pes[i]);
    if (rc) { g_err = "ELF " + std::to_string(i) + ": " + g_err; return rc; }
    const u32 lo = hart_lo + i * harts_per_elf, hi = lo + harts_per_elf;
    // Segments are loaded in file order, so where two overlap the later one wins (Memory.cpp:450-489).  All pieces are scattered
    // by one launch here, in no particular order: keep of every segment only the parts no later segment covers.
    std::vector<std::pair<u64, u64>> covered;            // [lo, hi) byte ranges of the later segments
    for (size_t k = pes[i].loads.size(); k-- > 0;) {
      const ElfSeg& sg = pes[i].loads[k];
      std::vector<std::pair<u64, u64>> keep{ { sg.paddr, sg.paddr + sg.size } };
      for (auto& c : covered) {
        std::vector<std::pair<u64, u64>> nx;
        for (auto& r : keep) {
          if (c.second <= r.first || c.first >= r.second) { nx.push_back(r); continue; }
          if (c.first > r.first) nx.push_back({ r.first, c.first });
          if (c.second < r.second) nx.push_back({ c.second, r.second });
        }
        keep.swap(nx);
      }
      for (auto& r : keep) {
        segs.push_back({ (u64)stage.size(), r.first - d.mem_base, r.second - r.first, lo, hi });
        const u8* src = (const u8*)elfs[i] + sg.file_off + (r.first - sg.paddr);
        stage.insert(stage.end(), src, src + (r.second - r.first));
      }
      covered.push_back({ sg.paddr, sg.paddr + sg.size });
    }
    const u64 nb = pes[i].text.size() & ~(size_t)1;
    if (nb / 2 >= (1u << 29)) return fail(WSP_ERANGE, "code image of 1 GiB or more");
    mps[i] = { (u64)imgs.size(), nb, rec_total };
    imgs.insert(imgs.end(), pes[i].text.begin(), pes[i].text.begin() + nb);
    rec_total += nb ? nb / 2 + 2 : 0;
  }
  DevBuf dstage, dsegs, dimgs, dmps, drecs;
  if (!segs.empty()) {
    HIPOK(dstage.alloc(stage.size())); HIPOK(hipMemcpy(dstage.p, stage.data(), stage.size(), hipMemcpyHostToDevice));
    HIPOK(dsegs.alloc(segs.size() * sizeof(ManySeg))); HIPOK(hipMemcpy(dsegs.p, segs.data(), segs.size() * sizeof(ManySeg), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(wsp_scatter_many_kernel, dim3((unsigned)segs.size()), dim3(256), 0, b->stream, d, dsegs.as<ManySeg>(), dstage.as<u8>());
    HIPOK(hipGetLastError());
  }
  Rec* recs_base = nullptr;
  if (rec_total) {
    HIPOK(dimgs.alloc(imgs.size())); HIPOK(hipMemcpy(dimgs.p, imgs.data(), imgs.size(), hipMemcpyHostToDevice));
    HIPOK(dmps.alloc(count * sizeof(ManyProg))); HIPOK(hipMemcpy(dmps.p, mps.data(), count * sizeof(ManyProg), hipMemcpyHostToDevice));
    HIPOK(drecs.alloc(rec_total * sizeof(Rec)));
    hipLaunchKernelGGL(wsp_predecode_many_kernel, dim3(count), dim3(128), 0, b->stream, dmps.as<ManyProg>(), dimgs.as<u8>(), d.isa, drecs.as<Rec>());
    HIPOK(hipGetLastError());
    recs_base = drecs.as<Rec>();
  }
  HIPOK(hipStreamSynchronize(b->stream));
  if (recs_base) { b->rec_pools.push_back(recs_base); drecs.p = nullptr; }     // one allocation for all images, owned by the batch
  // program table + per-hart rows, each uploaded once
  const u32 id0 = (u32)b->progs.size();
  for (u32 i = 0; i < count; ++i) {
    Program p; p.symbols = std::move(pes[i].syms);
    if (mps[i].n_bytes) { p.text_lo = pes[i].text_lo; p.text_hi = p.text_lo + mps[i].n_bytes; p.d_recs = recs_base + mps[i].rec_off; p.pooled = true; }
    if (harts_per_elf > 1) { rc = build_hrecs(b, p); if (rc) return rc; }
    b->progs.push_back(std::move(p));
  }
  rc = upload_progs(b); if (rc) return rc;
  std::vector<u32> prow(S); std::vector<u64> pcrow(S), throw_(S), gprow(S), tprow(S);
  HIPOK(hipMemcpy(prow.data(), d.prog, S * 4, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(pcrow.data(), d.hot64 + (size_t)H64_PC * S, S * 8, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(throw_.data(), d.tohost, S * 8, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(gprow.data(), d.x + (size_t)3 * S, S * 8, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(tprow.data(), d.x + (size_t)4 * S, S * 8, hipMemcpyDeviceToHost));
  for (u32 i = 0; i < count; ++i) {
    const ParsedElf& pe = pes[i];
    for (u32 h = hart_lo + i * harts_per_elf; h < hart_lo + (i + 1) * harts_per_elf; ++h) {
      const u32 s = slot_of(d, h);
      prow[s] = id0 + i; b->hart_prog[h] = id0 + i;
      if (pe.entry) pcrow[s] = pe.entry & ~1ULL;
      if (pe.tohost != ~0ULL) throw_[s] = pe.tohost;
      if (pe.gp && !gprow[s]) gprow[s] = pe.gp;
      if (pe.tdata && !tprow[s]) tprow[s] = pe.tdata;
      if (b->hart_stop_pc[h] != ~0ULL) { rc = patch_stop(b, id0 + i, b->hart_stop_pc[h]); if (rc) return rc; }
    }
  }
  HIPOK(hipMemcpy(d.prog, prow.data(), S * 4, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(d.hot64 + (size_t)H64_PC * S, pcrow.data(), S * 8, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(d.tohost, throw_.data(), S * 8, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(d.x + (size_t)3 * S, gprow.data(), S * 8, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(d.x + (size_t)4 * S, tprow.data(), S * 8, hipMemcpyHostToDevice));
  return WSP_OK;
}

extern "C" int wsp_batch_load_elf(wsp_batch* b, uint32_t lo, uint32_t hi, const char* path)
{
  if (!path) return fail(WSP_EINVAL, "null path");
  FILE* fp = fopen(path, "rb");
  if (!fp) return fail(WSP_EIO, std::string("cannot open ") + path);
  std::vector<u8> buf; u8 tmp[65536]; size_t k;
  while ((k = fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + k);
  fclose(fp);
  return wsp_batch_load_elf_mem(b, lo, hi, buf.data(), buf.size());
}

extern "C" int wsp_batch_symbol(wsp_batch* b, uint32_t hart, const char* name, uint64_t* addr)
{
  int rc = check_range(b, hart, hart + 1); if (rc) return rc;
  u32 pg = b->hart_prog[hart];
  if (pg == ~0u) return fail(WSP_EINVAL, "no ELF loaded for this hart");
  auto it = b->progs[pg].symbols.find(name ? name : "");
  if (it == b->progs[pg].symbols.end()) return fail(WSP_EINVAL, std::string("symbol not found: ") + (name ? name : ""));
  *addr = it->second;
  return WSP_OK;
}

// Harts of [lo, hi) that ended on an instruction / retired count limit or on their stop address become runnable again.
__global__ void wsp_resume_kernel(DevState d, u32 lo, u32 hi)
{
  const u32 h = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= hi) return;
  u32* st = d.hot32 + (size_t)H32_STATUS * d.S + slot_of(d, h);
  if (*st == ST_STOP_LIMIT || *st == ST_STOP_ADDR) *st = ST_RUNNING;
}

extern "C" int wsp_batch_resume(wsp_batch* b, uint32_t lo, uint32_t hi)
{
  int rc = check_range(b, lo, hi); if (rc) return rc;
  HIPOK(hipSetDevice(b->device));
  hipLaunchKernelGGL(wsp_resume_kernel, dim3((hi - lo + 255) / 256), dim3(256), 0, b->stream, b->d, lo, hi);
  HIPOK(hipGetLastError()); HIPOK(hipStreamSynchronize(b->stream));
  return WSP_OK;
}

extern "C" int wsp_batch_set_stop(wsp_batch* b, uint32_t lo, uint32_t hi, uint64_t tohost, uint64_t max_inst, uint64_t max_ret, uint64_t stop_pc)
{
  int rc = check_range(b, lo, hi); if (rc) return rc;
  HIPOK(hipSetDevice(b->device));
  if (stop_pc != ~0ULL && (stop_pc & 1)) return fail(WSP_EINVAL, "stop_pc must be even (a pc never has bit 0 set)");
  if (tohost != ~0ULL) { rc = set_u64_slots(b, b->d.tohost, lo, hi, tohost); if (rc) return rc; }
  rc = set_u64_slots(b, b->d.max_exec, lo, hi, max_inst); if (rc) return rc;
  rc = set_u64_slots(b, b->d.max_ret, lo, hi, max_ret); if (rc) return rc;
  rc = set_u64_slots(b, b->d.stop_pc, lo, hi, stop_pc); if (rc) return rc;
  if (stop_pc != ~0ULL) {
    // The inline loops never compare pcs: the pre-decoded record at the stop address is turned into OP_STOPCHK, which is outside
    // their subset, so a hart arriving there takes the general path, which makes the comparison (Lane::cold_step).  The image may
    // be shared: harts with another stop address decode Rec.raw again and go on.
    u32 last = ~0u;
    for (u32 h = lo; h < hi; ++h) {
      const u32 pg = b->hart_prog[h];
      if (pg == last || pg == ~0u) continue;
      last = pg;
      rc = patch_stop(b, pg, stop_pc); if (rc) return rc;
    }
  }
  for (u32 h = lo; h < hi; ++h) b->hart_stop_pc[h] = stop_pc;
  return WSP_OK;
}

// location of a modelled CSR in device state: returns 0 hot64, 1 hot32, 2 cold, -1 none
int csr_loc(u32 num, u32* ix)
{
  switch (num) {
    case C_FCSR: *ix = H32_FCSR; return 1;
    case C_MSTATUS: *ix = H64_MSTATUS; return 0;
    case C_MCYCLE: *ix = H64_CYCLE; return 0; case C_MINSTRET: *ix = H64_INSTRET; return 0;
    case C_MEDELEG: *ix = CI_MEDELEG; return 2; case C_MIDELEG: *ix = CI_MIDELEG; return 2; case C_MIE: *ix = CI_MIE; return 2;
    case C_MIP: *ix = CI_MIP; return 2; case C_MTVEC: *ix = CI_MTVEC; return 2; case C_MCOUNTEREN: *ix = CI_MCOUNTEREN; return 2;
    case C_MENVCFG: *ix = CI_MENVCFG; return 2; case C_MCOUNTINHIBIT: *ix = CI_MCOUNTINHIBIT; return 2; case C_MSCRATCH: *ix = CI_MSCRATCH; return 2;
    case C_MEPC: *ix = CI_MEPC; return 2; case C_MCAUSE: *ix = CI_MCAUSE; return 2; case C_MTVAL: *ix = CI_MTVAL; return 2;
    case C_STVEC: *ix = CI_STVEC; return 2; case C_SCOUNTEREN: *ix = CI_SCOUNTEREN; return 2; case C_SENVCFG: *ix = CI_SENVCFG; return 2;
    case C_SSCRATCH: *ix = CI_SSCRATCH; return 2; case C_SEPC: *ix = CI_SEPC; return 2; case C_SCAUSE: *ix = CI_SCAUSE; return 2;
    case C_STVAL: *ix = CI_STVAL; return 2; case C_SATP: *ix = CI_SATP; return 2;
    case C_VSTART: *ix = CI_VSTART; return 2; case C_VCSR: *ix = CI_VCSR; return 2; case C_VL: *ix = CI_VL; return 2; case C_VTYPE: *ix = CI_VTYPE; return 2;
    default:
      if (num >= 0x3A0 && num <= 0x3AF && !(num & 1)) { *ix = CI_PMPCFG0 + ((num - 0x3A0) >> 1); return 2; }
      if (num >= 0x3B0 && num <= 0x3EF) { *ix = CI_PMPADDR0 + (num - 0x3B0); return 2; }
      return -1;
  }
}

extern "C" int wsp_batch_peek(wsp_batch* b, uint32_t hart, uint32_t res, uint64_t idx, void* val, size_t size)
{
  int rc = check_range(b, hart, hart + 1); if (rc) return rc;
  if (!val) return fail(WSP_EINVAL, "null value");
  HIPOK(hipSetDevice(b->device));
  DevState& d = b->d; u32 s = slot_of(d, hart); size_t S = d.S;
  u64 v = 0;
  switch (res) {
    case WSP_RES_PC: {   // idx 0: pc; 1: executed-instruction count (the log's tag, snapshot "po"); 2: retired-instruction count ("pr"); 3: privilege mode ("pm"); 4: stop state (WSP_ST_*, peek only)
      if (idx > 4) return fail(WSP_ERANGE, "pc resource index");
      if (idx >= 3) { u32 pm; HIPOK(hipMemcpy(&pm, d.hot32 + (idx == 3 ? H32_PRIV : H32_STATUS) * S + s, 4, hipMemcpyDeviceToHost)); v = pm; memcpy(val, &v, std::min<size_t>(size, 8)); return WSP_OK; }
      const u32 row = idx == 0 ? H64_PC : idx == 1 ? H64_EXEC : H64_RETIRED;
      HIPOK(hipMemcpy(&v, d.hot64 + row * S + s, 8, hipMemcpyDeviceToHost));
      if (idx) { memcpy(val, &v, std::min<size_t>(size, 8)); return WSP_OK; }
      break; }
    case WSP_RES_X: if (idx > 31) return fail(WSP_ERANGE, "x register"); HIPOK(hipMemcpy(&v, d.x + idx * S + s, 8, hipMemcpyDeviceToHost)); break;
    case WSP_RES_F: if (idx > 31) return fail(WSP_ERANGE, "f register"); HIPOK(hipMemcpy(&v, d.f + idx * S + s, 8, hipMemcpyDeviceToHost)); break;
    case WSP_RES_CSR: {
      u32 ix; u32 n = (u32)idx;
      if (n == C_FFLAGS || n == C_FRM) { u32 f; HIPOK(hipMemcpy(&f, d.hot32 + H32_FCSR * S + s, 4, hipMemcpyDeviceToHost)); v = n == C_FFLAGS ? (f & 0x1f) : ((f >> 5) & 7); break; }
      if (n == C_MISA) { v = MISA_RESET; break; }
      if (n == C_MHARTID) { v = d.hart_id_base + hart; break; }
      if (n == C_SSTATUS) { HIPOK(hipMemcpy(&v, d.hot64 + H64_MSTATUS * S + s, 8, hipMemcpyDeviceToHost)); v &= SSTATUS_RMASK; break; }
      int loc = csr_loc(n, &ix);
      if (loc == 0) HIPOK(hipMemcpy(&v, d.hot64 + ix * S + s, 8, hipMemcpyDeviceToHost));
      else if (loc == 1) { u32 f; HIPOK(hipMemcpy(&f, d.hot32 + ix * S + s, 4, hipMemcpyDeviceToHost)); v = f; }
      else if (loc == 2) HIPOK(hipMemcpy(&v, d.cold + ix * S + s, 8, hipMemcpyDeviceToHost));
      else return fail(WSP_EUNSUPPORTED, "CSR not modelled");
      break; }
    case WSP_RES_V:                      // Hart::peekVecReg (Hart.hpp:219-325): the whole register, element 0 first
      if (!d.vlenb || idx > 31 || size < d.vlenb) return fail(WSP_ERANGE, "vector register");
      HIPOK(hipMemcpy(val, d.vreg + ((size_t)hart * 32 + idx) * d.vlenb, d.vlenb, hipMemcpyDeviceToHost));
      return WSP_OK;
    case WSP_RES_MEM: {
      if (idx < d.mem_base || idx - d.mem_base > d.mem_size - size || size > d.mem_size) return fail(WSP_ERANGE, "memory peek outside arena");
      DevBuf tmp; HIPOK(tmp.alloc(size));
      hipLaunchKernelGGL(wsp_gather_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, b->stream, d, hart, idx - d.mem_base, tmp.as<u8>(), (u64)size);
      HIPOK(hipGetLastError()); HIPOK(hipStreamSynchronize(b->stream));
      HIPOK(hipMemcpy(val, tmp.p, size, hipMemcpyDeviceToHost));
      return WSP_OK; }
    default: return fail(WSP_EINVAL, "unknown resource");
  }
  if (d.isa & ISA_RV32) {   // the 32-bit view of a value kept in its RV64 shape (x registers sign-extended, CSR file as CsRegs<uint32_t> shows it)
    if (res == WSP_RES_CSR && (idx == C_MSTATUS || idx == C_SSTATUS || idx == C_MCAUSE || idx == C_SCAUSE)) v = (v & 0x7fffffffULL) | ((v >> 63) << 31);
    else if (res == WSP_RES_CSR && idx == C_MISA) v = 0x4034112dULL;
    else if (res != WSP_RES_F) v &= 0xffffffffULL;
  }
  memcpy(val, &v, std::min<size_t>(size, 8));
  return WSP_OK;
}

extern "C" int wsp_batch_poke(wsp_batch* b, uint32_t hart, uint32_t res, uint64_t idx, const void* val, size_t size)
{
  int rc = check_range(b, hart, hart + 1); if (rc) return rc;
  if (!val) return fail(WSP_EINVAL, "null value");
  HIPOK(hipSetDevice(b->device));
  DevState& d = b->d; u32 s = slot_of(d, hart); size_t S = d.S;
  u64 v = 0; memcpy(&v, val, std::min<size_t>(size, 8));
  const bool rv32 = (d.isa & ISA_RV32) != 0;
  if (rv32 && res != WSP_RES_F && res != WSP_RES_MEM && res != WSP_RES_V) {
    v &= 0xffffffffULL;
    if (res == WSP_RES_X) v = (u64)(i64)(i32)v;                                                        // registers are held sign-extended
    if (res == WSP_RES_CSR && (idx == C_MCAUSE || idx == C_SCAUSE)) v = (v & 0x7fffffffULL) | ((v >> 31) << 63);
  }
  switch (res) {
    case WSP_RES_PC: {   // idx as in wsp_batch_peek (the counters are restored by snapshot import)
      if (idx > 3) return fail(WSP_ERANGE, "pc resource index");
      if (idx == 0) v &= ~1ULL; else memcpy(&v, val, std::min<size_t>(size, 8));
      if (idx == 3) {
        if (v != PRV_M && !(v == PRV_U && (d.isa & ISA_U)) && !(v == PRV_S && (d.isa & ISA_S))) return fail(WSP_EINVAL, "privilege mode not in the ISA");
        const u32 pm = (u32)v; HIPOK(hipMemcpy(d.hot32 + H32_PRIV * S + s, &pm, 4, hipMemcpyHostToDevice)); break;
      }
      const u32 row = idx == 0 ? H64_PC : idx == 1 ? H64_EXEC : H64_RETIRED;
      HIPOK(hipMemcpy(d.hot64 + row * S + s, &v, 8, hipMemcpyHostToDevice)); break; }
    case WSP_RES_X: if (idx > 31) return fail(WSP_ERANGE, "x register"); if (idx) HIPOK(hipMemcpy(d.x + idx * S + s, &v, 8, hipMemcpyHostToDevice)); break;
    case WSP_RES_F: if (idx > 31) return fail(WSP_ERANGE, "f register"); HIPOK(hipMemcpy(d.f + idx * S + s, &v, 8, hipMemcpyHostToDevice)); break;
    case WSP_RES_CSR: {
      u32 ix; u32 n = (u32)idx; int loc = csr_loc(n, &ix);
      if (n == C_MSTATUS) v = ((rv32 ? (3ULL << MST_MPP_SH) : MSTATUS_RESET) & ~MSTATUS_WMASK) | (v & MSTATUS_WMASK) | ((v & (rv32 ? (1ULL << 31) : MST_SD)) ? MST_SD : 0);
      if (loc == 0) HIPOK(hipMemcpy(d.hot64 + ix * S + s, &v, 8, hipMemcpyHostToDevice));
      else if (loc == 1) { u32 f = (u32)v & 0xff; HIPOK(hipMemcpy(d.hot32 + ix * S + s, &f, 4, hipMemcpyHostToDevice)); }
      else if (loc == 2) {
        if (ix >= CI_PMPCFG0 && ix < CI_PMPADDR0 + 64) {   // pmpcfg / pmpaddr pokes are legalised and honour locks like CSR writes (PmpManager.hpp:480-551, CsRegs.cpp:6773)
          u64 cfg[8]; for (u32 k = 0; k < 8; ++k) HIPOK(hipMemcpy(&cfg[k], d.cold + (size_t)(CI_PMPCFG0 + k) * S + s, 8, hipMemcpyDeviceToHost));
          auto cb = [&](u32 e) { return (u32)(cfg[e >> 3] >> (8 * (e & 7))) & 0xff; };
          if (ix < CI_PMPADDR0) v = Lane<DevMem>::legalize_pmpcfg(cfg[ix - CI_PMPCFG0], v) & 0x9f9f9f9f9f9f9f9fULL;
          else {
            const u32 e = ix - CI_PMPADDR0;
            if ((cb(e) & 0x80) || (e < 63 && (cb(e + 1) & 0x80) && ((cb(e + 1) >> 3) & 3) == 1)) break;
            v &= 0x003fffffffffffffULL;
          }
        }
        if (n == C_SATP && d.tlb_n) {   // a MODE change empties the TLB (Tlb::setMode, Tlb.hpp:378-383): valid and counter cleared, the rest kept
          u64 old = 0; HIPOK(hipMemcpy(&old, d.cold + ix * S + s, 8, hipMemcpyDeviceToHost));
          if ((old >> 60) != (v >> 60)) {
            std::vector<u64> t(2 * (size_t)d.tlb_n); u64* dt = d.tlb + (size_t)hart * 2 * d.tlb_n;
            HIPOK(hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < t.size(); i += 2) t[i] &= ~((3ULL << 54) | ((u64)TE_VALID << 56));
            HIPOK(hipMemcpy(dt, t.data(), t.size() * 8, hipMemcpyHostToDevice));
          }
        }
        HIPOK(hipMemcpy(d.cold + ix * S + s, &v, 8, hipMemcpyHostToDevice));
        if (n == C_MCOUNTINHIBIT) { u32 f; HIPOK(hipMemcpy(&f, d.hot32 + H32_FLAGS * S + s, 4, hipMemcpyDeviceToHost)); f = (f & ~(HF_INH_CY | HF_INH_IR)) | ((v & 1) ? (u32)HF_INH_CY : 0) | ((v & 4) ? (u32)HF_INH_IR : 0); HIPOK(hipMemcpy(d.hot32 + H32_FLAGS * S + s, &f, 4, hipMemcpyHostToDevice)); }
      } else return fail(WSP_EUNSUPPORTED, "CSR not modelled");
      break; }
    case WSP_RES_V:
      if (!d.vlenb || idx > 31 || size < d.vlenb) return fail(WSP_ERANGE, "vector register");
      HIPOK(hipMemcpy(d.vreg + ((size_t)hart * 32 + idx) * d.vlenb, val, d.vlenb, hipMemcpyHostToDevice));
      break;
    case WSP_RES_MEM: return wsp_batch_load_image(b, hart, hart + 1, idx, val, size, 0);
    default: return fail(WSP_EINVAL, "unknown resource");
  }
  return WSP_OK;
}

extern "C" int wsp_batch_poke_each(wsp_batch* b, uint32_t lo, uint32_t hi, uint32_t res, uint64_t idx, const uint64_t* vals)
{
  int rc = check_range(b, lo, hi); if (rc) return rc;
  if (!vals) return fail(WSP_EINVAL, "null values");
  HIPOK(hipSetDevice(b->device));
  DevState& d = b->d;
  u64 key = idx;
  if (res == WSP_RES_X) { if (idx == 0 || idx > 31) return fail(WSP_ERANGE, "x register"); }
  else if (res == WSP_RES_MEM) {
    if ((idx & 7) || idx < d.mem_base || idx - d.mem_base > d.mem_size - 8) return fail(WSP_ERANGE, "memory dword outside arena or unaligned");
    key = idx - d.mem_base;
  } else return fail(WSP_EINVAL, "poke_each supports X and MEM");
  DevBuf dv; size_t n = hi - lo;
  HIPOK(dv.alloc(n * 8)); HIPOK(hipMemcpy(dv.p, vals, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(wsp_poke_each_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->stream, d, lo, hi, res, key, dv.as<u64>());
  HIPOK(hipGetLastError()); HIPOK(hipStreamSynchronize(b->stream));
  return WSP_OK;
}

template <bool TRACE, bool FP> static hipError_t launch_step(wsp_batch* b, u64 steps, u32 notrace)
{
  DevState& d = b->d;
  size_t lds = (size_t)d.lpw * WSP_XSLOTS * 8 * (FP ? 2 : 1);
  if (d.vreg) lds = (size_t)d.lpw * (2 * WSP_XSLOTS * 8 + 32 * d.vlenb + 32);     // x block, f block, vector register files, vector CSRs
  d.mtab_off = (u32)lds; lds += 64 * 4;                                          // the lane table (ln_of)
  hipError_t e = hipMemcpyAsync(b->d_state, &d, sizeof(DevState), hipMemcpyHostToDevice, b->stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((wsp_step_kernel<TRACE, FP>), dim3(d.n_waves), dim3(64), lds, b->stream, d, (const DevState*)b->d_state, steps, notrace);
  return hipGetLastError();
}

extern "C" int64_t wsp_batch_run(wsp_batch* b, uint64_t max_steps, uint32_t flags)
{
  if (!b) return fail(WSP_EINVAL, "null batch");
  HIPOK(hipSetDevice(b->device));
  u64 remaining = max_steps; u32 running = 0;
  b->last_kernel_s = 0; b->last_launches = 0;
  bool trace = (flags & WSP_RUN_TRACE_HASH) != 0;
  // The lean TRACE = false kernels only for untraced batches whose every wave runs a single program image (their converged
  // loop); any other untraced batch runs the TRACE = true kernel with `notrace` set.
  bool uniform = true;
  {
    const DevState& d = b->d;
    for (u32 w = 0; w < d.n_waves && uniform; ++w) {
      const u32 h0 = w * d.lpw, h1 = std::min(d.n_harts, h0 + d.lpw), pg = b->hart_prog[h0];
      if (pg == ~0u) uniform = false;
      for (u32 h = h0 + 1; h < h1 && uniform; ++h) uniform = b->hart_prog[h] == pg;
    }
  }
  const bool lean = !trace && uniform;
  // Steps per launch.  The lanes without a running hart are dealt out as copies of the running ones at the start of a launch
  // (step kernel, "Mirror lanes").  Cutting the run of a batch of different programs into launches of 4096 steps, so that the
  // lanes of finished harts are dealt out again, measured no gain (1600 different Sv39 fuzz programs, 4 per wave: 351 ms sliced,
  // 357 ms in one launch, 360 ms with the lanes dealt out again inside the launch; 144 ms one hart per wave - profiles/r03_exp_*),
  // so a launch runs 2^24 steps; WSP_SLICE_STEPS overrides it (tuning aid).
  u64 SLICE = 1ULL << 24;
  if (const char* e = getenv("WSP_SLICE_STEPS")) { const unsigned long long v = strtoull(e, nullptr, 0); if (v) SLICE = v; }
  (void)uniform;
  const u32 nt = trace ? 0u : 1u;
  do {
    u64 n = std::min(remaining, SLICE);
    HIPOK(hipMemsetAsync(b->d.running, 0, 4, b->stream));
    HIPOK(hipEventRecord(b->ev0, b->stream));
    hipError_t e = !lean ? (b->has_fp ? launch_step<true, true>(b, n, nt) : launch_step<true, false>(b, n, nt))
                         : (b->has_fp ? launch_step<false, true>(b, n, nt) : launch_step<false, false>(b, n, nt));
    if (e != hipSuccess) return fail(WSP_EHIP, std::string("step kernel launch: ") + hipGetErrorString(e));
    HIPOK(hipEventRecord(b->ev1, b->stream));
    HIPOK(hipMemcpyAsync(&running, b->d.running, 4, hipMemcpyDeviceToHost, b->stream));
    HIPOK(hipStreamSynchronize(b->stream));
    float ms = 0; HIPOK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->last_kernel_s += ms * 1e-3; b->last_launches++;
    remaining -= n;
  } while (running && remaining);
  return (int64_t)running;
}

#ifdef WSP_PROF
extern "C" int wsp_debug_prof(uint64_t* out, int reset)
{
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(wsp_prof), sizeof h) != hipSuccess) return -1;
  for (int i = 0; i < 16; ++i) out[i] = h[i];
  if (reset) { memset(h, 0, sizeof h); hipMemcpyToSymbol(HIP_SYMBOL(wsp_prof), h, sizeof h); }
  return 0;
}
#endif

extern "C" int wsp_batch_last_run_stats(wsp_batch* b, double* s, uint32_t* launches)
{ if (!b) return fail(WSP_EINVAL, "null batch"); if (s) *s = b->last_kernel_s; if (launches) *launches = b->last_launches; return WSP_OK; }

// memory digest + result records: partial sums in `memhash` (S words, zeroed here), records into `d_out` (device memory)
static int results_on_device(wsp_batch* b, u64* memhash, wsp_result* d_out)
{
  DevState& d = b->d;
  HIPOK(hipMemsetAsync(memhash, 0, (size_t)d.S * 8, b->stream));
  const u64 total = (d.mem_size >> 3) * d.lpw;          // words per wave block; mem_size is a multiple of 4096
  // enough chunks that the grid has >= ~8 workgroups per CU even for small batches; chunk = multiple of 256 words
  u64 chunks = 1;
  while (chunks * d.n_waves < 2048 && total / (chunks * 2) >= 4 * WSP_MH_THREADS) chunks *= 2;
  u64 wpc = (total / chunks + WSP_MH_THREADS - 1) / WSP_MH_THREADS * WSP_MH_THREADS;
  chunks = (total + wpc - 1) / wpc;
  HIPOK(hipEventRecord(b->ev0, b->stream));
  hipLaunchKernelGGL(wsp_memhash_kernel, dim3(d.n_waves, (unsigned)chunks), dim3(WSP_MH_THREADS), 0, b->stream, d, wpc, memhash);
  HIPOK(hipGetLastError());
  HIPOK(hipEventRecord(b->ev1, b->stream));
  hipLaunchKernelGGL(wsp_results_kernel, dim3((d.n_harts + 127) / 128), dim3(128), 0, b->stream, d, (const u64*)memhash, d_out);
  HIPOK(hipGetLastError());
  return WSP_OK;
}

extern "C" int wsp_batch_results(wsp_batch* b, wsp_result* out)
{
  if (!b || !out) return fail(WSP_EINVAL, "null argument");
  HIPOK(hipSetDevice(b->device));
  DevState& d = b->d;
  int rc = results_on_device(b, b->d_memhash, b->d_results); if (rc) return rc;
  HIPOK(hipMemcpyAsync(out, b->d_results, (size_t)d.n_harts * sizeof(wsp_result), hipMemcpyDeviceToHost, b->stream));
  HIPOK(hipStreamSynchronize(b->stream));
  float ms = 0; HIPOK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
  b->last_hash_s = ms * 1e-3; b->last_hash_bytes = (u64)d.n_harts * d.mem_size;
  return WSP_OK;
}

// The variant of wsp_batch_results that allocates nothing (for callers that cannot: tight serving loops, harnesses that forbid
// hipMalloc, code around a graph capture - the call itself waits for the batch's stream, so it cannot sit INSIDE a capture).
// Same two kernels; the digest's partial sums live in the static device array wsp_scratch instead of a buffer of the batch,
// and the records are written straight into the caller's device buffer.  Every call on a GPU shares that GPU's array: the
// part used is cleared at the start of each call, and calls that overlap on one GPU race.
extern "C" int wsp_batch_results_noalloc(wsp_batch* b, wsp_result* d_out)
{
  if (!b || !d_out) return fail(WSP_EINVAL, "null argument");
  if (b->d.S > WSP_SCRATCH_SLOTS) return fail(WSP_ERANGE, "wsp_batch_results_noalloc: batch has more than 1048576 hart slots (n_waves * 64): use wsp_batch_results");
  u64* sp = b->device >= 0 && b->device < 64 ? g_scratch_ptr[b->device] : nullptr;   // looked up when the batch was created
  if (!sp) return fail(WSP_EHIP, "wsp_batch_results_noalloc: no scratch address for this device");
  HIPOK(hipSetDevice(b->device));
  int rc = results_on_device(b, sp, d_out); if (rc) return rc;
  HIPOK(hipStreamSynchronize(b->stream));
  return WSP_OK;
}

extern "C" int wsp_batch_last_hash_stats(wsp_batch* b, double* s, uint64_t* bytes)
{ if (!b) return fail(WSP_EINVAL, "null batch"); if (s) *s = b->last_hash_s; if (bytes) *bytes = b->last_hash_bytes; return WSP_OK; }

extern "C" int wsp_batch_trace_ring(wsp_batch* b, uint32_t words_per_hart)
{
  if (!b) return fail(WSP_EINVAL, "null batch");
  HIPOK(hipSetDevice(b->device));
  HIPOK(hipStreamSynchronize(b->stream));
  DevState& d = b->d;
  if (d.ring) { HIPOK(hipFree(d.ring)); d.ring = nullptr; }
  if (d.ring_pos) { HIPOK(hipFree(d.ring_pos)); d.ring_pos = nullptr; }
  d.ring_cap = 0;
  if (words_per_hart == 0) return WSP_OK;
  if (words_per_hart < 5) return fail(WSP_EINVAL, "trace ring: at least 5 words per hart (one record is pc, meta, kinds and one pair)");
  const size_t bytes = (size_t)d.n_harts * words_per_hart * 8;
  if (hipMalloc((void**)&d.ring, bytes) != hipSuccess) { d.ring = nullptr; return fail(WSP_ENOMEM, "trace ring: hipMalloc failed"); }
  if (hipMalloc((void**)&d.ring_pos, (size_t)d.n_harts * 4) != hipSuccess) { hipFree(d.ring); d.ring = nullptr; d.ring_pos = nullptr; return fail(WSP_ENOMEM, "trace ring: hipMalloc failed"); }
  HIPOK(hipMemset(d.ring_pos, 0, (size_t)d.n_harts * 4));
  d.ring_cap = words_per_hart;
  return WSP_OK;
}

extern "C" int64_t wsp_batch_trace_read(wsp_batch* b, uint32_t hart, uint64_t* words, size_t cap)
{
  if (!b) return fail(WSP_EINVAL, "null batch");
  const DevState& d = b->d;
  if (!d.ring) return fail(WSP_EINVAL, "trace ring is off");
  if (hart >= d.n_harts) return fail(WSP_ERANGE, "hart out of range");
  HIPOK(hipSetDevice(b->device));
  HIPOK(hipStreamSynchronize(b->stream));
  u32 n = 0;
  HIPOK(hipMemcpy(&n, d.ring_pos + hart, 4, hipMemcpyDeviceToHost));
  if (n > cap) return fail(WSP_ERANGE, "trace buffer too small");
  if (n) HIPOK(hipMemcpy(words, d.ring + (size_t)hart * d.ring_cap, (size_t)n * 8, hipMemcpyDeviceToHost));
  return (int64_t)n;
}

// ---- the commit log as text (printTrace.cpp:302-517).  One record of the ring: pc, meta (instruction | priv << 32 | trap << 40 |
// size << 48 | has-ld/st-address << 55 | pairs << 56), kinds (4 bits per pair), the pairs, the ld/st address pair if flagged, then
// vlenb / 8 words for every vector register pair.  Lines in the reference's layouts for the two widths (printTrace.cpp:56-209).
static void format_record(std::string& out, const u64* w, u64 tag, u32 vlenb, bool rv32, bool has_d, u32 np, bool ldst)
{
  const u64 pc = w[0], meta = w[1], kinds = w[2];
  const u32 inst = (u32)meta, priv = (u32)(meta >> 32) & 0xff, trap = (u32)(meta >> 40) & 0xff, size = (u32)(meta >> 48) & 0x7f;
  const u64* pairs = w + 3;
  const u64* extra = pairs + 2 * np;
  std::string text = wspdis::disassemble(size == 4 ? inst : (inst & 0xffff), !rv32);
  char buf[256];
  if (ldst) {
    if (extra[0] == extra[1]) snprintf(buf, sizeof buf, " [0x%llx]", (unsigned long long)extra[0]);
    else snprintf(buf, sizeof buf, " [0x%llx:0x%llx]", (unsigned long long)extra[0], (unsigned long long)extra[1]);
    text += buf; extra += 2;
  }
  if (trap == 2) text += " (interrupted)"; else if (trap == 1) text += " (exception)";
  char op[16];
  if (size == 4) snprintf(op, sizeof op, "%08x", inst); else snprintf(op, sizeof op, "%04x", inst & 0xffff);
  const char* pm = priv == 3 ? "M" : priv == 1 ? "S" : "U";
  char head[96];
  if (rv32) snprintf(head, sizeof head, "#%llu %d %2s %08x %8s ", (unsigned long long)tag, 0, pm, (u32)pc, op);
  else snprintf(head, sizeof head, "#%llu %d %2s %016llx %8s ", (unsigned long long)tag, 0, pm, (unsigned long long)pc, op);
  bool pending = false;
  for (u32 k = 0; k < np; ++k) {
    const u32 kind = (u32)(kinds >> (4 * k)) & 15;
    const u64 key = pairs[2 * k], val = pairs[2 * k + 1];
    if (pending) out += kind == 3 ? " +\n" : "  +\n";
    out += head;
    if (kind == 3) {                                                     // formatVecInstTrace: the register's bytes, most significant first
      snprintf(buf, sizeof buf, "v %02x ", (u32)key); out += buf;
      for (u32 i = 0; i < vlenb; ++i) { const u32 bi = vlenb - 1 - i; snprintf(buf, sizeof buf, "%02x", (u32)(extra[bi >> 3] >> (8 * (bi & 7))) & 0xff); out += buf; }
      out += ' '; out += text;
      extra += vlenb / 8;
    } else if (kind == 2) {                                              // formatFpInstTrace
      if (rv32) { if (has_d) snprintf(buf, sizeof buf, "f %02x %016llx  ", (u32)key, (unsigned long long)val); else snprintf(buf, sizeof buf, "f %02x         %08x  ", (u32)key, (u32)val); }
      else { if (has_d) snprintf(buf, sizeof buf, "f %016llx %016llx ", (unsigned long long)key, (unsigned long long)val); else snprintf(buf, sizeof buf, "f %016llx         %08x ", (unsigned long long)key, (u32)val); }
      out += buf; out += text;
    } else {
      const char c = kind == 1 ? 'r' : kind == 4 ? 'm' : 'c';
      if (!rv32) snprintf(buf, sizeof buf, "%c %016llx %016llx ", c, (unsigned long long)key, (unsigned long long)val);
      else if (c == 'r') snprintf(buf, sizeof buf, "r %02x         %08x  ", (u32)key, (u32)val);
      else if (c == 'c' && (key >> 16) == 0) snprintf(buf, sizeof buf, "c %04x       %08x  ", (u32)key, (u32)val);
      else snprintf(buf, sizeof buf, "%c %08x   %08x  ", c, (u32)key, (u32)val);
      out += buf; out += text;
    }
    pending = true;
  }
  out += '\n';
}

extern "C" int64_t wsp_batch_trace_text(wsp_batch* b, uint32_t hart, uint64_t first_tag, char* text, size_t cap)
{
  if (!b) return fail(WSP_EINVAL, "null batch");
  const DevState& d = b->d;
  if (!d.ring) return fail(WSP_EINVAL, "trace ring is off");
  std::vector<u64> w(d.ring_cap);
  const int64_t n = wsp_batch_trace_read(b, hart, w.data(), w.size());
  if (n < 0) return n;
  const bool rv32 = (d.isa & ISA_RV32) != 0, has_d = (d.isa & ISA_D) != 0;
  std::string out;
  size_t i = 0; u64 tag = first_tag;
  while (i + 3 <= (size_t)n) {
    const u64 meta = w[i + 1], kinds = w[i + 2];
    const u32 np = (u32)(meta >> 56); const bool ldst = (meta >> 55) & 1;
    u32 nv = 0;
    for (u32 k = 0; k < np && k < 16; ++k) nv += ((kinds >> (4 * k)) & 15) == 3;
    const size_t len = 3 + 2 * (size_t)np + (ldst ? 2 : 0) + (size_t)nv * (d.vlenb / 8);
    if (np == 0 || np > 16 || i + len > (size_t)n) break;                  // a record cut off by the end of the ring is dropped
    format_record(out, w.data() + i, tag++, d.vlenb, rv32, has_d, np, ldst);
    i += len;
  }
  if (text && cap) { const size_t m = std::min(cap - 1, out.size()); memcpy(text, out.data(), m); text[m] = 0; }
  return (int64_t)out.size();
}

extern "C" int wsp_disassemble(uint32_t inst, uint32_t xlen, char* text, size_t cap)
{
  if (!text || !cap) return fail(WSP_EINVAL, "null buffer");
  const std::string s = wspdis::disassemble(inst, xlen != 32);
  snprintf(text, cap, "%s", s.c_str());
  return (int)s.size();
}

extern "C" int64_t wsp_batch_signature(wsp_batch* b, uint32_t hart, uint32_t* words, size_t cap)
{
  u64 lo = 0, hi = 0;
  int rc = wsp_batch_symbol(b, hart, "begin_signature", &lo); if (rc) return rc;
  rc = wsp_batch_symbol(b, hart, "end_signature", &hi); if (rc) return rc;
  if (hi < lo || ((hi - lo) & 3)) return fail(WSP_EFORMAT, "bad signature bounds");
  size_t n = (hi - lo) / 4;
  if (n > cap) return fail(WSP_ERANGE, "signature buffer too small");
  if (n) { rc = wsp_batch_peek(b, hart, WSP_RES_MEM, lo, words, n * 4); if (rc) return rc; }
  return (int64_t)n;
}

extern "C" int wsp_batch_snapshot(wsp_batch* b)
{
  if (!b) return fail(WSP_EINVAL, "null batch");
  HIPOK(hipSetDevice(b->device));
  SnapPart parts[WSP_SNAP_PARTS]; snap_parts(b, parts);
  for (int i = 0; i < WSP_SNAP_PARTS; ++i) {
    if (!parts[i].bytes) continue;
    if (!b->snap[i]) { if (hipMalloc(&b->snap[i], parts[i].bytes) != hipSuccess) return fail(WSP_ENOMEM, "snapshot allocation failed"); }
    HIPOK(hipMemcpyAsync(b->snap[i], parts[i].live, parts[i].bytes, hipMemcpyDeviceToDevice, b->stream));
  }
  HIPOK(hipStreamSynchronize(b->stream));
  b->snap_hart_prog = b->hart_prog; b->snap_hart_stop_pc = b->hart_stop_pc;      // the host's view of prog / stop_pc rows goes with them
  return WSP_OK;
}

extern "C" int wsp_batch_restore(wsp_batch* b)
{
  if (!b) return fail(WSP_EINVAL, "null batch");
  if (!b->snap[0]) return fail(WSP_EINVAL, "no snapshot taken");
  HIPOK(hipSetDevice(b->device));
  SnapPart parts[WSP_SNAP_PARTS]; snap_parts(b, parts);
  for (int i = 0; i < WSP_SNAP_PARTS; ++i) if (parts[i].bytes) HIPOK(hipMemcpyAsync(parts[i].live, b->snap[i], parts[i].bytes, hipMemcpyDeviceToDevice, b->stream));
  HIPOK(hipStreamSynchronize(b->stream));
  b->hart_prog = b->snap_hart_prog; b->hart_stop_pc = b->snap_hart_stop_pc;
  return WSP_OK;
}

extern "C" int wsp_decode_words_host(const char* isa, const uint32_t* words, size_t n, wsp_decoded* out)
{
  u32 f; std::string why; int rc = parse_isa(isa, &f, &why); if (rc) return fail(rc, why);
  for (size_t i = 0; i < n; ++i) { Rec r = decode(words[i], f); out[i].w0 = r.w0; out[i].imm = r.imm; out[i].raw = r.raw; out[i].pad = 0; }
  return WSP_OK;
}

extern "C" int wsp_decode_words(const char* isa, const uint32_t* words, size_t n, wsp_decoded* out)
{
  u32 f; std::string why; int rc = parse_isa(isa, &f, &why); if (rc) return fail(rc, why);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(WSP_ENODEV, "no HIP device");
  if (n == 0) return WSP_OK;
  DevBuf dw, dr;
  HIPOK(dw.alloc(n * 4)); HIPOK(dr.alloc(n * sizeof(Rec)));
  HIPOK(hipMemcpy(dw.p, words, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(wsp_decode_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dw.as<u32>(), (u64)n, f, dr.as<Rec>());
  HIPOK(hipGetLastError()); HIPOK(hipDeviceSynchronize());
  HIPOK(hipMemcpy(out, dr.p, n * sizeof(Rec), hipMemcpyDeviceToHost));
  return WSP_OK;
}


extern "C" int wsp_sf_eval(uint32_t op, uint32_t rm, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t n, uint64_t* out, uint32_t* flags)
{
  if (!a || !b || !c || !out || !flags) return fail(WSP_EINVAL, "null argument");
  if (rm > 4 || (op & 0xff) > 14 || (op & ~0x3ffu)) return fail(WSP_EINVAL, "wsp_sf_eval: bad op or rounding mode");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(WSP_ENODEV, "no HIP device");
  if (n == 0) return WSP_OK;
  DevBuf da, db, dc, dout, dfl;
  HIPOK(da.alloc(n * 8)); HIPOK(db.alloc(n * 8)); HIPOK(dc.alloc(n * 8)); HIPOK(dout.alloc(n * 8)); HIPOK(dfl.alloc(n * 4));
  HIPOK(hipMemcpy(da.p, a, n * 8, hipMemcpyHostToDevice)); HIPOK(hipMemcpy(db.p, b, n * 8, hipMemcpyHostToDevice)); HIPOK(hipMemcpy(dc.p, c, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(wsp_sf_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, (int)rm, da.as<u64>(), db.as<u64>(), dc.as<u64>(), (u64)n, dout.as<u64>(), dfl.as<u32>());
  HIPOK(hipGetLastError()); HIPOK(hipDeviceSynchronize());
  HIPOK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost)); HIPOK(hipMemcpy(flags, dfl.p, n * 4, hipMemcpyDeviceToHost));
  return WSP_OK;
}
