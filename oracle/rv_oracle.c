/*
 * TEST INFRASTRUCTURE ONLY (see rv_oracle.h).  Scalar CPU restatement of the reference's
 * per-instruction path for one RV64 hart with a private RAM arena.
 *
 * Reference anchors (all under /root/reference):
 *   run loop            Hart.cpp:6003-6204 (untilAddress: the loop used whenever a trace is on)
 *   fetch               Hart.cpp:3052-3206 (fetchInstNoTrap), 5943-5998 (extra mcycle on fault)
 *   decode              Decoder.cpp:2232 (32-bit), 1455-1819 (16-bit)
 *   base I / M          Hart.cpp:1645-1801, 11594-11935, 13964-14672, 14364-14546
 *   loads / stores      Hart.cpp:1988-2154, 2302-2338, 2477-2553, 2615-2800, 13786-13935
 *   A extension         amo.cpp:28-480
 *   CSR instructions    Hart.cpp:12723-13010, 13244-13361, 13401-13748; CsRegs.cpp:1252, 3908, 4680
 *   traps / xRET        Hart.cpp:3324-3397, 3565-3929, 12247-12336, 12428-12560
 *   F/D                 float.cpp:140-3303, float-util.hpp:230-930, FpRegs.hpp:248-295
 *   commit record       printTrace.cpp:302-517, 935-1000 (logStop)
 */
#include "rv_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <dlfcn.h>

typedef uint64_t u64;
typedef int64_t  i64;
typedef uint32_t u32;
typedef int32_t  i32;

/* ------------------------------------------------------------------ hashing */
static inline u64 rotl64(u64 x, unsigned r) { return (x << r) | (x >> (64 - r)); }
u64 rvo_fold(u64 h, u64 w) { h ^= w; h = rotl64(h, 27); return h * 5 + 0x52dce729u; }
u64 rvo_mix(u64 x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
  return x;
}
#define GOLDEN 0x9E3779B97F4A7C15ULL

/* ------------------------------------------------------------------ softfloat binding */
typedef struct { uint32_t v; } sf32;
typedef struct { uint64_t v; } sf64;
static struct {
  void* lib;
  uint_fast8_t* rm;     /* softfloat_roundingMode (thread local in the library) */
  uint_fast8_t* flags;  /* softfloat_exceptionFlags */
  sf32 (*f32_add)(sf32, sf32); sf32 (*f32_sub)(sf32, sf32); sf32 (*f32_mul)(sf32, sf32);
  sf32 (*f32_div)(sf32, sf32); sf32 (*f32_sqrt)(sf32); sf32 (*f32_mulAdd)(sf32, sf32, sf32);
  sf64 (*f64_add)(sf64, sf64); sf64 (*f64_sub)(sf64, sf64); sf64 (*f64_mul)(sf64, sf64);
  sf64 (*f64_div)(sf64, sf64); sf64 (*f64_sqrt)(sf64); sf64 (*f64_mulAdd)(sf64, sf64, sf64);
  int_fast32_t (*f32_to_i32)(sf32, uint_fast8_t, int); uint_fast32_t (*f32_to_ui32)(sf32, uint_fast8_t, int);
  int_fast64_t (*f32_to_i64)(sf32, uint_fast8_t, int); uint_fast64_t (*f32_to_ui64)(sf32, uint_fast8_t, int);
  int_fast32_t (*f64_to_i32)(sf64, uint_fast8_t, int); uint_fast32_t (*f64_to_ui32)(sf64, uint_fast8_t, int);
  int_fast64_t (*f64_to_i64)(sf64, uint_fast8_t, int); uint_fast64_t (*f64_to_ui64)(sf64, uint_fast8_t, int);
  sf32 (*i32_to_f32)(int32_t); sf32 (*ui32_to_f32)(uint32_t); sf32 (*i64_to_f32)(int64_t); sf32 (*ui64_to_f32)(uint64_t);
  sf64 (*i32_to_f64)(int32_t); sf64 (*ui32_to_f64)(uint32_t); sf64 (*i64_to_f64)(int64_t); sf64 (*ui64_to_f64)(uint64_t);
  sf64 (*f32_to_f64)(sf32); sf32 (*f64_to_f32)(sf64);
  /* accessors exported by oracle/ref_glue.cpp are not needed: TLS symbols resolve via dlsym */
} SF;

int rvo_bind_softfloat(const char* path)
{
  if (SF.lib) return 0;
  void* l = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!l) { fprintf(stderr, "rv_oracle: dlopen(%s): %s\n", path, dlerror()); return -1; }
#define B(n) do { *(void**)(&SF.n) = dlsym(l, #n); if (!SF.n) { fprintf(stderr, "rv_oracle: missing %s\n", #n); return -2; } } while (0)
  B(f32_add); B(f32_sub); B(f32_mul); B(f32_div); B(f32_sqrt); B(f32_mulAdd);
  B(f64_add); B(f64_sub); B(f64_mul); B(f64_div); B(f64_sqrt); B(f64_mulAdd);
  B(f32_to_i32); B(f32_to_ui32); B(f32_to_i64); B(f32_to_ui64);
  B(f64_to_i32); B(f64_to_ui32); B(f64_to_i64); B(f64_to_ui64);
  B(i32_to_f32); B(ui32_to_f32); B(i64_to_f32); B(ui64_to_f32);
  B(i32_to_f64); B(ui32_to_f64); B(i64_to_f64); B(ui64_to_f64);
  B(f32_to_f64); B(f64_to_f32);
#undef B
  SF.rm = (uint_fast8_t*)dlsym(l, "softfloat_roundingMode");
  SF.flags = (uint_fast8_t*)dlsym(l, "softfloat_exceptionFlags");
  if (!SF.rm || !SF.flags) { fprintf(stderr, "rv_oracle: missing softfloat globals\n"); return -3; }
  SF.lib = l;
  return 0;
}

/* Batch evaluation of the reference SoftFloat for the device softfloat sweep (tests/test_gpu_softfloat.py).  Same op numbering as
 * wsp_sf_eval (include/whisper_amd.h): op & 0x100 = double; 0 add 1 sub 2 mul 3 div 4 sqrt 5 mulAdd, 6..9 to int (bit 0 signed,
 * bit 1 64-bit), 10..13 from int c (same bits), 14 convert to the other format.  rm 0..4 in RISC-V order (float-util.hpp:80-91). */
int rvo_sf_ref(uint32_t op, uint32_t rm, const u64* a, const u64* b, const u64* c, size_t n, u64* out, uint32_t* flags)
{
  static const uint8_t rm_sf[5] = { 0 /*near_even*/, 1 /*minMag*/, 2 /*min*/, 3 /*max*/, 4 /*near_maxMag*/ };
  if (!SF.lib || rm > 4) return -1;
  const int dbl = (op & 0x100) != 0; const unsigned k = op & 0xff;
  for (size_t i = 0; i < n; ++i) {
    *SF.rm = rm_sf[rm]; *SF.flags = 0;
    u64 r = 0;
    sf32 x = { (uint32_t)a[i] }, y = { (uint32_t)b[i] }, z = { (uint32_t)c[i] };
    sf64 X = { a[i] }, Y = { b[i] }, Z = { c[i] };
    switch (k) {
      case 0: r = dbl ? SF.f64_add(X, Y).v : SF.f32_add(x, y).v; break;
      case 1: r = dbl ? SF.f64_sub(X, Y).v : SF.f32_sub(x, y).v; break;
      case 2: r = dbl ? SF.f64_mul(X, Y).v : SF.f32_mul(x, y).v; break;
      case 3: r = dbl ? SF.f64_div(X, Y).v : SF.f32_div(x, y).v; break;
      case 4: r = dbl ? SF.f64_sqrt(X).v : SF.f32_sqrt(x).v; break;
      case 5: r = dbl ? SF.f64_mulAdd(X, Y, Z).v : SF.f32_mulAdd(x, y, z).v; break;
      case 6: r = dbl ? (u64)(uint32_t)SF.f64_to_ui32(X, rm_sf[rm], 1) : (u64)(uint32_t)SF.f32_to_ui32(x, rm_sf[rm], 1); break;
      case 7: r = dbl ? (u64)(i64)(int32_t)SF.f64_to_i32(X, rm_sf[rm], 1) : (u64)(i64)(int32_t)SF.f32_to_i32(x, rm_sf[rm], 1); break;
      case 8: r = dbl ? (u64)SF.f64_to_ui64(X, rm_sf[rm], 1) : (u64)SF.f32_to_ui64(x, rm_sf[rm], 1); break;
      case 9: r = dbl ? (u64)SF.f64_to_i64(X, rm_sf[rm], 1) : (u64)SF.f32_to_i64(x, rm_sf[rm], 1); break;
      case 10: r = dbl ? SF.ui32_to_f64((uint32_t)c[i]).v : SF.ui32_to_f32((uint32_t)c[i]).v; break;
      case 11: r = dbl ? SF.i32_to_f64((int32_t)c[i]).v : SF.i32_to_f32((int32_t)c[i]).v; break;
      case 12: r = dbl ? SF.ui64_to_f64(c[i]).v : SF.ui64_to_f32(c[i]).v; break;
      case 13: r = dbl ? SF.i64_to_f64((i64)c[i]).v : SF.i64_to_f32((i64)c[i]).v; break;
      case 14: r = dbl ? SF.f64_to_f32(X).v : SF.f32_to_f64(x).v; break;
      default: return -2;
    }
    out[i] = r; flags[i] = *SF.flags;
  }
  return 0;
}

/* ------------------------------------------------------------------ hart state */
enum { PRV_U = 0, PRV_S = 1, PRV_M = 3 };

/* CSR numbers */
enum {
  C_FFLAGS = 0x001, C_FRM = 0x002, C_FCSR = 0x003,
  C_VSTART = 0x008, C_VXSAT = 0x009, C_VXRM = 0x00A, C_VCSR = 0x00F, C_VL = 0xC20, C_VTYPE = 0xC21, C_VLENB = 0xC22,
  C_SSTATUS = 0x100, C_SIE = 0x104, C_STVEC = 0x105, C_SCOUNTEREN = 0x106, C_SENVCFG = 0x10A,
  C_SSCRATCH = 0x140, C_SEPC = 0x141, C_SCAUSE = 0x142, C_STVAL = 0x143, C_SIP = 0x144, C_SATP = 0x180,
  C_MSTATUS = 0x300, C_MISA = 0x301, C_MEDELEG = 0x302, C_MIDELEG = 0x303, C_MIE = 0x304, C_MTVEC = 0x305,
  C_MCOUNTEREN = 0x306, C_MENVCFG = 0x30A, C_MCOUNTINHIBIT = 0x320,
  C_MSCRATCH = 0x340, C_MEPC = 0x341, C_MCAUSE = 0x342, C_MTVAL = 0x343, C_MIP = 0x344,
  C_PMPCFG0 = 0x3A0, C_PMPADDR0 = 0x3B0,
  C_MCYCLE = 0xB00, C_MINSTRET = 0xB02,
  C_CYCLE = 0xC00, C_TIME = 0xC01, C_INSTRET = 0xC02,
  C_MVENDORID = 0xF11, C_MARCHID = 0xF12, C_MIMPID = 0xF13, C_MHARTID = 0xF14, C_MCONFIGPTR = 0xF15
};

#define MST_SIE   (1ULL << 1)
#define MST_MIE   (1ULL << 3)
#define MST_SPIE  (1ULL << 5)
#define MST_MPIE  (1ULL << 7)
#define MST_SPP   (1ULL << 8)
#define MST_VS_SH 9
#define MST_MPP_SH 11
#define MST_FS_SH 13
#define MST_XS_SH 15
#define MST_MPRV  (1ULL << 17)
#define MST_SUM   (1ULL << 18)
#define MST_MXR   (1ULL << 19)
#define MST_TVM   (1ULL << 20)
#define MST_TW    (1ULL << 21)
#define MST_TSR   (1ULL << 22)
#define MST_SD    (1ULL << 63)
/* CsRegs.cpp:4704 (write mask), 4979/4990 (sstatus write / read masks) */
#define MSTATUS_WMASK 0x7FFFAAULL
#define SSTATUS_WMASK 0x000DE722ULL
#define SSTATUS_RMASK 0x80000003000DE762ULL

/* TlbEntry (Tlb.hpp:28-46) without the STEE world id and VMID */
struct rvo_tlbe { u64 vpn, ppn; u32 asid; uint8_t counter, valid, global, user, read, write, exec, accessed, dirty, level; };

struct rvo_hart {
  rvo_config cfg;
  int rv32;                  /* System<uint32_t> (Session.cpp:1373-1424): x registers are held sign-extended in 64 bits, the pc zero-extended */
  uint8_t* mem;
  u64 x[32], f[32];
  u64 pc, curr_pc;
  unsigned priv;
  /* CSRs */
  u64 mstatus, medeleg, mideleg, mie, mip, mtvec, mcounteren, menvcfg, mcountinhibit;
  u64 mscratch, mepc, mcause, mtval;
  u64 stvec, scounteren, senvcfg, sscratch, sepc, scause, stval, satp;
  u64 fcsr;
  u64 cycle, instret, time_;
  u64 max_ret, stop_pc; int stop_pc_valid;   /* retCountLim_, untilAddress' address */
  /* run bookkeeping (Hart.hpp:6707-6717) */
  u64 exec_count, retire_count;
  u64 exec_last_ill, consec_ill;
  int status; u64 stop_value;
  u32 htif_dev;
  /* reservation (Memory.hpp:450-543) */
  int lr_valid; u64 lr_addr; unsigned lr_size;
  u64 pmpcfg[8];            /* pmpcfg0,2,..,14 (RV64: even numbered only, CsRegs.cpp:4777-4784) */
  u64 pmpaddr[64];
  int pmp_on;               /* some entry is not Off (cache over pmpcfg, kept by csr_store) */
  /* TLB (Tlb.hpp:28-46), direct mapped, tlb_n entries (Hart.cpp:106: 2048 by default) */
  struct rvo_tlbe* tlb; unsigned tlb_n;
  /* vector unit (VecRegs.hpp): 32 registers of vlenb bytes; vtype reset = vill (CsRegs.cpp:5363) */
  uint8_t vreg[32 * 64]; unsigned vlenb;
  u64 vstart, vl, vtype; unsigned vxsat, vxrm;
  /* per-instruction trace capture */
  int has_exc;
  int ch_ireg; int ch_freg;
  int ch_vreg; int ch_vgroup;       /* first register (operand 0) and register count of the vector group written */
  int ch_mem; u64 ch_mem_addr, ch_mem_val;
  int ld_has; u64 ld_va, ld_pa;     /* ldStSize_ != 0, ldStAddr_, ldStPhysAddr1_ */
  unsigned ch_csr[8]; int ch_ncsr;
  u64 trace_hash;
};

static inline int ext(const rvo_hart* h, unsigned e) { return (h->cfg.isa & e) != 0; }

rvo_hart* rvo_create(const rvo_config* cfg)
{
  if ((cfg->xlen != 64 && cfg->xlen != 32) || (cfg->mem_size & 7) || cfg->mem_size == 0) return NULL;
  /* RV32 is modelled for M / U mode integer, A, F, D and C code: no S mode (Sv32) and no V */
  if (cfg->xlen == 32 && ((cfg->isa & (RVO_ISA_S | RVO_ISA_V)) || cfg->mem_base + cfg->mem_size > (1ULL << 32))) return NULL;
  rvo_hart* h = (rvo_hart*)calloc(1, sizeof(*h));
  h->cfg = *cfg;
  h->rv32 = cfg->xlen == 32;
  h->mem = (uint8_t*)calloc(1, cfg->mem_size);
  h->priv = PRV_M;
  /* reset mstatus: MPP=M, SXL=UXL=2 (CsRegs.cpp:4705-4710); FS/VS Off */
  h->mstatus = (3ULL << MST_MPP_SH) | (h->rv32 ? 0 : (0xAULL << 32));   /* RV32: no SXL / UXL fields (mstatus = 0x1800) */
  h->pc = cfg->reset_pc;
  h->vlenb = cfg->vlenb ? cfg->vlenb : 64;                                   /* Hart.cpp:1081 */
  if (h->vlenb != 16 && h->vlenb != 32 && h->vlenb != 64) { free(h->mem); free(h); return NULL; }
  h->vtype = 1ULL << 63;
  h->tlb_n = cfg->tlb_entries ? cfg->tlb_entries : 2048;
  if (h->tlb_n & (h->tlb_n - 1)) { free(h->mem); free(h); return NULL; }   /* Tlb.cpp:11 */
  h->tlb = (struct rvo_tlbe*)calloc(h->tlb_n, sizeof(struct rvo_tlbe));
  h->max_ret = ~0ULL;
  return h;
}

void rvo_destroy(rvo_hart* h) { if (h) { free(h->mem); free(h->tlb); free(h); } }

static inline int in_mem(const rvo_hart* h, u64 a, u64 n)
{ return a >= h->cfg.mem_base && a - h->cfg.mem_base <= h->cfg.mem_size - n && n <= h->cfg.mem_size; }

int rvo_write_mem(rvo_hart* h, u64 addr, const void* src, size_t n)
{ if (!in_mem(h, addr, n)) return -1; memcpy(h->mem + (addr - h->cfg.mem_base), src, n); return 0; }
int rvo_read_mem(const rvo_hart* h, u64 addr, void* dst, size_t n)
{ if (!in_mem(h, addr, n)) return -1; memcpy(dst, h->mem + (addr - h->cfg.mem_base), n); return 0; }
void rvo_set_pc(rvo_hart* h, u64 pc) { h->pc = pc & ~1ULL; if (h->rv32) h->pc &= 0xffffffffULL; }
u64  rvo_get_pc(const rvo_hart* h) { return h->pc; }
void rvo_set_x(rvo_hart* h, unsigned i, u64 v) { if (i && i < 32) h->x[i] = h->rv32 ? (u64)(i64)(int32_t)v : v; }
u64  rvo_get_x(const rvo_hart* h, unsigned i) { return h->rv32 ? (uint32_t)h->x[i & 31] : h->x[i & 31]; }
void rvo_set_f(rvo_hart* h, unsigned i, u64 v) { h->f[i & 31] = v; }
u64  rvo_get_f(const rvo_hart* h, unsigned i) { return h->f[i & 31]; }
unsigned rvo_priv(const rvo_hart* h) { return h->priv; }
unsigned rvo_get_vreg(const rvo_hart* h, unsigned i, void* dst) { memcpy(dst, h->vreg + (size_t)(i & 31) * h->vlenb, h->vlenb); return h->vlenb; }
void rvo_set_vreg(rvo_hart* h, unsigned i, const void* src) { memcpy(h->vreg + (size_t)(i & 31) * h->vlenb, src, h->vlenb); }
int  rvo_status(const rvo_hart* h) { return h->status; }
u64  rvo_stop_value(const rvo_hart* h) { return h->stop_value; }
u64  rvo_exec_count(const rvo_hart* h) { return h->exec_count; }
u64  rvo_retired_count(const rvo_hart* h) { return h->retire_count; }
u64  rvo_trace_hash(const rvo_hart* h) { return h->trace_hash; }

static u64 ld_le(const uint8_t* p, unsigned n) { u64 v = 0; memcpy(&v, p, n); return v; }
static void st_le(uint8_t* p, u64 v, unsigned n) { memcpy(p, &v, n); }

/* ------------------------------------------------------------------ CSR file */
static u64 misa_value(void) { return 0x800000000034112dULL; }  /* CsRegs.cpp:4721-4724: reset, read-only; RV32 reads 0x4034112d */

/* CsRegs::legalizeMstatus (CsRegs.cpp:2617-2648) */
static u64 legalize_mstatus(const rvo_hart* h, u64 v)
{
  unsigned fs = (v >> MST_FS_SH) & 3, xs = (v >> MST_XS_SH) & 3, vs = (v >> MST_VS_SH) & 3;
  if (fs == 3 || xs == 3 || vs == 3) v |= MST_SD; else v &= ~MST_SD;
  unsigned mpp = (v >> MST_MPP_SH) & 3, spp = (v >> 8) & 1;
  int s = ext(h, RVO_ISA_S), u = ext(h, RVO_ISA_U);
  if (!s) spp = 0;
  if (mpp == PRV_S && !s) mpp = PRV_U;
  if (mpp == 2) mpp = PRV_U;
  if (mpp == PRV_U && !u) mpp = PRV_M;
  v = (v & ~(3ULL << MST_MPP_SH)) | ((u64)mpp << MST_MPP_SH);
  v = (v & ~MST_SPP) | ((u64)spp << 8);
  return v;
}

static u64 mstatus_read_mask(const rvo_hart* h)
{
  /* CsRegs::enableRvf (CsRegs.cpp:1977-1987): FS reads as zero unless F or S enabled. */
  u64 m = ~0ULL;
  if (!ext(h, RVO_ISA_F) && !ext(h, RVO_ISA_S)) m &= ~(3ULL << MST_FS_SH);
  return m;
}

/* CY/TM/IR of m/scounteren read as zero while cycle/time/instret are not implemented (CsRegs.cpp:1524-1539, 2117-2130). */
static u64 counteren_rmask(const rvo_hart* h) { return ext(h, RVO_ISA_ZICNTR) ? ~0ULL : ~7ULL; }

/* CsRegs::enableRvc (CsRegs.hpp:2119-2127): xEPC bit 0 reads zero with C, bits 1:0 without; the stored value keeps bit 1. */
static u64 epc_rmask(const rvo_hart* h) { return ext(h, RVO_ISA_C) ? ~1ULL : ~3ULL; }

/* CsRegs::legalizeEnvcfg (CsRegs.cpp:6719-6727): a reserved CBIE value (2) keeps the old field. */
static u64 legalize_envcfg(u64 prev, u64 next) { return ((next >> 4) & 3) == 2 ? (next & ~0x30ULL) | (prev & 0x30) : next; }

static void record_csr(rvo_hart* h, unsigned num)
{
  for (int i = 0; i < h->ch_ncsr; ++i) if (h->ch_csr[i] == num) return;   /* CsRegs.cpp:4673 */
  if (h->ch_ncsr < 8) h->ch_csr[h->ch_ncsr++] = num;
}

static void write_mstatus_raw(rvo_hart* h, u64 v) { h->mstatus = v; record_csr(h, C_MSTATUS); }

/* Is CSR `num` implemented for this configuration? 1 = modelled, 0 = not implemented (illegal),
 * -1 = implemented by the reference but outside the modelled subset. */
static int csr_class(const rvo_hart* h, unsigned num)
{
  int s = ext(h, RVO_ISA_S);
  switch (num) {
    case C_FFLAGS: case C_FRM: case C_FCSR: return ext(h, RVO_ISA_F) ? 1 : 0;
    case C_SSTATUS: case C_SIE: case C_STVEC: case C_SCOUNTEREN: case C_SENVCFG: case C_SSCRATCH:
    case C_SEPC: case C_SCAUSE: case C_STVAL: case C_SIP: case C_SATP:
    case C_MEDELEG: case C_MIDELEG: return s ? 1 : 0;                      /* CsRegs.cpp:1488-1492 */
    case C_MSTATUS: case C_MISA: case C_MIE: case C_MTVEC: case C_MCOUNTEREN: case C_MENVCFG:
    case C_MCOUNTINHIBIT: case C_MSCRATCH: case C_MEPC: case C_MCAUSE: case C_MTVAL: case C_MIP:
    case C_MCYCLE: case C_MINSTRET: case C_MVENDORID: case C_MARCHID: case C_MIMPID: case C_MHARTID:
    case C_MCONFIGPTR: return 1;
    case C_CYCLE: case C_TIME: case C_INSTRET: return ext(h, RVO_ISA_ZICNTR) ? 1 : 0;  /* CsRegs.cpp:2102 */
    case C_VSTART: case C_VXSAT: case C_VXRM: case C_VCSR: case C_VL: case C_VTYPE: case C_VLENB: return ext(h, RVO_ISA_V) ? 1 : 0;   /* CsRegs::enableVector (CsRegs.cpp:2240) */
    default: break;
  }
  if (h->rv32) {   /* CsRegs<uint32_t>: the high halves exist, and so do the odd pmpcfg registers (pinned by tests/test_oracle_rv32.py) */
    if (num == 0x310 || num == 0x31A || num == 0xB80 || num == 0xB82) return 1;          /* mstatush menvcfgh mcycleh minstreth */
    if (num >= 0xC80 && num <= 0xC82) return ext(h, RVO_ISA_ZICNTR) ? 1 : 0;             /* cycleh timeh instreth */
    if (num >= 0x3A0 && num <= 0x3EF) return 1;
    if ((num >= 0x723 && num <= 0x73F) || num == 0x757 || (num >= 0xB83 && num <= 0xB9F)) return 1;      /* mhpmevent<n>h mseccfgh mhpmcounter<n>h: hardwired zero */
  }
  /* Implemented in the reference's default configuration but not modelled here. */
  if (num >= 0x3A0 && num <= 0x3EF) return (num < 0x3B0 && (num & 1)) ? 0 : 1;   /* pmpcfg (even only on rv64), pmpaddr0..63 */
  /* mhpmcounter3..31, mhpmevent3..31, mseccfg: present with all-zero write masks while no performance counters and no Smepmp
   * are configured (CsRegs.cpp:4810, 4848-4866): read zero, a write changes nothing (and is logged like any other) */
  if ((num >= 0xB03 && num <= 0xB1F) || (num >= 0x323 && num <= 0x33F) || num == 0x747) return 1;
  if (num == 0x7C7) return -1;                      /* c_matp: a custom register of the reference, not modelled */
  return 0;                                         /* dcsr / dpc / dscratch: debug mode only (CsRegs.cpp:1252): illegal here */
}

static u64 csr_peek64(const rvo_hart* h, unsigned num)
{
  switch (num) {
    case C_FFLAGS: return h->fcsr & 0x1f;
    case C_FRM: return (h->fcsr >> 5) & 7;
    case C_FCSR: return h->fcsr;
    case C_VSTART: return h->vstart; case C_VXSAT: return h->vxsat; case C_VXRM: return h->vxrm;
    case C_VCSR: return (h->vxrm << 1) | h->vxsat;                         /* CsRegs.cpp:4556-4606 */
    case C_VL: return h->vl; case C_VTYPE: return h->vtype; case C_VLENB: return h->vlenb;
    case C_SSTATUS: return h->mstatus & SSTATUS_RMASK;
    case C_SIE: return h->mie & h->mideleg & ~0x1444ULL;                /* CsRegs::readSie (CsRegs.cpp:195-229) */
    case C_SIP: return h->mip & h->mideleg & ~0x1444ULL;
    case C_STVEC: return h->stvec; case C_SCOUNTEREN: return h->scounteren & counteren_rmask(h); case C_SENVCFG: return h->senvcfg;
    case C_SSCRATCH: return h->sscratch; case C_SEPC: return h->sepc & epc_rmask(h); case C_SCAUSE: return h->scause;
    case C_STVAL: return h->stval; case C_SATP: return h->satp;
    case C_MSTATUS: return h->mstatus & mstatus_read_mask(h);
    case C_MISA: return misa_value();
    case C_MEDELEG: return h->medeleg; case C_MIDELEG: return h->mideleg; case C_MIE: return h->mie;
    case C_MTVEC: return h->mtvec; case C_MCOUNTEREN: return h->mcounteren & counteren_rmask(h); case C_MENVCFG: return h->menvcfg;
    case C_MCOUNTINHIBIT: return h->mcountinhibit;
    case C_MSCRATCH: return h->mscratch; case C_MEPC: return h->mepc & epc_rmask(h); case C_MCAUSE: return h->mcause;
    case C_MTVAL: return h->mtval; case C_MIP: return h->mip;
    case C_MCYCLE: case C_CYCLE: return h->cycle;
    case C_MINSTRET: case C_INSTRET: return h->instret;
    case C_TIME: return h->time_;
    case C_MHARTID: return h->cfg.hart_id;
    default:
      if (num >= 0x3A0 && num <= 0x3AF) return h->pmpcfg[(num - 0x3A0) >> 1];
      if (num >= 0x3B0 && num <= 0x3EF) return h->pmpaddr[num - 0x3B0];    /* grain G = 0: read value = stored value */
      return 0;
  }
}

/* The CSR file is kept in its RV64 shape; an RV32 hart sees 32-bit views of it (CsRegs<uint32_t>): the low half of each
 * register with mstatus.SD and the interrupt bit of the cause registers at bit 31, the high halves under their own numbers,
 * pmpcfg<2k+1> = the upper half of row k. */
static u64 csr_peek(const rvo_hart* h, unsigned num)
{
  if (!h->rv32) return csr_peek64(h, num);
  switch (num) {
    case C_MSTATUS: case C_SSTATUS: case C_MCAUSE: case C_SCAUSE: { u64 v = csr_peek64(h, num); return (v & 0x7fffffffULL) | ((v >> 63) << 31); }
    case C_MISA: return 0x4034112dULL;
    case 0x310: return 0;                                                    /* mstatush: MBE / SBE hardwired zero */
    case 0x31A: return csr_peek64(h, C_MENVCFG) >> 32;
    case 0xB80: case 0xC80: return h->cycle >> 32;
    case 0xB82: case 0xC82: return h->instret >> 32;
    case 0xC81: return h->time_ >> 32;
    default:
      if (num >= 0x3A0 && num <= 0x3AF) return (h->pmpcfg[(num - 0x3A0) >> 1] >> (32 * (num & 1))) & 0xffffffffULL;
      return csr_peek64(h, num) & 0xffffffffULL;
  }
}

int rvo_peek_csr(const rvo_hart* h, unsigned num, u64* v)
{ if (csr_class(h, num) != 1) return -1; *v = csr_peek(h, num); return 0; }


/* ------------------------------------------------------------------ PMP (PmpManager.hpp, Hart.cpp:842-899) */
static unsigned pmp_cfg_byte(const rvo_hart* h, unsigned ix) { return (unsigned)(h->pmpcfg[ix >> 3] >> (8 * (ix & 7))) & 0xff; }

/* PmpManager::legalizePmpcfg (PmpManager.hpp:480-551) without Smepmp, grain G = 0, TOR and NA4 enabled */
static u64 legalize_pmpcfg(u64 prev, u64 next)
{
  u64 legal = 0;
  for (unsigned i = 0; i < 8; ++i) {
    unsigned pb = (unsigned)(prev >> (8 * i)) & 0xff, nb = (unsigned)(next >> (8 * i)) & 0xff;
    if (pb >> 7) nb = pb;                                   /* locked: keep */
    else if ((nb & 3) == 2) nb = (pb & 7) | (nb & ~7u);     /* w=1 r=0 reserved: keep old xwr */
    legal |= (u64)nb << (8 * i);
  }
  return legal;
}

/* CsRegs::isPmpaddrLocked (CsRegs.cpp:6773-6793) */
static int pmpaddr_locked(const rvo_hart* h, unsigned ix)
{
  if (pmp_cfg_byte(h, ix) & 0x80) return 1;
  if (ix >= 63) return 0;
  unsigned nb = pmp_cfg_byte(h, ix + 1);
  return (nb & 0x80) && ((nb >> 3) & 3) == 1;
}

/* Region of entry ix: PmpManager::unpackMemoryProtection (PmpManager.hpp:417-468) + defineRegion word alignment (:233-236).
 * Returns 0 when the entry is off. */
static int pmp_region(const rvo_hart* h, unsigned ix, u64* a0, u64* a1)
{
  unsigned cfg = pmp_cfg_byte(h, ix), type = (cfg >> 3) & 3;
  if (!type) return 0;
  u64 v = h->pmpaddr[ix], lo, hi;
  if (type == 1) {                                          /* TOR */
    lo = (ix ? h->pmpaddr[ix - 1] : 0) << 2;
    hi = v << 2; if (hi > 0) hi -= 1;
  } else {
    u64 size_m1 = 3, napot = v;
    if (type == 3) {
      unsigned rzi;
      if (v == ~0ULL) { napot = 0; rzi = 64; }
      else { rzi = (unsigned)__builtin_ctzll(~v); napot = (napot >> rzi) << rzi; }
      size_m1 = rzi + 3 >= 64 ? ~0ULL : ((1ULL << (rzi + 3)) - 1);
    }
    lo = napot << 2; hi = lo + size_m1;
  }
  *a0 = (lo >> 2) << 2; *a1 = ((hi >> 2) << 2) + 3;
  return 1;
}

/* PmpManager::getPmp (PmpManager.hpp:155-177): r=1 w=2 x=4 for privilege pm at addr.  First matching entry in index
 * order; M mode ignores entries that are not locked; no match: M everything, S/U nothing. */
unsigned rvo_pmp_perm(const rvo_hart* h, unsigned pm, u64 addr)
{
  if (!h->pmp_on) return pm == PRV_M ? 7 : 0;
  for (unsigned ix = 0; ix < 64; ++ix) {
    u64 a0, a1;
    if (!pmp_region(h, ix, &a0, &a1)) continue;
    if (addr >= a0 && addr <= a1) {
      unsigned cfg = pmp_cfg_byte(h, ix);
      if (pm == PRV_M && !(cfg & 0x80)) return 7;
      return cfg & 7;
    }
  }
  return pm == PRV_M ? 7 : 0;
}

/* Poke = host-side write with the write mask applied, no side effects on counters. */
static void csr_store(rvo_hart* h, unsigned num, u64 v, int record);
int rvo_poke_csr(rvo_hart* h, unsigned num, u64 v)
{ if (csr_class(h, num) != 1) return -1; csr_store(h, num, v, 0); h->ch_ncsr = 0; return 0; }

/* M ecall (11), double trap (16), reserved (17) hardwired zero (CsRegs.cpp:4726-4732); VS ecall (10) and the guest
 * page fault / virtual instruction causes (23:20) read as zero without H (CsRegs.cpp:1780-1788). */
#define MEDELEG_MASK (~((1ULL << 11) | (1ULL << 16) | (1ULL << 17) | (0xFULL << 20) | (1ULL << 10)))
#define MIE_WMASK(h) (0x999ULL | (ext(h, RVO_ISA_S) ? 0x222ULL : 0))
#define MIP_WMASK(h) (ext(h, RVO_ISA_S) ? 0x222ULL : 0)           /* CsRegs.cpp:4770 read-only + S bits (CsRegs.cpp:1509-1521) */
/* Value update of CsRegs::write (CsRegs.cpp:3908-4165) for the modelled subset. */
static void csr_store(rvo_hart* h, unsigned num, u64 v, int record)
{
  if (h->rv32) {                                       /* a 32-bit write, brought into the RV64 shape of the file */
    v &= 0xffffffffULL;
    switch (num) {
      case C_MCAUSE: case C_SCAUSE: v = (v & 0x7fffffffULL) | ((v >> 31) << 63); break;
      case C_MENVCFG: v |= h->menvcfg & ~0xffffffffULL; break;
      case 0x31A: num = C_MENVCFG; v = (v << 32) | (h->menvcfg & 0xffffffffULL); if (record) record_csr(h, 0x31A); record = 0; break;
      case 0x310: if (record) record_csr(h, num); return;                    /* mstatush: nothing writable */
      case C_MCYCLE: v |= h->cycle & ~0xffffffffULL; break;
      case C_MINSTRET: v |= h->instret & ~0xffffffffULL; break;
      case 0xB80: h->cycle = (h->cycle & 0xffffffffULL) | (v << 32); if (record) record_csr(h, num); return;
      case 0xB82: h->instret = (h->instret & 0xffffffffULL) | (v << 32); if (record) record_csr(h, num); return;
      default:
        if (num >= 0x3A0 && num <= 0x3AF) {              /* pmpcfg<n>: four entries, half of row n / 2 */
          const unsigned k = (num - 0x3A0) >> 1, sh = 32 * (num & 1);
          const u64 keep = h->pmpcfg[k] & ~(0xffffffffULL << sh);
          h->pmpcfg[k] = legalize_pmpcfg(h->pmpcfg[k], keep | (v << sh)) & 0x9f9f9f9f9f9f9f9fULL;
          h->pmp_on = 0; for (unsigned i = 0; i < 8; ++i) if (h->pmpcfg[i] & 0x1818181818181818ULL) h->pmp_on = 1;
          if (record) record_csr(h, num);
          return;
        }
        break;
    }
  }
  switch (num) {
    case C_FFLAGS: h->fcsr = (h->fcsr & ~0x1fULL) | (v & 0x1f); break;         /* CsRegs.cpp:4462-4472 */
    case C_FRM:    h->fcsr = (h->fcsr & ~0xe0ULL) | ((v & 7) << 5); break;
    case C_FCSR:   h->fcsr = v & 0xff; break;
    case C_VSTART: h->vstart = v & (h->vlenb * 8 - 1); break;                   /* mask = VLEN - 1 (Hart.cpp:1088-1103) */
    case C_VXSAT:  h->vxsat = v & 1; break;
    case C_VXRM:   h->vxrm = v & 3; break;
    case C_VCSR:   h->vxsat = v & 1; h->vxrm = (v >> 1) & 3; break;
    case C_VL:     h->vl = v; break;                                            /* poke only (write mask 0) */
    case C_VTYPE:  h->vtype = v & 0x80000000000000ffULL; break;
    case C_MSTATUS: {                                                           /* CsRegs.cpp:3964-3976 */
      u64 nv = (h->mstatus & ~MSTATUS_WMASK) | (v & MSTATUS_WMASK & mstatus_read_mask(h));   /* value &= writeMask & readMask */
      h->mstatus = legalize_mstatus(h, nv);
      break; }
    case C_SSTATUS: {
      u64 m = SSTATUS_WMASK;
      u64 nv = (h->mstatus & ~m) | (v & m);
      h->mstatus = legalize_mstatus(h, nv);
      break; }
    case C_MISA: break;                                                         /* write mask 0 */
    case C_MEDELEG: h->medeleg = v & MEDELEG_MASK; break;
    case C_MIDELEG: h->mideleg = v & ~0x1444ULL; break;       /* CsRegs.cpp:4734 all bits; VS/SGEI read-only zero without H (CsRegs.cpp:1766-1771) */
    case C_MIE: { u64 m = MIE_WMASK(h);   /* CsRegs.cpp:4739: 0xfff less S/VS bits; legacy U bits 0,4,8 writable */ h->mie = (h->mie & ~m) | (v & m); break; }
    case C_MIP: { u64 m = MIP_WMASK(h); h->mip = (h->mip & ~m) | (v & m); break; }
    case C_SIE: { u64 m = 0x222ULL & MIE_WMASK(h) & h->mideleg; h->mie = (h->mie & ~m) | (v & m); break; }   /* CsRegs::writeSie (CsRegs.cpp:2745-2790) */
    case C_SIP: { u64 m = 0x002ULL & MIP_WMASK(h) & h->mideleg; h->mip = (h->mip & ~m) | (v & m); break; }   /* CsRegs::writeSip (CsRegs.cpp:2680-2740) */
    case C_MTVEC: h->mtvec = v & ~2ULL; break;
    case C_STVEC: h->stvec = v & ~2ULL; break;
    case C_MCOUNTEREN: h->mcounteren = v & 0xffffffffULL; break;
    case C_SCOUNTEREN: h->scounteren = v; break;
    case C_MENVCFG: h->menvcfg = legalize_envcfg(h->menvcfg, v) & 0xe0000003000000f5ULL; break;
    case C_SENVCFG: h->senvcfg = legalize_envcfg(h->senvcfg, v) & 0x00000003000000f5ULL; break;
    case C_MCOUNTINHIBIT: h->mcountinhibit = v & 0xfffffffdULL; break;
    case C_MSCRATCH: h->mscratch = v; break;
    case C_SSCRATCH: h->sscratch = v; break;
    case C_MEPC: h->mepc = v & ~1ULL; break;
    case C_SEPC: h->sepc = v & ~1ULL; break;
    case C_MCAUSE: h->mcause = v; break;
    case C_SCAUSE: h->scause = v; break;
    case C_MTVAL: h->mtval = v; break;
    case C_STVAL: h->stval = v; break;
    case C_SATP: {                                                              /* Hart::updateAddressTranslation (Hart.cpp:904-919) -> VirtMem::configTranslation -> Tlb::setMode (Tlb.hpp:378-383) */
      if ((h->satp >> 60) != (v >> 60)) for (unsigned i = 0; i < h->tlb_n; ++i) { h->tlb[i].valid = 0; h->tlb[i].counter = 0; }
      h->satp = v; break; }
    case C_MCYCLE: h->cycle = v; break;
    case C_MINSTRET: h->instret = v; break;
    default:
      if (num >= 0x3A0 && num <= 0x3AF) { unsigned k = (num - 0x3A0) >> 1; h->pmpcfg[k] = legalize_pmpcfg(h->pmpcfg[k], v) & 0x9f9f9f9f9f9f9f9fULL;
        h->pmp_on = 0; for (unsigned i = 0; i < 8; ++i) if (h->pmpcfg[i] & 0x1818181818181818ULL) h->pmp_on = 1; }
      else if (num >= 0x3B0 && num <= 0x3EF) { if (!pmpaddr_locked(h, num - 0x3B0)) h->pmpaddr[num - 0x3B0] = v & 0x003fffffffffffffULL; }
      break;
  }
  if (record) record_csr(h, num);
}

/* ------------------------------------------------------------------ trace capture */
static void begin_inst(rvo_hart* h)
{
  h->has_exc = 0; h->ch_ireg = -1; h->ch_freg = -1; h->ch_mem = 0; h->ch_ncsr = 0; h->ch_vreg = -1; h->ch_vgroup = 0; h->ld_has = 0;
}

static inline void wx(rvo_hart* h, unsigned rd, u64 v)
{ if (rd) h->x[rd] = h->rv32 ? (u64)(i64)(int32_t)v : v; h->ch_ireg = (int)rd; }      /* IntRegs.hpp:58-64: x0 write still recorded */

static void mark_fs_dirty(rvo_hart* h)
{
  /* Hart::setFpStatus / markFsDirty (float.cpp:165-191): write only when FS changes. */
  if (((h->mstatus >> MST_FS_SH) & 3) != 3) {
    u64 v = h->mstatus | (3ULL << MST_FS_SH);
    v |= MST_SD;
    write_mstatus_raw(h, v);
  }
}

static inline void wf(rvo_hart* h, unsigned rd, u64 v) { h->f[rd] = v; h->ch_freg = (int)rd; }

static void finish_commit(rvo_hart* h, u32 inst, unsigned size, unsigned priv, int trap,
                          rvo_commit* log, size_t cap, size_t* n)
{
  rvo_commit c; memset(&c, 0, sizeof c);
  c.tag = h->exec_count; c.pc = h->curr_pc; c.inst = inst; c.size = (uint8_t)size;
  c.priv = (uint8_t)priv; c.trap = (uint8_t)trap;
  unsigned k = 0;
  if (h->ch_ireg > 0) { c.ch[k].kind = RVO_CH_R; c.ch[k].key = (u64)h->ch_ireg; c.ch[k].val = h->rv32 ? (uint32_t)h->x[h->ch_ireg] : h->x[h->ch_ireg]; ++k; }
  if (h->ch_freg >= 0) { c.ch[k].kind = RVO_CH_F; c.ch[k].key = (u64)h->ch_freg; c.ch[k].val = h->f[h->ch_freg]; ++k; }
  /* vector group (printTrace.cpp:424-441; Hart::lastVecReg Hart.cpp:5753): one v line per register of the group starting at
   * operand 0; the record carries a digest of the register's bytes */
  if (h->ch_vgroup) for (int r = 0; r < h->ch_vgroup && k < RVO_MAX_CHANGES; ++r) {
    unsigned reg = (unsigned)(h->ch_vreg + r) & 31; u64 d = 0;
    for (unsigned w = 0; w < h->vlenb / 8; ++w) d = rvo_fold(d, ld_le(h->vreg + (size_t)reg * h->vlenb + 8 * w, 8));
    c.ch[k].kind = RVO_CH_V; c.ch[k].key = reg; c.ch[k].val = d; ++k;
  }
  if (h->ch_mem) { c.ch[k].kind = RVO_CH_M; c.ch[k].key = h->ch_mem_addr; c.ch[k].val = h->ch_mem_val; ++k; }
  /* CSRs sorted by number (printTrace.cpp:479-489) */
  for (int i = 0; i < h->ch_ncsr; ++i)
    for (int j = i + 1; j < h->ch_ncsr; ++j)
      if (h->ch_csr[j] < h->ch_csr[i]) { unsigned t = h->ch_csr[i]; h->ch_csr[i] = h->ch_csr[j]; h->ch_csr[j] = t; }
  for (int i = 0; i < h->ch_ncsr && k < RVO_MAX_CHANGES; ++i) {
    c.ch[k].kind = RVO_CH_C; c.ch[k].key = h->ch_csr[i]; c.ch[k].val = csr_peek(h, h->ch_csr[i]); ++k;
  }
  if (k == 0) { c.ch[0].kind = RVO_CH_R; c.ch[0].key = 0; c.ch[0].val = 0; k = 1; }  /* printTrace.cpp:491-493 */
  c.nchanges = (uint8_t)k;
  c.has_ld = (uint32_t)h->ld_has; c.ld_va = h->ld_va; c.ld_pa = h->ld_pa;
  u64 hh = h->trace_hash;
  hh = rvo_fold(hh, c.pc);
  hh = rvo_fold(hh, (u64)c.inst | ((u64)c.priv << 32) | ((u64)c.trap << 40) | ((u64)c.size << 48));
  for (unsigned i = 0; i < k; ++i) {
    hh = rvo_fold(hh, c.ch[i].key ^ ((u64)c.ch[i].kind << 60));
    hh = rvo_fold(hh, c.ch[i].val);
  }
  h->trace_hash = hh;
  if (log && n && *n < cap) log[(*n)++] = c;
}

/* ------------------------------------------------------------------ traps */
static void stop(rvo_hart* h, int status, u64 value) { h->status = status; h->stop_value = value; }

/* Hart::initiateException + initiateTrap (Hart.cpp:3324-3397, 3565-3929), exceptions only.
 * `counts` = the reference passed no DecodedInst (di == nullptr) or an illegal one, so the
 * consecutive-illegal heuristic applies (true for every call site modelled here). */
static void take_exception_ex(rvo_hart* h, unsigned cause, u64 epc, u64 tval, int counts)
{
  if (counts) {
    if (h->exec_count == h->exec_last_ill + 1) h->consec_ill++; else h->consec_ill = 0;
    if (h->consec_ill > 16) { stop(h, RVO_STOP_ILLEGAL, 3); return; }
    h->exec_last_ill = h->exec_count;
  }
  h->has_exc = 1;
  if (ext(h, RVO_CFG_CANCEL_LR_TRAP)) h->lr_valid = 0;        /* cancelLrOnTrap_ (Hart.cpp:3570-3571) */

  unsigned next = PRV_M;
  if (ext(h, RVO_ISA_S) && h->priv != PRV_M && ((h->medeleg >> cause) & 1)) next = PRV_S;
  unsigned orig = h->priv;
  h->priv = next;
  u64 tvec;
  if (next == PRV_M) {
    csr_store(h, C_MEPC, epc & ~1ULL, 1);
    csr_store(h, C_MCAUSE, cause, 1);
    csr_store(h, C_MTVAL, tval, 1);
    u64 v = h->mstatus;
    v = (v & ~(3ULL << MST_MPP_SH)) | ((u64)orig << MST_MPP_SH);
    v = (v & ~MST_MPIE) | ((v & MST_MIE) ? MST_MPIE : 0);
    v &= ~MST_MIE;
    write_mstatus_raw(h, v);
    tvec = h->mtvec;
  } else {
    csr_store(h, C_SEPC, epc & ~1ULL, 1);
    csr_store(h, C_SCAUSE, cause, 1);
    csr_store(h, C_STVAL, tval, 1);
    u64 v = h->mstatus;
    v = (v & ~MST_SPP) | ((u64)(orig & 1) << 8);
    v = (v & ~MST_SPIE) | ((v & MST_SIE) ? MST_SPIE : 0);
    v &= ~MST_SIE;
    h->mstatus = v; record_csr(h, C_SSTATUS);
    tvec = h->stvec;
  }
  h->pc = (tvec & ~3ULL) & ~1ULL;
}

/* Hart::isInterruptPossible (Hart.cpp:7028-7083, 7135-7157) without H/AIA/NMI: pending = mip (no SEI pin,
 * no timer source in a bare system), M-destined = mip & mie & ~mideleg, S-destined = mip & mie & mideleg. */
static int interrupt_possible(const rvo_hart* h, unsigned* cause, unsigned* next)
{
  static const unsigned char m_order[] = { 11, 3, 7, 9, 1, 5, 12, 13 };                 /* mInterrupts_ (Hart.cpp:141-143) */
  static const unsigned char s_order[] = { 11, 3, 7, 9, 1, 5, 12, 10, 2, 6, 13 };       /* sInterrupts_ (Hart.cpp:146-149) */
  u64 pend = h->mip & h->mie;
  if (!pend) return 0;
  u64 mdest = pend & ~h->mideleg;
  if (((h->mstatus & MST_MIE) || h->priv != PRV_M) && mdest)
    for (unsigned i = 0; i < sizeof m_order; ++i)
      if ((mdest >> m_order[i]) & 1) { *cause = m_order[i]; *next = PRV_M; return 1; }
  if (h->priv == PRV_M) return 0;
  u64 sdest = pend & h->mideleg;
  if (((h->mstatus & MST_SIE) || h->priv == PRV_U) && sdest)
    for (unsigned i = 0; i < sizeof s_order; ++i)
      if ((sdest >> s_order[i]) & 1) { *cause = s_order[i]; *next = PRV_S; return 1; }
  return 0;
}

/* Hart::initiateInterrupt -> initiateTrap(interrupt=true) (Hart.cpp:3287-3300, 3565-3905). */
static void take_interrupt(rvo_hart* h, unsigned cause, unsigned next, u64 epc)
{
  unsigned orig = h->priv;
  if (ext(h, RVO_CFG_CANCEL_LR_TRAP)) h->lr_valid = 0;        /* Hart.cpp:3570-3571 */
  h->priv = next;
  u64 cv = (1ULL << 63) | cause, tvec;
  if (next == PRV_M) {
    csr_store(h, C_MEPC, epc & ~1ULL, 1); csr_store(h, C_MCAUSE, cv, 1); csr_store(h, C_MTVAL, 0, 1);
    u64 v = h->mstatus;
    v = (v & ~(3ULL << MST_MPP_SH)) | ((u64)orig << MST_MPP_SH);
    v = (v & ~MST_MPIE) | ((v & MST_MIE) ? MST_MPIE : 0);
    v &= ~MST_MIE;
    write_mstatus_raw(h, v);
    tvec = h->mtvec;
  } else {
    csr_store(h, C_SEPC, epc & ~1ULL, 1); csr_store(h, C_SCAUSE, cv, 1); csr_store(h, C_STVAL, 0, 1);
    u64 v = h->mstatus;
    v = (v & ~MST_SPP) | ((u64)(orig & 1) << 8);
    v = (v & ~MST_SPIE) | ((v & MST_SIE) ? MST_SPIE : 0);
    v &= ~MST_SIE;
    h->mstatus = v; record_csr(h, C_SSTATUS);
    tvec = h->stvec;
  }
  u64 base = tvec & ~3ULL;
  h->pc = (tvec & 3) == 1 ? base + 4ULL * cause : base;      /* vectored mode applies to interrupts only (Hart.cpp:3849-3853) */
}

/* di == nullptr call sites: illegal, ecall, ebreak, fetch faults, misaligned targets. */
static void take_exception(rvo_hart* h, unsigned cause, u64 epc, u64 tval) { take_exception_ex(h, cause, epc, tval, 1); }
/* load/store/AMO faults pass their DecodedInst (Hart.cpp:1972-1983): heuristic untouched. */
static void ldst_exception(rvo_hart* h, unsigned cause, u64 tval) { take_exception_ex(h, cause, h->curr_pc, tval, 0); }
static void illegal(rvo_hart* h, u32 inst) { take_exception(h, 2, h->curr_pc, ext(h, RVO_CFG_CLR_MTVAL_ILL) ? 0 : inst); }   /* clearMtvalOnIllInst_ (Hart.cpp:3256) */

/* ------------------------------------------------------------------ memory access checks */
/* PMP with no programmed entries: M-mode allowed, S/U denied (PmpManager.hpp:155-177). */
static inline int pmp_ok(const rvo_hart* h, unsigned priv, u64 addr, unsigned need) { return (rvo_pmp_perm(h, priv, addr) & need) != 0; }

static unsigned eff_ldst_priv(const rvo_hart* h)
{
  if ((h->mstatus & MST_MPRV) && h->priv == PRV_M) return (h->mstatus >> MST_MPP_SH) & 3;  /* Hart.hpp:3444-3450 */
  return h->priv;
}

/* ------------------------------------------------------------------ Sv39 + TLB (virtual_memory/VirtMem.cpp, Tlb.hpp, Tlb.cpp)
 * Single-stage Sv39 as the reference runs it by default: no Svadu (faultOnFirstAccess_ = true, VirtMem.hpp:1159),
 * no Svpbmt, no Svnapot, no shadow stack.  The 8192-entry PTE cache (VirtMem.hpp:769-832) is coherent with stores
 * and flushed on sfence/satp/PMP changes, so it is not modelled.  Sv48/Sv57 stop the hart as unsupported. */
static inline int paging_on(const rvo_hart* h, unsigned pm) { return ext(h, RVO_ISA_S) && pm != PRV_M && (h->satp >> 60) == 8; }
static inline unsigned pf_cause(unsigned acc) { return acc == 4 ? 12 : acc == 1 ? 13 : 15; }   /* stage1PageFaultType (VirtMem.cpp:31-40) */
static inline unsigned af_cause(unsigned acc) { return acc == 4 ? 1 : acc == 1 ? 5 : 7; }      /* accessFaultType (VirtMem.cpp:57-66) */

/* VirtMem::pageTableWalk<Pte39, Va39> (VirtMem.cpp:694-892) behind translateNoTlb's canonical-address check (:393-397).
 * ad_check = accessDirtyCheck_.  Fills *e on success. */
static unsigned page_walk(rvo_hart* h, u64 va, unsigned pm, unsigned acc, int ad_check, u64* pa, struct rvo_tlbe* e)
{
  const unsigned pf = pf_cause(acc);
  if ((u64)(((i64)va << 25) >> 25) != va) return pf;
  const int sum = (h->mstatus & MST_SUM) != 0, mxr = (h->mstatus & MST_MXR) != 0;
  u64 root = (h->satp & 0xfffffffffffULL) << 12, pte;
  int ii = 2, global = 0;
  for (;;) {
    u64 pte_addr = root + ((va >> (12 + 9 * ii)) & 0x1ff) * 8;
    /* isAddrReadable: PMP as Supervisor (Hart.cpp:424-441), then memRead: PMA read (Hart.cpp:337-339) */
    if (!pmp_ok(h, PRV_S, pte_addr, 1) || !in_mem(h, pte_addr, 8)) return af_cause(acc);
    pte = ld_le(h->mem + (pte_addr - h->cfg.mem_base), 8);
    if (pte >> 63) return pf;                                                     /* napotCheck without Svnapot (VirtMem.hpp:1090) */
    /* isValidPte (VirtMem.hpp:558): V, not (W without R), reserved bits 60:54 zero.  The xwr = 010 encoding that
     * isSsPte lets through for fetches (VirtMem.cpp:740-743) faults below all the same. */
    if (!(pte & 1) || (!(pte & 2) && (pte & 4)) || ((pte >> 54) & 0x7f)) return pf;
    global |= (pte >> 5) & 1;
    if (!(pte & 0xe)) {                                                           /* pointer to the next level (:747-756) */
      if ((pte & 0xd0) || ((pte >> 61) & 3)) return pf;                           /* A, D, U, PBMT must be zero */
      if (--ii < 0) return pf;
      root = ((pte >> 10) & 0xfffffffffffULL) << 12;
      continue;
    }
    if ((pte >> 61) & 3) return pf;                                               /* Svpbmt off (:772-773) */
    const int u = (pte >> 4) & 1, r = (pte >> 1) & 1, w = (pte >> 2) & 1, x = (pte >> 3) & 1;
    if (pm == PRV_U && !u) return pf;
    if (pm == PRV_S && u && (!sum || acc == 4)) return pf;
    if ((acc == 1 && !(r || (mxr && x))) || (acc == 2 && !w) || (acc == 4 && !x)) return pf;
    for (int j = 0; j < ii; ++j)                                                  /* misaligned superpage (:804-806) */
      if ((pte >> (10 + 9 * j)) & 0x1ff) return pf;
    if (ad_check && (!((pte >> 6) & 1) || (acc == 2 && !((pte >> 7) & 1)))) return pf;   /* faultOnFirstAccess_ (:809-814) */
    break;
  }
  /* computePhysAddr (VirtMem.cpp:665-689) */
  u64 a = va & 0xfff;
  for (int j = 0; j < ii; ++j) a |= ((va >> (12 + 9 * j)) & 0x1ff) << (12 + 9 * j);
  for (int j = ii; j < 3; ++j) a |= ((pte >> (10 + 9 * j)) & (j == 2 ? 0x3ffffffULL : 0x1ffULL)) << (12 + 9 * j);
  *pa = a;
  memset(e, 0, sizeof *e);
  e->vpn = va >> 12; e->ppn = a >> 12; e->asid = (u32)((h->satp >> 44) & 0xffff); e->valid = 1; e->global = (uint8_t)global;
  e->user = (pte >> 4) & 1; e->read = (pte >> 1) & 1; e->write = (pte >> 2) & 1; e->exec = (pte >> 3) & 1;
  e->accessed = (pte >> 6) & 1; e->dirty = (pte >> 7) & 1; e->level = (uint8_t)(1 + ii);
  return 0;
}

/* Tlb::findEntry (Tlb.hpp:63-71) */
static struct rvo_tlbe* tlb_find(rvo_hart* h, u64 vpn)
{
  struct rvo_tlbe* e = &h->tlb[vpn & (h->tlb_n - 1)];
  u32 asid = (u32)((h->satp >> 44) & 0xffff);
  return (e->valid && e->vpn == vpn && (e->global || e->asid == asid)) ? e : NULL;
}

/* VirtMem::translate (VirtMem.cpp:261-329), single stage.  Returns 0 or the exception cause. */
static unsigned translate(rvo_hart* h, u64 va, unsigned pm, unsigned acc, u64* pa)
{
  const unsigned pf = pf_cause(acc);
  *pa = va;
  struct rvo_tlbe* e = tlb_find(h, va >> 12);
  if (e) {
    e->counter = (e->counter + 1) & 3;                                            /* findEntryUpdateTime (Tlb.hpp:91-100) */
    const int sum = (h->mstatus & MST_SUM) != 0, mxr = (h->mstatus & MST_MXR) != 0;
    if (pm == PRV_U && !e->user) return pf;
    if (pm == PRV_S && e->user && (acc == 4 || !sum)) return pf;
    if ((acc == 1 && !(e->read || (mxr && e->exec))) || (acc == 2 && !e->write) || (acc == 4 && !e->exec)) return pf;
    if (!e->accessed || (acc == 2 && !e->dirty)) e->valid = 0;                    /* :311-312: forget the entry and walk again */
    else { *pa = (e->ppn << 12) | (va & 0xfff); return 0; }
  }
  struct rvo_tlbe ne;
  unsigned cause = page_walk(h, va, pm, acc, 1, pa, &ne);
  if (cause) { *pa = va; return cause; }
  struct rvo_tlbe* slot = &h->tlb[ne.vpn & (h->tlb_n - 1)];                       /* Tlb::insertEntry (Tlb.cpp:74-92) */
  if (slot->valid && (slot->counter & 2)) slot->counter--;
  else *slot = ne;
  return 0;
}
unsigned rvo_translate(rvo_hart* h, u64 va, unsigned pm, unsigned acc, u64* pa) { if (!paging_on(h, pm)) { *pa = va; return 0; } return translate(h, va, pm, acc, pa); }

/* VirtMem::transAddrNoUpdate (VirtMem.cpp:143-205): no TLB update, no A/D check. */
static unsigned translate_noupdate(rvo_hart* h, u64 va, unsigned pm, unsigned acc, u64* pa)
{
  const unsigned pf = pf_cause(acc);
  *pa = va;
  struct rvo_tlbe* e = tlb_find(h, va >> 12);
  if (e) {
    const int sum = (h->mstatus & MST_SUM) != 0, mxr = (h->mstatus & MST_MXR) != 0;
    if (pm == PRV_U && !e->user) return pf;
    if (pm == PRV_S && e->user && (acc == 4 || !sum)) return pf;
    if ((acc == 1 && !(e->read || (mxr && e->exec))) || (acc == 2 && !e->write) || (acc == 4 && !e->exec)) return pf;
    *pa = (e->ppn << 12) | (va & 0xfff);
    return 0;
  }
  struct rvo_tlbe ne;
  unsigned cause = page_walk(h, va, pm, acc, 0, pa, &ne);
  if (cause) *pa = va;
  return cause;
}

unsigned rvo_translate_noupdate(rvo_hart* h, u64 va, unsigned pm, unsigned acc, u64* pa) { if (!paging_on(h, pm)) { *pa = va; return 0; } return translate_noupdate(h, va, pm, acc, pa); }

/* Tlb::sizeIn4kBytes / alignPageNumBySize for Sv39 (Tlb.hpp:389-456); Bare: size 0 */
static u64 tlb_size4k(const rvo_hart* h, unsigned level) { return (h->satp >> 60) != 8 ? 0 : level <= 1 ? 1 : level == 2 ? 512 : 262144; }
static u64 tlb_align(const rvo_hart* h, u64 vpn, unsigned level) { return ((h->satp >> 60) != 8 || level <= 1) ? vpn : level == 2 ? (vpn >> 9) << 9 : (vpn >> 18) << 18; }

/* execSfence_vma's TLB part (Hart.cpp:12139-12173) over Tlb::invalidate / invalidateAsid / invalidateVirtualPage /
 * invalidateVirtualPageAsid (Tlb.hpp:176-284, 368-375).  Entries are matched whether or not they are valid. */
void rvo_sfence(rvo_hart* h, int has_addr, u64 addr, int has_asid, u64 asid_v)
{
  const u32 asid = (u32)asid_v;
  const u64 vpn = addr >> 12;
  if (!has_addr) {
    for (unsigned i = 0; i < h->tlb_n; ++i) {
      struct rvo_tlbe* e = &h->tlb[i];
      if (!has_asid || (!e->global && e->asid == asid)) { e->valid = 0; e->counter = 0; }
    }
    return;
  }
  unsigned max_size = 0; u64 vpn_of_max = 0;
  for (unsigned i = 0; i < h->tlb_n; ++i) {
    struct rvo_tlbe* e = &h->tlb[i];
    u64 size = tlb_size4k(h, e->level), pn = tlb_align(h, e->vpn, e->level);
    if (pn <= vpn && vpn < pn + size && (!has_asid || (e->asid == asid && !e->global))) {
      if (size > max_size) { max_size = (unsigned)size; vpn_of_max = e->vpn; }
      e->valid = 0; e->counter = 0;
    }
  }
  if (max_size > 1)                                                               /* "invalidate subpages covered by super-page" */
    for (unsigned i = 0; i < h->tlb_n; ++i) {
      struct rvo_tlbe* e = &h->tlb[i];
      if (vpn_of_max <= vpn && vpn < vpn_of_max + max_size && (!has_asid || (e->asid == asid && !e->global))) { e->valid = 0; e->counter = 0; }
    }
}

/* Same two-word packing as the device keeps in HBM (wsp_core.h) */
u64 rvo_tlb_hash(const rvo_hart* h)
{
  u64 s = 0;
  for (unsigned i = 0; i < h->tlb_n; ++i) {
    const struct rvo_tlbe* e = &h->tlb[i];
    u64 fl = (u64)e->valid | (u64)e->global << 1 | (u64)e->user << 2 | (u64)e->read << 3 | (u64)e->write << 4 | (u64)e->exec << 5 |
             (u64)e->accessed << 6 | (u64)e->dirty << 7;
    s = rvo_fold(s, (e->vpn & 0xfffffffffffffULL) | (u64)(e->level & 3) << 52 | (u64)(e->counter & 3) << 54 | fl << 56);
    s = rvo_fold(s, (e->ppn & 0xfffffffffffULL) | (u64)(e->asid & 0xffff) << 44);
  }
  return s;
}

/* determineLoadException / determineStoreException (Hart.cpp:1988-2154, 13786-13935).  need: 1 load, 2 store/amo.
 * Returns 0 and the physical addresses of the (up to two) parts, or the exception cause with *tval. */
static unsigned ldst_addr(rvo_hart* h, u64 va, unsigned size, unsigned need, u64* pa1, u64* pa2, u64* tval)
{
  const unsigned pm = eff_ldst_priv(h), af = need == 1 ? 5 : 7;
  const int tr = paging_on(h, pm);
  const u64 mask = size - 1;
  const int misal = (va & mask) != 0;
  const u64 va2 = misal ? ((va + size - 1) & ~mask) : va;
  unsigned c;
  *pa1 = va;
  if (misal && ext(h, RVO_CFG_NO_MISAL)) { *tval = va; return need == 1 ? 4 : 6; }   /* misalHasPriority_ and not misalDataOk_ (Hart.cpp:2013-2016, 13810-13813) */
  if (tr && (c = translate(h, va, pm, need, pa1)) != 0) { *tval = va; return c; }
  if (!misal) {
    if (!pmp_ok(h, pm, *pa1, need) || !in_mem(h, *pa1, size)) { *tval = va; return af; }
    *pa2 = *pa1;
    return 0;
  }
  const int cross = (va >> 12) != (va2 >> 12);
  *pa2 = (*pa1 + size - 1) & ~mask;
  if (cross && tr && (c = translate(h, va2, pm, need, pa2)) != 0) { *tval = va2; return c; }
  const unsigned n1 = (unsigned)(((*pa1 + size - 1) & ~mask) - *pa1);              /* lower part stops at the alignment boundary (:2068-2069) */
  if (!pmp_ok(h, pm, *pa1, need) || !in_mem(h, *pa1, n1)) { *tval = va; return af; }
  if (!pmp_ok(h, pm, *pa2, need) || !in_mem(h, *pa2, size)) { *tval = va2; return af; }
  if (!cross) *pa2 = *pa1;
  return 0;
}

/* Hart::memRead / memWrite over a possibly page-crossing item (Hart.hpp:3469-3567) */
static u64 mem_read2(const rvo_hart* h, u64 pa1, u64 pa2, unsigned size)
{
  if (pa1 == pa2) return ld_le(h->mem + (pa1 - h->cfg.mem_base), size);
  unsigned size1 = size - (unsigned)(pa1 & (size - 1));
  u64 v = 0;
  for (unsigned i = 0; i < size; ++i) {
    u64 a = i < size1 ? pa1 + i : pa2 + (i - size1);
    v |= (u64)h->mem[a - h->cfg.mem_base] << (8 * i);
  }
  return v;
}
static void mem_write2(rvo_hart* h, u64 pa1, u64 pa2, unsigned size, u64 v)
{
  if (pa1 == pa2) { st_le(h->mem + (pa1 - h->cfg.mem_base), v, size); return; }
  unsigned size1 = size - (unsigned)(pa1 & (size - 1));
  for (unsigned i = 0; i < size; ++i) {
    u64 a = i < size1 ? pa1 + i : pa2 + (i - size1);
    h->mem[a - h->cfg.mem_base] = (uint8_t)(v >> (8 * i));
  }
}

/* satp.MODE Sv48 / Sv57 are outside the modelled subset */
static int ldst_translate_unsupported(rvo_hart* h)
{
  if (ext(h, RVO_ISA_S) && eff_ldst_priv(h) != PRV_M && (h->satp >> 60) > 8) { stop(h, RVO_STOP_UNSUPPORTED, 0x180); return 1; }
  return 0;
}

/* Hart::load (Hart.cpp:2302) + readForLoad (2477). Returns 1 on success. */
static int do_load(rvo_hart* h, u64 va, unsigned size, u64* out)
{
  u64 tval, pa1, pa2;
  if (h->rv32) va &= 0xffffffffULL;
  h->ld_has = 1; h->ld_va = va; h->ld_pa = va;
  if (ldst_translate_unsupported(h)) return 0;
  unsigned c = ldst_addr(h, va, size, 1, &pa1, &pa2, &tval);
  if (c) { ldst_exception(h, c, tval); return 0; }
  h->ld_pa = pa1;
  if (h->cfg.tohost_valid && pa1 == h->cfg.tohost) { *out = 0; return 1; }   /* Hart.cpp:2509-2513 */
  *out = mem_read2(h, pa1, pa2, size);
  return 1;
}

/* Hart::handleStoreToHost (Hart.cpp:2615-2667) */
static void store_tohost(rvo_hart* h, u64 pa, u64 val, unsigned size)
{
  st_le(h->mem + (pa - h->cfg.mem_base), val, size);
  u64 v = val;
  if (size == 4) {
    if (pa == h->cfg.tohost + 4) { h->htif_dev = (u32)val; return; }
    v |= (u64)h->htif_dev << 32;
  } else if (size == 8)
    h->htif_dev = 0;
  unsigned cmd = (v >> 48) & 0xff, dev = (v >> 56) & 0xff;
  if (dev == 0 && cmd == 0 && (val & 1)) stop(h, RVO_STOP_TOHOST, v);
  /* dev==1 console output / getc have no architectural effect on a batched hart. */
}

/* Hart::store (Hart.cpp:2674) + writeForStore (2727). Returns 1 on success. */
static int do_store(rvo_hart* h, u64 va, unsigned size, u64 val)
{
  u64 tval, pa1, pa2;
  if (h->rv32) va &= 0xffffffffULL;
  h->ld_has = 1; h->ld_va = va; h->ld_pa = va;
  if (ldst_translate_unsupported(h)) return 0;
  unsigned c = ldst_addr(h, va, size, 2, &pa1, &pa2, &tval);
  if (c) { ldst_exception(h, c, tval); return 0; }
  h->ld_pa = pa1;
  u64 sv = size == 8 ? val : (val & ((1ULL << (8 * size)) - 1));
  h->ch_mem = 1; h->ch_mem_addr = va; h->ch_mem_val = sv;                    /* the log's m line carries the virtual address (printTrace.cpp:457) */
  if (h->cfg.tohost_valid && (pa1 == h->cfg.tohost || pa1 == h->cfg.tohost + 4)) { store_tohost(h, pa1, sv, size); return 1; }
  mem_write2(h, pa1, pa2, size, sv);
  return 1;
}

/* ------------------------------------------------------------------ FP helpers */
#define F32_CANON 0x7fc00000u
#define F64_CANON 0x7ff8000000000000ULL
static inline int fp_legal(const rvo_hart* h) { return ext(h, RVO_ISA_F) && ((h->mstatus >> MST_FS_SH) & 3) != 0; }
static inline int dp_legal(const rvo_hart* h) { return ext(h, RVO_ISA_D) && fp_legal(h); }
static inline u32 rd_s(const rvo_hart* h, unsigned r)      /* FpRegs::readSingle (FpRegs.hpp:285-295) */
{ u64 v = h->f[r]; if (ext(h, RVO_ISA_D) && (v >> 32) != 0xffffffffu) return F32_CANON; return (u32)v; }
static inline void wr_s(rvo_hart* h, unsigned r, u32 v)    /* NaN-boxing when flen==64 (FpRegs.hpp:248-269) */
{ wf(h, r, ext(h, RVO_ISA_D) ? (0xffffffff00000000ULL | v) : (u64)v); }
static inline int is_nan32(u32 a) { return (a & 0x7f800000u) == 0x7f800000u && (a & 0x7fffffu); }
static inline int is_snan32(u32 a) { return is_nan32(a) && !(a & 0x400000u); }
static inline int is_nan64(u64 a) { return (a & 0x7ff0000000000000ULL) == 0x7ff0000000000000ULL && (a & 0xfffffffffffffULL); }
static inline int is_snan64(u64 a) { return is_nan64(a) && !(a & 0x8000000000000ULL); }

static const uint8_t rm_to_sf[5] = { 0 /*near_even*/, 1 /*minMag*/, 2 /*min*/, 3 /*max*/, 4 /*near_maxMag*/ };

/* checkRoundingModeCommon (float.cpp:200-216): returns 0 if the effective mode is invalid. */
static int fp_begin(rvo_hart* h, unsigned rm)
{
  if (rm == 7) rm = (h->fcsr >> 5) & 7;
  if (rm >= 5) return 0;
  *SF.flags = 0; *SF.rm = rm_to_sf[rm];
  return 1;
}

/* softfloat flag bits (softfloat.h): inexact 1, underflow 2, overflow 4, infinite 8, invalid 16
 * -> RISC-V fflags NX 1, UF 2, OF 4, DZ 8, NV 16 (float-util.hpp:161-170): identical layout. */
static void or_fflags(rvo_hart* h, unsigned flags)
{
  u64 prev = h->fcsr & 0x1f, val = prev | (flags & 0x1f);
  if (val != prev) { h->fcsr = (h->fcsr & ~0x1fULL) | val; record_csr(h, C_FCSR); }   /* Hart.hpp:3821-3828 */
  if (flags) mark_fs_dirty(h);                                                          /* float.cpp:158-161 */
}
static void fp_accrue(rvo_hart* h) { or_fflags(h, *SF.flags); }

static u32 neg32(u32 a)   /* doNegate (float-util.hpp:743-758) */
{
  if ((a & 0x7fffffffu) == 0) return a ^ 0x80000000u;
  sf32 z = { 0 }, x = { a }; return SF.f32_sub(z, x).v;
}
static u64 neg64(u64 a)
{
  if ((a & 0x7fffffffffffffffULL) == 0) return a ^ 0x8000000000000000ULL;
  sf64 z = { 0 }, x = { a }; return SF.f64_sub(z, x).v;
}

static unsigned fclass32(u32 a)
{
  unsigned s = a >> 31, e = (a >> 23) & 0xff; u32 m = a & 0x7fffff;
  if (e == 0xff) return m ? ((m & 0x400000) ? 0x200 : 0x100) : (s ? 1 : 0x80);
  if (e == 0) return m ? (s ? 4 : 0x20) : (s ? 8 : 0x10);
  return s ? 2 : 0x40;
}
static unsigned fclass64(u64 a)
{
  unsigned s = (unsigned)(a >> 63), e = (a >> 52) & 0x7ff; u64 m = a & 0xfffffffffffffULL;
  if (e == 0x7ff) return m ? ((m & 0x8000000000000ULL) ? 0x200 : 0x100) : (s ? 1 : 0x80);
  if (e == 0) return m ? (s ? 4 : 0x20) : (s ? 8 : 0x10);
  return s ? 2 : 0x40;
}
/* ordered compare of non-NaN values by sign-magnitude */
static int lt32(u32 a, u32 b)
{
  int sa = a >> 31, sb = b >> 31;
  if (((a | b) & 0x7fffffffu) == 0) return 0;
  if (sa != sb) return sa;
  return sa ? (a > b) : (a < b);
}
static int eq32(u32 a, u32 b) { return a == b || ((a | b) & 0x7fffffffu) == 0; }
static int lt64(u64 a, u64 b)
{
  int sa = (int)(a >> 63), sb = (int)(b >> 63);
  if (((a | b) & 0x7fffffffffffffffULL) == 0) return 0;
  if (sa != sb) return sa;
  return sa ? (a > b) : (a < b);
}
static int eq64(u64 a, u64 b) { return a == b || ((a | b) & 0x7fffffffffffffffULL) == 0; }

/* ------------------------------------------------------------------ CSR instructions */
static int csr_readable(const rvo_hart* h, unsigned num)
{
  unsigned need = (num >> 8) & 3;
  if (need == 2) need = 3;
  if (h->priv < need) return 0;
  if (num == C_CYCLE || num == C_TIME || num == C_INSTRET) {      /* CsRegs::updateCounterPrivilege */
    unsigned bit = num - C_CYCLE;
    if (h->priv < PRV_M && !((h->mcounteren >> bit) & 1)) return 0;
    if (h->priv == PRV_U && ext(h, RVO_ISA_S) && !((h->scounteren >> bit) & 1)) return 0;
  }
  return 1;
}
static int csr_writeable(const rvo_hart* h, unsigned num)
{
  if (((num >> 10) & 3) == 3) return 0;       /* read-only address space */
  return csr_readable(h, num);
}

/* execCsrr* (Hart.cpp:13401-13748) -> doCsrRead (12987) / doCsrWrite (13244). op: 1 rw, 2 rs, 3 rc;
 * src is the register value or zero-extended immediate; src_nonzero_spec = rs1/uimm field != 0. */
static void exec_csr(rvo_hart* h, u32 inst, unsigned op, unsigned rd, unsigned num, u64 src, int src_field_nz)
{
  int cls = csr_class(h, num);
  int do_read = (op != 1) || rd != 0;
  int do_write = (op == 1) || src_field_nz;
  if (cls < 0) { stop(h, RVO_STOP_UNSUPPORTED, 0x10000 | num); return; }
  if (cls == 0) { illegal(h, inst); return; }
  if ((num == C_FFLAGS || num == C_FRM || num == C_FCSR) && !fp_legal(h)) { illegal(h, inst); return; }
  if ((num == C_VSTART || num == C_VXSAT || num == C_VXRM || num == C_VCSR || num == C_VL || num == C_VTYPE || num == C_VLENB) &&
      !(ext(h, RVO_ISA_V) && ((h->mstatus >> MST_VS_SH) & 3) != 0)) { illegal(h, inst); return; }      /* Hart.cpp:12952-12958 */
  if (!csr_readable(h, num) || (do_write && !csr_writeable(h, num))) { illegal(h, inst); return; }
  if (num == C_SATP && h->priv == PRV_S && (h->mstatus & MST_TVM)) { illegal(h, inst); return; }
  u64 prev = do_read ? csr_peek(h, num) : 0;
  if (!do_write) { wx(h, rd, prev); return; }
  u64 next = op == 1 ? src : op == 2 ? (prev | src) : (prev & ~src);
  /* doCsrWrite: counters pre-increment so the loop's own increment is cancelled (13251-13255) */
  const int w_ir = (num == C_MINSTRET || (h->rv32 && num == 0xB82)) && !(h->mcountinhibit & 4), w_cy = num == C_MCYCLE || (h->rv32 && num == 0xB80);
  if (w_ir) h->instret++;
  if (w_cy) h->cycle++;
  wx(h, rd, prev);
  if (num == C_SATP) {
    unsigned mode = (unsigned)(next >> 60);
    if (mode == 8 && ext(h, RVO_CFG_NO_SV39)) goto done;   /* configAddressTranslationModes without sv39 */
    if (mode != 0 && mode != 8 && mode != 9 && mode != 10) goto done;  /* unsupported MODE: dropped (13284-13294) */
    if (mode > 8) { stop(h, RVO_STOP_UNSUPPORTED, 0x180); return; }   /* Sv48 / Sv57: not modelled */
  }
  if (num >= 0x3B0 && num < 0x3EF && !pmpaddr_locked(h, num - 0x3B0)) {
    /* postCsrUpdate (Hart.cpp:4672-4679) refreshes the region table only when the written entry itself is on: a
     * pmpaddr write under an Off entry whose successor is an active TOR entry leaves the reference with a stale lower
     * bound.  Not modelled: stop instead of diverging. */
    unsigned ix = num - 0x3B0;
    if (((pmp_cfg_byte(h, ix) >> 3) & 3) == 0 && ((pmp_cfg_byte(h, ix + 1) >> 3) & 3) == 1 && (next & 0x003fffffffffffffULL) != h->pmpaddr[ix])
    { stop(h, RVO_STOP_UNSUPPORTED, 0x181); return; }
  }
  csr_store(h, num, next, 1);
  if ((num == C_FFLAGS || num == C_FRM || num == C_FCSR)) mark_fs_dirty(h);  /* postCsrUpdate: FS dirty on fp csr write */
  if (num == C_VSTART || num == C_VXSAT || num == C_VXRM || num == C_VCSR)   /* postCsrUpdate: markVsDirty (Hart.cpp:4731-4734) */
    if (((h->mstatus >> MST_VS_SH) & 3) != 3) write_mstatus_raw(h, h->mstatus | (3ULL << MST_VS_SH) | MST_SD);
done:
  if (w_ir) h->instret--;
  if (w_cy) h->cycle--;
}

/* ------------------------------------------------------------------ 16-bit expansion */
/* Decoder::decode16 (Decoder.cpp:1455-1819) restated as expansion to the equivalent 32-bit
 * encoding (0 = illegal). Zcb/Zcmop code points are illegal here (those extensions are off). */
static u32 enc_i(u32 op, u32 f3, u32 rd, u32 rs1, i32 imm) { return ((u32)imm << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op; }
static u32 enc_s(u32 op, u32 f3, u32 rs1, u32 rs2, i32 imm)
{ u32 i = (u32)imm & 0xfff; return ((i >> 5) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | ((i & 31) << 7) | op; }
static u32 enc_r(u32 op, u32 f3, u32 f7, u32 rd, u32 rs1, u32 rs2) { return (f7 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op; }
static u32 enc_b(u32 f3, u32 rs1, u32 rs2, i32 off)
{ u32 i = (u32)off; return (((i >> 12) & 1) << 31) | (((i >> 5) & 0x3f) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (((i >> 1) & 0xf) << 8) | (((i >> 11) & 1) << 7) | 0x63; }
static u32 enc_j(u32 rd, i32 off)
{ u32 i = (u32)off; return (((i >> 20) & 1) << 31) | (((i >> 1) & 0x3ff) << 21) | (((i >> 11) & 1) << 20) | (((i >> 12) & 0xff) << 12) | (rd << 7) | 0x6f; }
static i32 sx(u32 v, unsigned bits) { return (i32)(v << (32 - bits)) >> (32 - bits); }

static u32 expand16(u32 c, int rv32)
{
  u32 q = c & 3, f3 = (c >> 13) & 7;
  u32 rdp = 8 + ((c >> 2) & 7), rs1p = 8 + ((c >> 7) & 7);
  u32 rd = (c >> 7) & 31, rs2 = (c >> 2) & 31;
  i32 imm6 = sx((((c >> 12) & 1) << 5) | ((c >> 2) & 31), 6);
  if (q == 0) {
    u32 uimm_d = (((c >> 10) & 7) << 3) | (((c >> 5) & 3) << 6);
    u32 uimm_w = (((c >> 10) & 7) << 3) | (((c >> 6) & 1) << 2) | (((c >> 5) & 1) << 6);
    switch (f3) {
      case 0: {
        u32 imm = (((c >> 11) & 3) << 4) | (((c >> 7) & 0xf) << 6) | (((c >> 6) & 1) << 2) | (((c >> 5) & 1) << 3);
        if (c == 0 || imm == 0) return 0;
        return enc_i(0x13, 0, rdp, 2, (i32)imm); }
      case 1: return enc_i(0x07, 3, rdp, rs1p, (i32)uimm_d);          /* c.fld */
      case 2: return enc_i(0x03, 2, rdp, rs1p, (i32)uimm_w);          /* c.lw */
      case 3: if (rv32) return enc_i(0x07, 2, rdp, rs1p, (i32)uimm_w);   /* c.flw (Decoder.cpp:1490-1503) */
              return enc_i(0x03, 3, rdp, rs1p, (i32)uimm_d);          /* c.ld */
      case 5: return enc_s(0x27, 3, rs1p, rdp, (i32)uimm_d);          /* c.fsd */
      case 6: return enc_s(0x23, 2, rs1p, rdp, (i32)uimm_w);          /* c.sw */
      case 7: if (rv32) return enc_s(0x27, 2, rs1p, rdp, (i32)uimm_w);   /* c.fsw (Decoder.cpp:1548-1556) */
              return enc_s(0x23, 3, rs1p, rdp, (i32)uimm_d);          /* c.sd */
      default: return 0;
    }
  }
  if (q == 1) {
    switch (f3) {
      case 0: return enc_i(0x13, 0, rd, rd, imm6);                    /* c.addi / c.nop */
      case 1:
        if (rv32) {                                                   /* c.jal (Decoder.cpp:1573-1588) */
          i32 off = sx((((c >> 12) & 1) << 11) | (((c >> 11) & 1) << 4) | (((c >> 9) & 3) << 8) | (((c >> 8) & 1) << 10) |
                       (((c >> 7) & 1) << 6) | (((c >> 6) & 1) << 7) | (((c >> 3) & 7) << 1) | (((c >> 2) & 1) << 5), 12);
          return enc_j(1, off);
        }
        if (rd == 0) return 0;
        return enc_i(0x1b, 0, rd, rd, imm6);  /* c.addiw */
      case 2: return enc_i(0x13, 0, rd, 0, imm6);                     /* c.li */
      case 3: {
        if (imm6 == 0) return 0;                                      /* reserved / c.mop */
        if (rd == 2) {
          i32 imm = sx((((c >> 12) & 1) << 9) | (((c >> 6) & 1) << 4) | (((c >> 5) & 1) << 6) | (((c >> 3) & 3) << 7) | (((c >> 2) & 1) << 5), 10);
          return enc_i(0x13, 0, 2, 2, imm);
        }
        return ((u32)imm6 << 12) | (rd << 7) | 0x37; }                /* c.lui (sign-extended to 20 bits by the shift) */
      case 4: {
        u32 f2 = (c >> 10) & 3, sh = (((c >> 12) & 1) << 5) | ((c >> 2) & 31);
        if (rv32 && f2 < 2 && (sh & 32)) return 0;                    /* Decoder.cpp:1628, 1635 */
        if (f2 == 0) return (0x00u << 26) | (sh << 20) | (rs1p << 15) | (5 << 12) | (rs1p << 7) | 0x13;
        if (f2 == 1) return (0x10u << 26) | (sh << 20) | (rs1p << 15) | (5 << 12) | (rs1p << 7) | 0x13;
        if (f2 == 2) return enc_i(0x13, 7, rs1p, rs1p, imm6);
        u32 k = (c >> 5) & 3;
        if (!((c >> 12) & 1)) {
          static const u32 f3s[4] = { 0, 4, 6, 7 };
          return enc_r(0x33, f3s[k], k == 0 ? 0x20 : 0, rs1p, rs1p, rdp);
        }
        if (rv32) return 0;                                           /* Decoder.cpp:1686 */
        if (k == 0) return enc_r(0x3b, 0, 0x20, rs1p, rs1p, rdp);     /* c.subw */
        if (k == 1) return enc_r(0x3b, 0, 0, rs1p, rs1p, rdp);        /* c.addw */
        return 0; }                                                   /* c.mul / Zcb */
      case 5: {
        i32 off = sx((((c >> 12) & 1) << 11) | (((c >> 11) & 1) << 4) | (((c >> 9) & 3) << 8) | (((c >> 8) & 1) << 10) |
                     (((c >> 7) & 1) << 6) | (((c >> 6) & 1) << 7) | (((c >> 3) & 7) << 1) | (((c >> 2) & 1) << 5), 12);
        return enc_j(0, off); }
      default: {
        i32 off = sx((((c >> 12) & 1) << 8) | (((c >> 10) & 3) << 3) | (((c >> 5) & 3) << 6) | (((c >> 3) & 3) << 1) | (((c >> 2) & 1) << 5), 9);
        return enc_b(f3 == 6 ? 0 : 1, rs1p, 0, off); }
    }
  }
  if (q == 2) {
    u32 b12 = (c >> 12) & 1;
    switch (f3) {
      case 0: if (rv32 && b12) return 0;                                                      /* Decoder.cpp:1719 */
              return ((b12 << 5 | rs2) << 20) | (rd << 15) | (1 << 12) | (rd << 7) | 0x13;   /* c.slli */
      case 1: return enc_i(0x07, 3, rd, 2, (i32)((b12 << 5) | (((c >> 5) & 3) << 3) | (((c >> 2) & 7) << 6)));   /* c.fldsp */
      case 2: if (rd == 0) return 0; return enc_i(0x03, 2, rd, 2, (i32)((b12 << 5) | (((c >> 4) & 7) << 2) | (((c >> 2) & 3) << 6)));
      case 3: if (rv32) return enc_i(0x07, 2, rd, 2, (i32)((b12 << 5) | (((c >> 4) & 7) << 2) | (((c >> 2) & 3) << 6)));   /* c.flwsp (Decoder.cpp:1746-1757) */
              if (rd == 0) return 0;
              return enc_i(0x03, 3, rd, 2, (i32)((b12 << 5) | (((c >> 5) & 3) << 3) | (((c >> 2) & 7) << 6)));
      case 4:
        if (!b12) {
          if (rs2 == 0) { if (rd == 0) return 0; return enc_i(0x67, 0, 0, rd, 0); }   /* c.jr */
          return enc_r(0x33, 0, 0, rd, 0, rs2);                                      /* c.mv */
        }
        if (rs2 == 0) { if (rd == 0) return 0x00100073; return enc_i(0x67, 0, 1, rd, 0); }  /* c.ebreak / c.jalr */
        return enc_r(0x33, 0, 0, rd, rd, rs2);                                       /* c.add */
      case 5: return enc_s(0x27, 3, 2, rs2, (i32)((((c >> 10) & 7) << 3) | (((c >> 7) & 7) << 6)));   /* c.fsdsp */
      case 6: return enc_s(0x23, 2, 2, rs2, (i32)((((c >> 9) & 0xf) << 2) | (((c >> 7) & 3) << 6)));  /* c.swsp */
      default: if (rv32) return enc_s(0x27, 2, 2, rs2, (i32)((((c >> 9) & 0xf) << 2) | (((c >> 7) & 3) << 6)));   /* c.fswsp (Decoder.cpp:1804-1815) */
               return enc_s(0x23, 3, 2, rs2, (i32)((((c >> 10) & 7) << 3) | (((c >> 7) & 7) << 6)));  /* c.sdsp */
    }
  }
  return 0;
}

/* test hook: the 32-bit encoding a 16-bit word stands for (0 = illegal) */
uint32_t rvo_expand16(uint32_t c, int rv32) { return expand16(c & 0xffff, rv32); }

/* ------------------------------------------------------------------ execute one 32-bit encoding */
typedef unsigned __int128 u128;
typedef __int128 i128;

static void exec_fp(rvo_hart* h, u32 inst, u32 raw);
static void exec_amo(rvo_hart* h, u32 inst, u32 raw);
#include "rv_oracle_vec.inc"


/* Zba / Zbb / Zbs / Zbc / Zicond (bitmanip.cpp:10-1110, Hart.cpp:14728-14760; encodings Decoder.cpp:2490-2557, 2596-2627,
 * 2771-2864 and the OP-32 block).  Returns 1 with the result in *out when `inst` is one of them and its extension is on; 0
 * otherwise (the caller goes on to the base ISA and ends in an illegal-instruction trap).  a, b: x[rs1], x[rs2] as held (an RV32
 * hart's are sign-extended; the caller's wx() brings the result back into that shape). */
static int exec_bitmanip(const rvo_hart* h, u32 inst, u64 a, u64 b, u64* out)
{
  const u32 op = inst & 0x7f, f3 = (inst >> 12) & 7, f7 = inst >> 25, imm12 = inst >> 20;
  const int rv32 = h->rv32, zba = ext(h, RVO_ISA_ZBA), zbb = ext(h, RVO_ISA_ZBB), zbs = ext(h, RVO_ISA_ZBS), zbc = ext(h, RVO_ISA_ZBC);
  const unsigned xlen = rv32 ? 32 : 64, smask = xlen - 1;
  const u64 ua = rv32 ? (u64)(u32)a : a;                       /* the register as the URV the reference reads */
  const u32 aw = (u32)a, bw = (u32)b;
  u64 r;
  if (op == 0x33) {
    const unsigned ix = (unsigned)(b & smask);
    switch (f7) {
      case 0x04: if (!(rv32 && zbb && f3 == 4 && ((inst >> 20) & 31) == 0)) return 0; r = a & 0xffff; break;         /* zext.h = pack rd, rs1, x0 on RV32 (bitmanip.cpp:435-451) */
      case 0x05:
        if (f3 >= 4) { if (!zbb) return 0;
          r = f3 == 4 ? ((i64)a < (i64)b ? a : b) : f3 == 6 ? ((i64)a > (i64)b ? a : b) : f3 == 5 ? (a < b ? a : b) : (a > b ? a : b); break; }
        if (!zbc || f3 == 0) return 0;
        { const u64 ub = rv32 ? (u64)(u32)b : b; unsigned i; r = 0;
          if (f3 == 1) { for (i = 0; i < xlen; ++i) if ((ub >> i) & 1) r ^= ua << i; }
          else if (f3 == 3) { for (i = 1; i < xlen; ++i) if ((ub >> i) & 1) r ^= ua >> (xlen - i); }
          else { for (i = 0; i < xlen; ++i) if ((ub >> i) & 1) r ^= ua >> (xlen - i - 1); } }
        break;
      case 0x07: if (!ext(h, RVO_ISA_ZICOND) || (f3 != 5 && f3 != 7)) return 0; r = ((b == 0) == (f3 == 5)) ? 0 : a; break;
      case 0x10: if (!zba || (f3 != 2 && f3 != 4 && f3 != 6)) return 0; r = (a << (f3 >> 1)) + b; break;
      case 0x14: if (!zbs || f3 != 1) return 0; r = a | (1ULL << ix); break;
      case 0x20: if (!zbb || f3 < 4 || f3 == 5) return 0; r = f3 == 4 ? a ^ ~b : f3 == 6 ? a | ~b : a & ~b; break;
      case 0x24: if (!zbs || (f3 != 1 && f3 != 5)) return 0; r = f3 == 1 ? a & ~(1ULL << ix) : (ua >> ix) & 1; break;
      case 0x30: if (!zbb || (f3 != 1 && f3 != 5)) return 0;
        if (rv32) { const u32 w = f3 == 1 ? (aw << ix) | (aw >> ((32 - ix) & 31)) : (aw >> ix) | (aw << ((32 - ix) & 31)); r = w; }
        else r = f3 == 1 ? (a << ix) | (a >> ((64 - ix) & 63)) : (a >> ix) | (a << ((64 - ix) & 63));
        break;
      case 0x34: if (!zbs || f3 != 1) return 0; r = a ^ (1ULL << ix); break;
      default: return 0;
    }
  } else if (op == 0x13) {
    const unsigned top5 = imm12 >> 7, amt = imm12 & 0x7f;
    if (f3 == 1) {
      if (top5 == 0x0c) {
        if (!zbb) return 0;
        switch (amt) {
          case 0: r = rv32 ? (aw ? (u64)__builtin_clz(aw) : 32) : (a ? (u64)__builtin_clzll(a) : 64); break;
          case 1: r = rv32 ? (aw ? (u64)__builtin_ctz(aw) : 32) : (a ? (u64)__builtin_ctzll(a) : 64); break;
          case 2: r = (u64)__builtin_popcountll(ua); break;
          case 4: r = (u64)(i64)(int8_t)a; break;
          case 5: r = (u64)(i64)(int16_t)a; break;
          default: return 0;
        }
      } else if (top5 == 5 || top5 == 9 || top5 == 0x0d) {
        if (!zbs || amt > smask) return 0;                                  /* checkShiftImmediate (Hart.cpp:11736-11746) */
        r = top5 == 5 ? a | (1ULL << amt) : top5 == 9 ? a & ~(1ULL << amt) : a ^ (1ULL << amt);
      } else return 0;
    } else if (f3 == 5) {
      if (top5 == 5) { unsigned i; if (!zbb || amt != 7) return 0; r = 0; for (i = 0; i < 64; i += 8) if ((a >> i) & 0xff) r |= 0xffULL << i; }
      else if (top5 == 9) { if (!zbs || amt > smask) return 0; r = (ua >> amt) & 1; }
      else if (top5 == 0x0c) {
        if (!zbb || amt > smask) return 0;
        if (rv32) r = (u32)((aw >> amt) | (aw << ((32 - amt) & 31))); else r = (a >> amt) | (a << ((64 - amt) & 63));
      }
      else if (!rv32 && imm12 == 0x6b8) { if (!zbb) return 0; r = __builtin_bswap64(a); }
      else if (rv32 && imm12 == 0x698) { if (!zbb) return 0; r = __builtin_bswap32(aw); }
      else return 0;
    } else return 0;
  } else if (op == 0x1b) {                                                  /* RV64 only: the caller has rejected OP-IMM-32 on RV32 */
    const unsigned top5 = imm12 >> 7, amt = imm12 & 0x7f;
    if (f3 == 1 && (imm12 >> 6) == 2) { if (!zba) return 0; r = (u64)aw << (amt & 63); }
    else if (f3 == 1 && top5 == 0x0c && amt <= 2) {
      if (!zbb) return 0;
      r = amt == 0 ? (aw ? (u64)__builtin_clz(aw) : 32) : amt == 1 ? (aw ? (u64)__builtin_ctz(aw) : 32) : (u64)__builtin_popcount(aw);
    }
    else if (f3 == 5 && f7 == 0x30) { const unsigned s = (inst >> 20) & 31; if (!zbb) return 0; r = (u64)(i64)(i32)((aw >> s) | (aw << ((32 - s) & 31))); }
    else return 0;
  } else if (op == 0x3b) {
    const unsigned s = bw & 31;
    if (f7 == 0x04 && f3 == 0) { if (!zba) return 0; r = (u64)aw + b; }
    else if (f7 == 0x04 && f3 == 4) { if (!zbb || ((inst >> 20) & 31) != 0) return 0; r = a & 0xffff; }                /* zext.h = packw rd, rs1, x0 (bitmanip.cpp:498-524) */
    else if (f7 == 0x10 && (f3 == 2 || f3 == 4 || f3 == 6)) { if (!zba) return 0; r = ((u64)aw << (f3 >> 1)) + b; }
    else if (f7 == 0x30 && (f3 == 1 || f3 == 5)) {
      if (!zbb) return 0;
      r = (u64)(i64)(i32)(f3 == 1 ? (aw << s) | (aw >> ((32 - s) & 31)) : (aw >> s) | (aw << ((32 - s) & 31)));
    }
    else return 0;
  } else return 0;
  *out = r;
  return 1;
}

/* `inst` is the (possibly expanded) 32-bit encoding; `raw` the bits as fetched (mtval on illegal). */
static void execute(rvo_hart* h, u32 inst, u32 raw, unsigned size)
{
  u32 op = inst & 0x7f, rd = (inst >> 7) & 31, f3 = (inst >> 12) & 7, rs1 = (inst >> 15) & 31, rs2 = (inst >> 20) & 31;
  u32 f7 = inst >> 25;
  i64 immi = (i32)inst >> 20;
  u64 a = h->x[rs1], b = h->x[rs2];
  int zca = ext(h, RVO_ISA_C);
  const int rv32 = h->rv32;
  const u64 pcm = rv32 ? 0xffffffffULL : ~0ULL;          /* RV32: pc and addresses wrap at 32 bits */
  if (rv32) {                                            /* encodings that exist on RV64 only */
    if (op == 0x1b || op == 0x3b || (op == 0x03 && (f3 == 3 || f3 == 6)) || (op == 0x23 && f3 == 3)) goto ill;
  }
  if ((h->cfg.isa & (RVO_ISA_ZBA | RVO_ISA_ZBB | RVO_ISA_ZBS | RVO_ISA_ZBC | RVO_ISA_ZICOND)) && (op == 0x13 || op == 0x1b || op == 0x33 || op == 0x3b)) {
    u64 r;
    if (exec_bitmanip(h, inst, a, b, &r)) { wx(h, rd, r); return; }
  }
  if (rv32 && op == 0x13 && (f3 == 1 || f3 == 5) && (inst & (1u << 25))) goto ill;   /* shamt[5] (Hart.cpp:11736-11746) */
  switch (op) {
    case 0x37: wx(h, rd, (u64)(i64)(i32)(inst & 0xfffff000u)); return;
    case 0x17: wx(h, rd, h->curr_pc + (u64)(i64)(i32)(inst & 0xfffff000u)); return;
    case 0x6f: {
      i64 off = ((i64)(i32)(inst & 0x80000000u) >> 11) | (inst & 0xff000) | ((inst >> 9) & 0x800) | ((inst >> 20) & 0x7fe);
      u64 t = (h->curr_pc + (u64)off) & pcm;
      if (!zca && (t & 3)) { take_exception(h, 0, h->curr_pc, t); return; }          /* Hart.cpp:11687-11690 */
      wx(h, rd, h->curr_pc + size); h->pc = t & ~1ULL; return; }
    case 0x67: {
      if (f3 != 0) break;
      u64 t = (a + (u64)immi) & ~1ULL & pcm;
      if (!zca && (t & 3)) { take_exception(h, 0, h->curr_pc, t); return; }          /* Hart.cpp:11711-11714 */
      wx(h, rd, h->curr_pc + size); h->pc = t; return; }
    case 0x63: {
      int take;
      switch (f3) {
        case 0: take = a == b; break; case 1: take = a != b; break;
        case 4: take = (i64)a < (i64)b; break; case 5: take = (i64)a >= (i64)b; break;
        case 6: take = a < b; break; case 7: take = a >= b; break;
        default: illegal(h, raw); return;
      }
      if (take) {
        i64 off = ((i64)(i32)(inst & 0x80000000u) >> 19) | ((inst & 0x80) << 4) | ((inst >> 20) & 0x7e0) | ((inst >> 7) & 0x1e);
        u64 t = (h->curr_pc + (u64)off) & pcm;
        if (!zca && (t & 3)) { take_exception(h, 0, h->curr_pc, t); return; }        /* Hart.cpp:1652-1656 */
        h->pc = t & ~1ULL;
      }
      return; }
    case 0x03: {
      static const unsigned sz[8] = { 1, 2, 4, 8, 1, 2, 4, 0 };
      if (f3 == 7) break;
      u64 v;
      if (!do_load(h, a + (u64)immi, sz[f3], &v)) return;
      switch (f3) { case 0: v = (u64)(i64)(int8_t)v; break; case 1: v = (u64)(i64)(int16_t)v; break; case 2: v = (u64)(i64)(i32)v; break; default: break; }
      wx(h, rd, v); return; }
    case 0x23: {
      if (f3 > 3) break;
      i64 imms = ((i64)(i32)(inst & 0xfe000000u) >> 20) | ((inst >> 7) & 31);
      do_store(h, a + (u64)imms, 1u << f3, b); return; }
    case 0x13: {
      u64 r; unsigned sh = (inst >> 20) & 63;
      switch (f3) {
        case 0: r = a + (u64)immi; break;
        case 1: if ((inst >> 26) != 0) goto ill; r = a << sh; break;                  /* Hart.cpp:11736-11746 (RV32: wx keeps the low word) */
        case 2: r = (i64)a < immi; break;
        case 3: r = a < (u64)immi; break;
        case 4: r = a ^ (u64)immi; break;
        case 5: if ((inst >> 26) == 0) r = (rv32 ? (u64)(u32)a : a) >> sh; else if ((inst >> 26) == 0x10) r = (u64)((i64)a >> sh); else goto ill; break;
        case 6: r = a | (u64)immi; break;
        default: r = a & (u64)immi; break;
      }
      wx(h, rd, r); return; }
    case 0x1b: {
      u32 r; unsigned sh = rs2;
      switch (f3) {
        case 0: r = (u32)a + (u32)immi; break;
        case 1: if (f7 != 0) goto ill; r = (u32)a << sh; break;
        case 5: if (f7 == 0) r = (u32)a >> sh; else if (f7 == 0x20) r = (u32)((i32)a >> sh); else goto ill; break;
        default: goto ill;
      }
      wx(h, rd, (u64)(i64)(i32)r); return; }
    case 0x33: {
      u64 r;
      if (f7 == 1) {
        if (!ext(h, RVO_ISA_M)) goto ill;                                            /* Hart.cpp:13966,14103 */
        if (rv32) {                                                                  /* Hart<uint32_t>: 32-bit operands (Hart.cpp:13964-14170) */
          const u32 aw = (u32)a, bw = (u32)b;
          switch (f3) {
            case 0: r = (u64)(aw * bw); break;
            case 1: r = (u64)(((i64)(i32)aw * (i64)(i32)bw) >> 32); break;
            case 2: r = (u64)(((i64)(i32)aw * (i64)(u64)bw) >> 32); break;
            case 3: r = ((u64)aw * (u64)bw) >> 32; break;
            case 4: r = bw == 0 ? ~0u : ((i32)aw == INT32_MIN && (i32)bw == -1) ? aw : (u32)((i32)aw / (i32)bw); break;
            case 5: r = bw == 0 ? ~0u : aw / bw; break;
            case 6: r = bw == 0 ? aw : ((i32)aw == INT32_MIN && (i32)bw == -1) ? 0 : (u32)((i32)aw % (i32)bw); break;
            default: r = bw == 0 ? aw : aw % bw; break;
          }
        } else
        switch (f3) {
          case 0: r = a * b; break;
          case 1: r = (u64)(((i128)(i64)a * (i128)(i64)b) >> 64); break;
          case 2: r = (u64)(((i128)(i64)a * (i128)(u128)b) >> 64); break;
          case 3: r = (u64)(((u128)a * (u128)b) >> 64); break;
          case 4: r = b == 0 ? ~0ULL : ((i64)a == INT64_MIN && (i64)b == -1) ? a : (u64)((i64)a / (i64)b); break;
          case 5: r = b == 0 ? ~0ULL : a / b; break;
          case 6: r = b == 0 ? a : ((i64)a == INT64_MIN && (i64)b == -1) ? 0 : (u64)((i64)a % (i64)b); break;
          default: r = b == 0 ? a : a % b; break;
        }
      } else if (f7 == 0) {
        switch (f3) {
          case 0: r = a + b; break; case 1: r = a << (b & (rv32 ? 31 : 63)); break; case 2: r = (i64)a < (i64)b; break;
          case 3: r = a < b; break; case 4: r = a ^ b; break; case 5: r = rv32 ? (u64)((u32)a >> (b & 31)) : a >> (b & 63); break;
          case 6: r = a | b; break; default: r = a & b; break;
        }
      } else if (f7 == 0x20) {
        if (f3 == 0) r = a - b; else if (f3 == 5) r = (u64)((i64)a >> (b & (rv32 ? 31 : 63))); else goto ill;
      } else goto ill;
      wx(h, rd, r); return; }
    case 0x3b: {
      u32 r, aw = (u32)a, bw = (u32)b;
      if (f7 == 1) {
        if (!ext(h, RVO_ISA_M)) goto ill;
        switch (f3) {
          case 0: r = aw * bw; break;
          case 4: r = bw == 0 ? ~0u : ((i32)aw == INT32_MIN && (i32)bw == -1) ? aw : (u32)((i32)aw / (i32)bw); break;
          case 5: r = bw == 0 ? ~0u : aw / bw; break;
          case 6: r = bw == 0 ? aw : ((i32)aw == INT32_MIN && (i32)bw == -1) ? 0 : (u32)((i32)aw % (i32)bw); break;
          case 7: r = bw == 0 ? aw : aw % bw; break;
          default: goto ill;
        }
      } else if (f7 == 0) {
        if (f3 == 0) r = aw + bw; else if (f3 == 1) r = aw << (bw & 31); else if (f3 == 5) r = aw >> (bw & 31); else goto ill;
      } else if (f7 == 0x20) {
        if (f3 == 0) r = aw - bw; else if (f3 == 5) r = (u32)((i32)aw >> (bw & 31)); else goto ill;
      } else goto ill;
      wx(h, rd, (u64)(i64)(i32)r); return; }
    case 0x0f:
      if (f3 == 0 || f3 == 1) return;                       /* fence / fence.i (Hart.cpp:11937-11980) */
      break;
    case 0x73: {
      if (f3 == 0) {
        if (rd != 0) break;
        if (inst == 0x00000073) { take_exception(h, h->priv == PRV_M ? 11 : h->priv == PRV_S ? 9 : 8, h->curr_pc, 0); return; }
        if (inst == 0x00100073) { take_exception(h, 3, h->curr_pc, ext(h, RVO_CFG_CLR_MTVAL_EBREAK) ? 0 : h->curr_pc); return; }      /* Hart.cpp:12094-12101 */
        if (inst == 0x30200073) {                                                              /* mret: Hart.cpp:12247-12336 */
          if (h->priv < PRV_M) goto ill;
          u64 v = h->mstatus; unsigned saved = (v >> MST_MPP_SH) & 3;
          v = (v & ~MST_MIE) | ((v & MST_MPIE) ? MST_MIE : 0);
          unsigned least = ext(h, RVO_ISA_U) ? PRV_U : ext(h, RVO_ISA_S) ? PRV_S : PRV_M;
          v = (v & ~(3ULL << MST_MPP_SH)) | ((u64)least << MST_MPP_SH);
          v |= MST_MPIE;
          if (saved != PRV_M && !ext(h, RVO_CFG_KEEP_MPRV)) v &= ~MST_MPRV;   /* clearMprvOnRet_ (Hart.cpp:12283) */
          csr_store(h, C_MSTATUS, v, 1);
          h->pc = h->mepc & epc_rmask(h); h->priv = saved; return;      /* readSignExtend(MEPC) applies the read mask */
        }
        if (inst == 0x10200073) {                                                              /* sret: Hart.cpp:12428-12560 */
          if (!ext(h, RVO_ISA_S)) goto ill;
          if (h->priv < PRV_S) goto ill;
          if ((h->mstatus & MST_TSR) && h->priv == PRV_S) goto ill;
          u64 v = h->mstatus; unsigned saved = (v & MST_SPP) ? PRV_S : PRV_U;
          v = (v & ~MST_SIE) | ((v & MST_SPIE) ? MST_SIE : 0);
          v = ext(h, RVO_ISA_U) ? (v & ~MST_SPP) : (v | MST_SPP);
          v |= MST_SPIE;
          if (!ext(h, RVO_CFG_KEEP_MPRV)) v &= ~MST_MPRV;    /* savedMode is never M for sret (Hart.cpp:12499) */
          h->mstatus = legalize_mstatus(h, v); record_csr(h, C_SSTATUS);
          h->pc = h->sepc & epc_rmask(h); h->priv = saved; return;
        }
        if (inst == 0x10500073) {                                                              /* wfi: Hart.cpp:12618 */
          if (h->priv == PRV_U && ext(h, RVO_ISA_S)) goto ill;
          if (h->priv != PRV_M && (h->mstatus & MST_TW)) goto ill;
          return;
        }
        if (f7 == 0x09) {                                                                      /* sfence.vma: Hart.cpp:12108 */
          if (!ext(h, RVO_ISA_S)) goto ill;
          if (h->priv == PRV_U) goto ill;
          if (h->priv == PRV_S && (h->mstatus & MST_TVM)) goto ill;
          rvo_sfence(h, rs1 != 0, h->x[rs1], rs2 != 0, h->x[rs2]);                           /* Hart.cpp:12139-12173: by register number */
          return;
        }
        break;
      }
      if (f3 == 4) break;
      { unsigned csrop = f3 & 3; u64 src = (f3 & 4) ? (u64)rs1 : (rv32 ? (u64)(u32)a : a);
        exec_csr(h, raw, csrop, rd, inst >> 20, src, rs1 != 0); }
      return; }
    case 0x2f: exec_amo(h, inst, raw); return;
    case 0x07: case 0x27:
      { unsigned w = (inst >> 12) & 7;                                      /* vector widths share the FP load/store opcodes */
        if (w == 0 || w >= 5) { h->ch_vreg = (int)((inst >> 7) & 31); exec_vector_ldst(h, inst, raw, op == 0x27); if (op == 0x27) h->ch_vgroup = 0; return; } }
      exec_fp(h, inst, raw); return;
    case 0x43: case 0x47: case 0x4b: case 0x4f: case 0x53: exec_fp(h, inst, raw); return;
    case 0x57: h->ch_vreg = (int)((inst >> 7) & 31); exec_vector(h, inst, raw); return;
    default: break;
  }
ill:
  illegal(h, raw);
}

static void exec_amo(rvo_hart* h, u32 inst, u32 raw)
{
  u32 rd = (inst >> 7) & 31, f3 = (inst >> 12) & 7, rs1 = (inst >> 15) & 31, rs2 = (inst >> 20) & 31, f5 = inst >> 27;
  if (!ext(h, RVO_ISA_A) || (f3 != 2 && f3 != 3) || (h->rv32 && f3 == 3)) { illegal(h, raw); return; }   /* .d forms: RV64 only (amo.cpp:886, 918, 964) */
  unsigned size = f3 == 2 ? 4 : 8;
  u64 va = h->rv32 ? (u64)(u32)h->x[rs1] : h->x[rs1], tval;
  if (f5 == 2) {                                            /* lr: amo.cpp:142-266 */
    if (rs2 != 0) { illegal(h, raw); return; }
    h->ld_has = 1; h->ld_va = va; h->ld_pa = va;
    if (ldst_translate_unsupported(h)) return;
    u64 pa1, pa2;
    /* determineLoadException first (a translation fault wins), then the alignment / reservability check; either way
     * the trap value is the LR's own address (amo.cpp:203-207) */
    unsigned c = ldst_addr(h, va, size, 1, &pa1, &pa2, &tval);
    if (c == 4) c = 5;                                      /* misalAtomicCauseAccessFault_ (amo.cpp:180-181) */
    if (c == 0 && (va & (size - 1))) c = 5;
    if (c) { ldst_exception(h, c, va); return; }
    h->ld_pa = pa1;
    u64 v = ld_le(h->mem + (pa1 - h->cfg.mem_base), size);
    if (size == 4) v = (u64)(i64)(i32)v;
    wx(h, rd, v);
    h->lr_valid = 1; h->lr_size = 8; h->lr_addr = pa1 & ~7ULL;      /* lrResSize_ = sizeof(URV) (Hart.hpp:6733); physical address (amo.cpp:254) */
    return;
  }
  if (f5 == 3) {                                            /* sc: amo.cpp:273-420 */
    h->ld_has = 1; h->ld_va = va; h->ld_pa = va;
    if (ldst_translate_unsupported(h)) return;
    int exc = 0;
    u64 pa1 = va, pa2;
    if (va & (size - 1)) { ldst_exception(h, 7, va); exc = 1; }          /* misalHasPriority_: before any translation (amo.cpp:304-309) */
    else { unsigned c = ldst_addr(h, va, size, 2, &pa1, &pa2, &tval); if (c) { ldst_exception(h, c, va); exc = 1; } }
    int had = h->lr_valid && h->lr_addr <= pa1 && pa1 + size <= h->lr_addr + h->lr_size;
    h->lr_valid = 0;                                                  /* cancelLr(SC) amo.cpp:403-405 */
    if (exc) return;
    h->ld_pa = pa1;
    if (!had) { wx(h, rd, 1); return; }
    u64 sv = size == 8 ? h->x[rs2] : (h->x[rs2] & 0xffffffffULL);
    h->ch_mem = 1; h->ch_mem_addr = va; h->ch_mem_val = sv;
    st_le(h->mem + (pa1 - h->cfg.mem_base), sv, size);
    wx(h, rd, 0);
    return;
  }
  switch (f5) { case 0: case 1: case 4: case 8: case 12: case 16: case 20: case 24: case 28: break; default: illegal(h, raw); return; }
  /* amo: execAmo32Op / execAmo64Op (amo.cpp:427-460): amoLoad then store<>() */
  h->ld_has = 1; h->ld_va = va; h->ld_pa = va;
  if (ldst_translate_unsupported(h)) return;
  u64 apa1 = va, apa2;
  if (va & (size - 1)) { ldst_exception(h, 7, va); return; }           /* validateAmoAddr: misaligned first (amo.cpp:35-38) */
  { unsigned c = ldst_addr(h, va, size, 2, &apa1, &apa2, &tval); if (c) { ldst_exception(h, c, va); return; } }   /* amoLoad validates with the store check only (amo.cpp:28-56) */
  u64 ld = ld_le(h->mem + (apa1 - h->cfg.mem_base), size);
  if (size == 4) ld = (u64)(i64)(i32)ld;
  u64 b = h->x[rs2], r;
  switch (f5) {
    case 0: r = b + ld; break; case 1: r = b; break; case 4: r = b ^ ld; break; case 8: r = b | ld; break; case 12: r = b & ld; break;
    case 16: r = size == 4 ? (u64)(i64)((i32)b < (i32)ld ? (i32)b : (i32)ld) : (u64)((i64)b < (i64)ld ? (i64)b : (i64)ld); break;
    case 20: r = size == 4 ? (u64)(i64)((i32)b > (i32)ld ? (i32)b : (i32)ld) : (u64)((i64)b > (i64)ld ? (i64)b : (i64)ld); break;
    case 24: r = size == 4 ? (u64)((u32)b < (u32)ld ? (u32)b : (u32)ld) : (b < ld ? b : ld); break;
    default: r = size == 4 ? (u64)((u32)b > (u32)ld ? (u32)b : (u32)ld) : (b > ld ? b : ld); break;
  }
  if (!do_store(h, va, size, r)) return;
  if (h->status == RVO_STOP_TOHOST) { wx(h, rd, ld); return; }
  wx(h, rd, ld);
}

static void exec_fp(rvo_hart* h, u32 inst, u32 raw)
{
  u32 op = inst & 0x7f, rd = (inst >> 7) & 31, f3 = (inst >> 12) & 7, rs1 = (inst >> 15) & 31, rs2 = (inst >> 20) & 31;
  u32 f7 = inst >> 25, rs3 = inst >> 27, fmt = (inst >> 25) & 3;
  i64 immi = (i32)inst >> 20;
  u64 xa = h->x[rs1];
  if (!SF.lib) { stop(h, RVO_STOP_UNSUPPORTED, 0xF10A7); return; }
  if (op == 0x07) {                                           /* flw / fld: float.cpp:281-298, 1019-1036 */
    u64 v;
    if (f3 == 2) { if (!fp_legal(h)) goto ill; if (!do_load(h, xa + (u64)immi, 4, &v)) return; wr_s(h, rd, (u32)v); mark_fs_dirty(h); return; }
    if (f3 == 3) { if (!dp_legal(h)) goto ill; if (!do_load(h, xa + (u64)immi, 8, &v)) return; wf(h, rd, v); mark_fs_dirty(h); return; }
    goto ill;
  }
  if (op == 0x27) {                                           /* fsw / fsd: float.cpp:303-321, 1041-1057 */
    i64 imms = ((i64)(i32)(inst & 0xfe000000u) >> 20) | ((inst >> 7) & 31);
    if (f3 == 2) { if (!fp_legal(h)) goto ill; do_store(h, xa + (u64)imms, 4, h->f[rs2] & 0xffffffffULL); return; }
    if (f3 == 3) { if (!dp_legal(h)) goto ill; do_store(h, xa + (u64)imms, 8, h->f[rs2]); return; }
    goto ill;
  }
  if (op != 0x53) {                                           /* fused multiply-add family */
    if (fmt == 0) {
      if (!fp_legal(h) || !fp_begin(h, f3)) goto ill;
      u32 a = rd_s(h, rs1), b = rd_s(h, rs2), c = rd_s(h, rs3);
      if (op == 0x47 || op == 0x4f) c = neg32(c);
      if (op == 0x4b || op == 0x4f) a = neg32(a);
      sf32 A = { a }, Bv = { b }, C = { c };
      wr_s(h, rd, SF.f32_mulAdd(A, Bv, C).v); fp_accrue(h); mark_fs_dirty(h); return;
    }
    if (fmt == 1) {
      if (!dp_legal(h) || !fp_begin(h, f3)) goto ill;
      u64 a = h->f[rs1], b = h->f[rs2], c = h->f[rs3];
      if (op == 0x47 || op == 0x4f) c = neg64(c);
      if (op == 0x4b || op == 0x4f) a = neg64(a);
      sf64 A = { a }, Bv = { b }, C = { c };
      wf(h, rd, SF.f64_mulAdd(A, Bv, C).v); fp_accrue(h); mark_fs_dirty(h); return;
    }
    goto ill;
  }
  int dbl = f7 & 1;
  if (f7 & 2) goto ill;                                       /* Zfh / Q not enabled */
  u32 sa = rd_s(h, rs1), sb = rd_s(h, rs2);
  u64 da = h->f[rs1], db = h->f[rs2];
  sf32 SA = { sa }, SB = { sb }; sf64 DA = { da }, DB = { db };
  switch (f7 >> 2) {
    case 0x00: case 0x01: case 0x02: case 0x03: {             /* fadd fsub fmul fdiv */
      unsigned k = f7 >> 2;
      if (!dbl) {
        if (!fp_legal(h) || !fp_begin(h, f3)) goto ill;
        sf32 r = k == 0 ? SF.f32_add(SA, SB) : k == 1 ? SF.f32_sub(SA, SB) : k == 2 ? SF.f32_mul(SA, SB) : SF.f32_div(SA, SB);
        wr_s(h, rd, r.v);
      } else {
        if (!dp_legal(h) || !fp_begin(h, f3)) goto ill;
        sf64 r = k == 0 ? SF.f64_add(DA, DB) : k == 1 ? SF.f64_sub(DA, DB) : k == 2 ? SF.f64_mul(DA, DB) : SF.f64_div(DA, DB);
        wf(h, rd, r.v);
      }
      fp_accrue(h); mark_fs_dirty(h); return; }
    case 0x0b:                                                /* fsqrt */
      if (rs2 != 0) goto ill;
      if (!dbl) { if (!fp_legal(h) || !fp_begin(h, f3)) goto ill; wr_s(h, rd, SF.f32_sqrt(SA).v); }
      else { if (!dp_legal(h) || !fp_begin(h, f3)) goto ill; wf(h, rd, SF.f64_sqrt(DA).v); }
      fp_accrue(h); mark_fs_dirty(h); return;
    case 0x04:                                                /* fsgnj* : float.cpp:499-564 */
      if (f3 > 2) goto ill;
      if (!dbl) {
        if (!fp_legal(h)) goto ill;
        u32 s = f3 == 0 ? (sb & 0x80000000u) : f3 == 1 ? (~sb & 0x80000000u) : ((sa ^ sb) & 0x80000000u);
        wr_s(h, rd, (sa & 0x7fffffffu) | s);
      } else {
        if (!dp_legal(h)) goto ill;
        u64 m = 0x8000000000000000ULL;
        u64 s = f3 == 0 ? (db & m) : f3 == 1 ? (~db & m) : ((da ^ db) & m);
        wf(h, rd, (da & ~m) | s);
      }
      mark_fs_dirty(h); return;
    case 0x05:                                                /* fmin / fmax : float.cpp:569-634 */
      if (f3 > 1) goto ill;
      if (!dbl) {
        if (!fp_legal(h)) goto ill;
        u32 r; int n1 = is_nan32(sa), n2 = is_nan32(sb);
        if (n1 && n2) r = F32_CANON; else if (n1) r = sb; else if (n2) r = sa;
        else if (f3 == 0) r = (lt32(sa, sb) || (eq32(sa, sb) && (sa >> 31))) ? sa : sb;
        else r = (lt32(sb, sa) || (eq32(sa, sb) && !(sa >> 31))) ? sa : sb;
        if (is_snan32(sa) || is_snan32(sb)) or_fflags(h, 16);
        wr_s(h, rd, r);
      } else {
        if (!dp_legal(h)) goto ill;
        u64 r; int n1 = is_nan64(da), n2 = is_nan64(db);
        if (n1 && n2) r = F64_CANON; else if (n1) r = db; else if (n2) r = da;
        else if (f3 == 0) r = (lt64(da, db) || (eq64(da, db) && (da >> 63))) ? da : db;
        else r = (lt64(db, da) || (eq64(da, db) && !(da >> 63))) ? da : db;
        if (is_snan64(da) || is_snan64(db)) or_fflags(h, 16);
        wf(h, rd, r);
      }
      mark_fs_dirty(h); return;
    case 0x08:                                                /* fcvt.s.d / fcvt.d.s */
      if (!dbl) { if (rs2 != 1) goto ill; if (!dp_legal(h) || !fp_begin(h, f3)) goto ill; wr_s(h, rd, SF.f64_to_f32(DA).v); }
      else { if (rs2 != 0) goto ill; if (!dp_legal(h) || !fp_begin(h, f3)) goto ill; wf(h, rd, SF.f32_to_f64(SA).v); }
      fp_accrue(h); mark_fs_dirty(h); return;
    case 0x14: {                                              /* fle flt feq : float.cpp:694-764 */
      if (f3 > 2) goto ill;
      u64 r = 0;
      if (!dbl) {
        if (!fp_legal(h)) goto ill;
        if (is_nan32(sa) || is_nan32(sb)) { if (f3 != 2 || is_snan32(sa) || is_snan32(sb)) or_fflags(h, 16); }
        else r = f3 == 2 ? eq32(sa, sb) : f3 == 1 ? lt32(sa, sb) : (lt32(sa, sb) || eq32(sa, sb));
      } else {
        if (!dp_legal(h)) goto ill;
        if (is_nan64(da) || is_nan64(db)) { if (f3 != 2 || is_snan64(da) || is_snan64(db)) or_fflags(h, 16); }
        else r = f3 == 2 ? eq64(da, db) : f3 == 1 ? lt64(da, db) : (lt64(da, db) || eq64(da, db));
      }
      wx(h, rd, r); return; }
    case 0x18: {                                              /* fcvt.{w,wu,l,lu}.{s,d} : float.cpp:639-669, 898-948 */
      if (rs2 > 3 || (h->rv32 && rs2 > 1)) goto ill;              /* the l / lu forms: RV64 only (float.cpp:900, 931, 1599, 1631) */
      if (!(dbl ? dp_legal(h) : fp_legal(h)) || !fp_begin(h, f3)) goto ill;
      uint_fast8_t rm = *SF.rm; u64 r;
      if (!dbl) {
        r = rs2 == 0 ? (u64)(i64)(i32)SF.f32_to_i32(SA, rm, 1) : rs2 == 1 ? (u64)(i64)(i32)(u32)SF.f32_to_ui32(SA, rm, 1)
          : rs2 == 2 ? (u64)SF.f32_to_i64(SA, rm, 1) : (u64)SF.f32_to_ui64(SA, rm, 1);
      } else {
        r = rs2 == 0 ? (u64)(i64)(i32)SF.f64_to_i32(DA, rm, 1) : rs2 == 1 ? (u64)(i64)(i32)(u32)SF.f64_to_ui32(DA, rm, 1)
          : rs2 == 2 ? (u64)SF.f64_to_i64(DA, rm, 1) : (u64)SF.f64_to_ui64(DA, rm, 1);
      }
      wx(h, rd, r); fp_accrue(h); return; }
    case 0x1a: {                                              /* fcvt.{s,d}.{w,wu,l,lu} */
      if (rs2 > 3 || (h->rv32 && rs2 > 1)) goto ill;              /* float.cpp:963, 996, 1655, 1680 */
      if (!(dbl ? dp_legal(h) : fp_legal(h)) || !fp_begin(h, f3)) goto ill;
      if (!dbl) {
        sf32 r = rs2 == 0 ? SF.i32_to_f32((i32)xa) : rs2 == 1 ? SF.ui32_to_f32((u32)xa) : rs2 == 2 ? SF.i64_to_f32((i64)xa) : SF.ui64_to_f32(xa);
        wr_s(h, rd, r.v);
      } else {
        sf64 r = rs2 == 0 ? SF.i32_to_f64((i32)xa) : rs2 == 1 ? SF.ui32_to_f64((u32)xa) : rs2 == 2 ? SF.i64_to_f64((i64)xa) : SF.ui64_to_f64(xa);
        wf(h, rd, r.v);
      }
      fp_accrue(h); mark_fs_dirty(h); return; }
    case 0x1c:                                                /* fmv.x.w / fmv.x.d / fclass */
      if (rs2 != 0) goto ill;
      if (f3 == 0) {
        if (!dbl) { if (!fp_legal(h)) goto ill; wx(h, rd, (u64)(i64)(i32)(u32)h->f[rs1]); }     /* raw bits: float.cpp:682-688 */
        else { if (!dp_legal(h) || h->rv32) goto ill; wx(h, rd, h->f[rs1]); }                   /* fmv.x.d: RV64 only (float.cpp:1705) */
        return;
      }
      if (f3 == 1) {
        if (!dbl) { if (!fp_legal(h)) goto ill; wx(h, rd, fclass32(sa)); }
        else { if (!dp_legal(h)) goto ill; wx(h, rd, fclass64(da)); }
        return;
      }
      goto ill;
    case 0x1e:                                                /* fmv.w.x / fmv.d.x */
      if (rs2 != 0 || f3 != 0) goto ill;
      if (!dbl) { if (!fp_legal(h)) goto ill; wr_s(h, rd, (u32)xa); }
      else { if (!dp_legal(h) || h->rv32) goto ill; wf(h, rd, xa); }                            /* fmv.d.x: RV64 only (float.cpp:1732) */
      mark_fs_dirty(h); return;
    default: break;
  }
ill:
  illegal(h, raw);
}

/* ------------------------------------------------------------------ fetch + step */
/* fetchInstNoTrap (Hart.cpp:3052-3206). Returns 0 on success, 1 with *cause / *tval on a fault, 2 when the hart stopped. */
static int fetch(rvo_hart* h, u64 pc, u32* inst, u64* tval, unsigned* cause)
{
  if (ext(h, RVO_ISA_S) && h->priv != PRV_M && (h->satp >> 60) > 8) { stop(h, RVO_STOP_UNSUPPORTED, 0x180); return 2; }
  const int tr = paging_on(h, h->priv);
  u64 pa = pc;
  unsigned c;
  *cause = 1;
  if (tr && (c = translate(h, pc, h->priv, 4, &pa)) != 0) { *cause = c; *tval = pc; return 1; }
  if (!pmp_ok(h, h->priv, pa, 4)) { *tval = pc; return 1; }                       /* Hart.cpp:3079-3084 */
  if (!(pa & 3)) {
    if (!in_mem(h, pa, 4)) { *tval = pc; return 1; }
    u32 w = (u32)ld_le(h->mem + (pa - h->cfg.mem_base), 4);
    *inst = (w & 3) != 3 ? (w & 0xffff) : w;
    return 0;
  }
  if (!in_mem(h, pa, 2)) { *tval = pc; return 1; }
  u32 lo = (u32)ld_le(h->mem + (pa - h->cfg.mem_base), 2);
  if ((lo & 3) != 3) { *inst = lo; return 0; }
  /* second half: pa2 starts out as the *virtual* address + 2 and is translated when its page index differs from the
   * first half's physical page index (Hart.cpp:3144-3156) */
  u64 pa2 = pc + 2;
  if ((pa >> 12) != (pa2 >> 12) && tr && (c = translate(h, pc + 2, h->priv, 4, &pa2)) != 0) { *cause = c; *tval = pc + 2; return 1; }
  if (!pmp_ok(h, h->priv, pa2, 4)) { *tval = pc + 2; return 1; }                  /* Hart.cpp:3158-3162 */
  if (!in_mem(h, pa2, 2)) { *tval = pc + 2; return 1; }
  *inst = lo | ((u32)ld_le(h->mem + (pa2 - h->cfg.mem_base), 2) << 16);
  return 0;
}

/* Hart::readInst (Hart.cpp:2997-3037): the interrupted instruction, for the log only */
static u32 read_inst_for_log(rvo_hart* h, u64 va)
{
  const int tr = paging_on(h, h->priv);
  u64 pa = va;
  if (tr && translate_noupdate(h, va, h->priv, 4, &pa)) return 0;
  if (!in_mem(h, pa, 2)) return 0;
  u32 iw = (u32)ld_le(h->mem + (pa - h->cfg.mem_base), 2);
  if ((iw & 3) != 3) return iw;
  u64 va2 = va + 2, pa2 = pa + 2;
  if (tr && (va >> 12) != (va2 >> 12) && translate_noupdate(h, va2, h->priv, 4, &pa2)) return 0;
  if (in_mem(h, pa2, 2)) iw |= (u32)ld_le(h->mem + (pa2 - h->cfg.mem_base), 2) << 16;
  return iw;
}

/* Hart::run / untilAddress / singleStep called again after a limit or the stop address ended the last call (Hart.cpp:6003, 6963) */
void rvo_resume(rvo_hart* h) { if (h->status == RVO_STOP_LIMIT || h->status == RVO_STOP_ADDR) h->status = RVO_RUNNING; }
void rvo_set_stop(rvo_hart* h, u64 max_ret, int stop_pc_valid, u64 stop_pc) { h->max_ret = max_ret; h->stop_pc_valid = stop_pc_valid; h->stop_pc = stop_pc; }

int rvo_run(rvo_hart* h, u64 max_exec, rvo_commit* log, size_t cap, size_t* n)
{
  if (n) *n = 0;
  while (h->status == RVO_RUNNING) {
    if (h->exec_count >= max_exec || h->retire_count >= h->max_ret) { h->status = RVO_STOP_LIMIT; break; }   /* Hart.cpp:6025-6026, 6224 */
    if (h->stop_pc_valid && h->pc == h->stop_pc) { h->status = RVO_STOP_ADDR; break; }                         /* Hart.cpp:6025, 6230 */
    begin_inst(h);
    h->time_++;                                              /* tickTime, timeDownSample_=0, 1 hart (Hart.hpp:3101) */
    h->curr_pc = h->pc;
    h->exec_count++;
    if (!(h->mcountinhibit & 1)) h->cycle++;                 /* Hart.cpp:6072-6074 */
    unsigned priv0 = h->priv;
    u32 inst = 0; u64 tval = 0;
    {                                                        /* processExternalInterrupt (Hart.cpp:7216-7262) */
      unsigned icause, inext;
      if (interrupt_possible(h, &icause, &inext)) {
        if (ext(h, RVO_ISA_S) && h->priv != PRV_M && (h->satp >> 60) > 8) { stop(h, RVO_STOP_UNSUPPORTED, 0x180); break; }
        /* the interrupted instruction is read for the log only: PMA check, no PMP, no trap (Hart::readInst, Hart.cpp:2997) */
        u32 iw = read_inst_for_log(h, h->pc);
        take_interrupt(h, icause, inext, h->pc);
        if (!(h->mcountinhibit & 1)) h->cycle++;             /* Hart.cpp:7255-7256 */
        finish_commit(h, iw, (iw & 3) == 3 ? 4 : 2, priv0, 2, log, cap, n);
        continue;
      }
    }
    unsigned fcause = 1;
    int fr = fetch(h, h->pc, &inst, &tval, &fcause);
    if (fr == 2) break;
    if (fr == 1) {
      take_exception(h, fcause, h->curr_pc, tval);
      if (h->status != RVO_RUNNING) {                        /* consecutive-illegal stop inside a fetch fault */
        h->retire_count++; if (!(h->mcountinhibit & 4)) h->instret++;
        finish_commit(h, 0, 2, priv0, 0, log, cap, n); break;
      }
      if (!(h->mcountinhibit & 1)) h->cycle++;               /* fetchInstWithTrigger: Hart.cpp:5966-5967 */
      finish_commit(h, 0, 2, priv0, 1, log, cap, n);
      continue;
    }
    unsigned size = (inst & 3) == 3 ? 4 : 2;
    h->pc += size;                                           /* Hart.cpp:6120 */
    if (h->rv32) h->pc &= 0xffffffffULL;
    if (size == 2) {
      u32 e = ext(h, RVO_ISA_C) ? expand16(inst, h->rv32) : 0;
      if (e == 0) illegal(h, inst); else execute(h, e, inst, 2);
    } else
      execute(h, inst, inst, 4);
    if (h->status == RVO_STOP_UNSUPPORTED) break;
    if (h->status != RVO_RUNNING) {                          /* logStop (printTrace.cpp:965-976) */
      if (!(h->mcountinhibit & 4)) h->instret++;
      h->retire_count++;
      finish_commit(h, inst, size, priv0, 0, log, cap, n);
      break;
    }
    if (h->has_exc) { finish_commit(h, inst, size, priv0, 1, log, cap, n); continue; }   /* Hart.cpp:6128-6135 */
    if (!(h->mcountinhibit & 4)) h->instret++;               /* Hart.cpp:6158-6162 */
    h->retire_count++;
    finish_commit(h, inst, size, priv0, 0, log, cap, n);
  }
  return h->status;
}

/* ------------------------------------------------------------------ final-state hashes */
static const unsigned state_csrs[] = {
  C_FCSR, C_STVEC, C_SCOUNTEREN, C_SENVCFG, C_SSCRATCH, C_SEPC, C_SCAUSE, C_STVAL, C_SATP,
  C_MSTATUS, C_MEDELEG, C_MIDELEG, C_MIE, C_MTVEC, C_MCOUNTEREN, C_MENVCFG, C_MCOUNTINHIBIT,
  C_MSCRATCH, C_MEPC, C_MCAUSE, C_MTVAL, C_MIP, C_MCYCLE, C_MINSTRET
};

u64 rvo_state_hash(const rvo_hart* h)
{
  u64 s = 0;
  s = rvo_fold(s, h->pc);
  for (int i = 1; i < 32; ++i) s = rvo_fold(s, h->x[i]);
  for (int i = 0; i < 32; ++i) s = rvo_fold(s, h->f[i]);
  s = rvo_fold(s, h->priv);
  for (size_t i = 0; i < sizeof state_csrs / sizeof state_csrs[0]; ++i) {
    unsigned num = state_csrs[i];
    u64 v;
    switch (num) {  /* raw values, independent of which extensions expose the CSR */
      case C_FCSR: v = h->fcsr; break; case C_STVEC: v = h->stvec; break; case C_SCOUNTEREN: v = h->scounteren; break;
      case C_SENVCFG: v = h->senvcfg; break; case C_SSCRATCH: v = h->sscratch; break; case C_SEPC: v = h->sepc; break;
      case C_SCAUSE: v = h->scause; break; case C_STVAL: v = h->stval; break; case C_SATP: v = h->satp; break;
      case C_MSTATUS: v = h->mstatus; break; case C_MCOUNTEREN: v = h->mcounteren; break; case C_MEPC: v = h->mepc; break;
      default: v = csr_peek64(h, num); break;          /* (an RV32 hart's file is digested in its RV64 shape, x registers sign-extended) */
    }
    s = rvo_fold(s, v);
  }
  s = rvo_fold(s, h->time_);
  for (int i = 0; i < 8; ++i) s = rvo_fold(s, h->pmpcfg[i]);
  for (int i = 0; i < 64; ++i) s = rvo_fold(s, h->pmpaddr[i]);
  if (ext(h, RVO_ISA_S)) s = rvo_fold(s, rvo_tlb_hash(h));
  if (ext(h, RVO_ISA_V)) {
    s = rvo_fold(s, h->vstart); s = rvo_fold(s, (h->vxrm << 1) | h->vxsat); s = rvo_fold(s, h->vl); s = rvo_fold(s, h->vtype);
    for (unsigned w = 0; w < 32 * h->vlenb / 8; ++w) s = rvo_fold(s, ld_le(h->vreg + 8 * w, 8));
  }
  return s;
}

u64 rvo_mem_hash(const rvo_hart* h)
{
  u64 s = 0, nw = h->cfg.mem_size / 8;
  for (u64 i = 0; i < nw; ++i) {
    u64 w = ld_le(h->mem + i * 8, 8);
    s += rvo_mix(w ^ (GOLDEN * (i + 1)));
  }
  return s;
}
