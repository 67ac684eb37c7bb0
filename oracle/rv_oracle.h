/*
 * TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from the product
 * (whisper_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * rv_oracle: a scalar CPU restatement, in plain C, of the reference ISS hot path
 * (tenstorrent/whisper Hart<URV>::untilAddress / singleStep, /root/reference/Hart.cpp:6003,7417)
 * for one hart with a private RAM arena.  Each function cites the reference lines it follows.
 *
 * Pinning (see DESIGN.md "Oracle"): Hart.cpp itself cannot be compiled in this image (it needs
 * Boost headers the image lacks), so the restatement is pinned by
 *   - the captured reference commit logs in SURVEY.md Appendix D (tests/golden/appendix_d_*.json),
 *   - the reference's own Decoder/Disassembler and vendored SoftFloat compiled from
 *     /root/reference into oracle/_ref (this file *calls* that SoftFloat for all F/D arithmetic).
 */
#ifndef RV_ORACLE_H
#define RV_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ISA option bits (letters of the --isa string that matter to the hot path). */
enum {
  RVO_ISA_M = 1 << 0, RVO_ISA_A = 1 << 1, RVO_ISA_F = 1 << 2, RVO_ISA_D = 1 << 3,
  RVO_ISA_C = 1 << 4, RVO_ISA_S = 1 << 5, RVO_ISA_U = 1 << 6, RVO_ISA_V = 1 << 7,
  RVO_ISA_ZICNTR = 1 << 8,
  /* (bit 9 is the library's RV32 mark; the oracle takes the width from rvo_config.xlen) */
  RVO_ISA_ZBA = 1 << 10, RVO_ISA_ZBB = 1 << 11, RVO_ISA_ZBS = 1 << 12, RVO_ISA_ZICOND = 1 << 13, RVO_ISA_ZBC = 1 << 14,
  /* configuration switches (HartConfig::applyConfig tags, HartConfig.cpp:2085-2958); all clear = the reference's defaults */
  RVO_CFG_NO_MISAL = 1 << 16,          /* enable_misaligned_data = false */
  RVO_CFG_CLR_MTVAL_ILL = 1 << 17,     /* clear_mtval_on_illegal_instruction */
  RVO_CFG_CLR_MTVAL_EBREAK = 1 << 18,  /* clear_mtval_on_ebreak */
  RVO_CFG_CANCEL_LR_TRAP = 1 << 19,    /* cancel_lr_on_trap */
  RVO_CFG_KEEP_MPRV = 1 << 20,         /* clear_mprv_on_ret = false */
  RVO_CFG_NO_SV39 = 1 << 21            /* address_translation_modes without sv39 */
};

/* Stop reasons (mirror include/whisper_amd.h WSP_ST_*). */
enum {
  RVO_RUNNING = 0,
  RVO_STOP_TOHOST = 1,     /* store to tohost with bit0 set; value = HTIF word (Hart.cpp:2665)      */
  RVO_STOP_ILLEGAL = 2,    /* >16 consecutive no-di exceptions (Hart.cpp:3329-3340); value = 3      */
  RVO_STOP_LIMIT = 3,      /* execCount reached the instruction limit (Hart.cpp:6025)               */
  RVO_STOP_UNSUPPORTED = 4,/* outside the modelled subset; value = reason code                      */
  RVO_STOP_ADDR = 5        /* pc reached the stop address (Hart.cpp:6025, 6230)                     */
};

/* Kinds of change in a commit record (printTrace.cpp:388-493 order r,f,v,m,c). */
enum { RVO_CH_R = 1, RVO_CH_F = 2, RVO_CH_V = 3, RVO_CH_M = 4, RVO_CH_C = 5 };

typedef struct {
  uint32_t kind;
  uint32_t pad;
  uint64_t key;   /* register number / address / CSR number */
  uint64_t val;
} rvo_change;

#define RVO_MAX_CHANGES 16
typedef struct {
  uint64_t tag;        /* execCount_ */
  uint64_t pc;         /* currPc_ */
  uint32_t inst;       /* raw bits (16-bit masked when compressed; 0 on fetch fault) */
  uint8_t  size;       /* 2 or 4 */
  uint8_t  priv;       /* privilege the instruction executed in: 0 U, 1 S, 3 M */
  uint8_t  trap;       /* 1 exception, 2 interrupt */
  uint8_t  nchanges;
  rvo_change ch[RVO_MAX_CHANGES];
  /* scalar load / store / atomic: the addresses the log prints as "[va]" / "[va:pa]" (ldStAddr_, ldStPhysAddr1_; Hart.cpp:2306-2333,
   * 2678-2710, amo.cpp:68-97, 146-183, 277-321).  Not part of the trace digest. */
  uint64_t ld_va, ld_pa;
  uint32_t has_ld, pad2;
} rvo_commit;

typedef struct {
  uint32_t xlen;        /* 64 (32 reserved) */
  uint32_t isa;         /* RVO_ISA_* */
  uint64_t mem_base;    /* arena = [mem_base, mem_base+mem_size): RWX, everything else faults */
  uint64_t mem_size;
  uint64_t tohost;      /* valid iff tohost_valid */
  uint32_t tohost_valid;
  uint32_t hart_id;
  uint64_t reset_pc;
  uint32_t tlb_entries; /* TLB size, a power of two; 0 = the reference default 2048 (Hart.cpp:106) */
  uint32_t vlenb;       /* bytes per vector register: 16, 32 or 64; 0 = the reference default 64 (Hart.cpp:1081) */
} rvo_config;

typedef struct rvo_hart rvo_hart;

rvo_hart* rvo_create(const rvo_config* cfg);
void      rvo_destroy(rvo_hart* h);
/* Point the oracle at the reference SoftFloat (oracle/_ref/libsoftfloat_ref.so). Returns 0 on
 * success. Must be called before any F/D instruction executes. */
int       rvo_bind_softfloat(const char* path);

int       rvo_write_mem(rvo_hart* h, uint64_t addr, const void* src, size_t n);
int       rvo_read_mem(const rvo_hart* h, uint64_t addr, void* dst, size_t n);
void      rvo_set_pc(rvo_hart* h, uint64_t pc);
uint64_t  rvo_get_pc(const rvo_hart* h);
void      rvo_set_x(rvo_hart* h, unsigned i, uint64_t v);
uint64_t  rvo_get_x(const rvo_hart* h, unsigned i);
void      rvo_set_f(rvo_hart* h, unsigned i, uint64_t v);
uint64_t  rvo_get_f(const rvo_hart* h, unsigned i);
/* peek/poke of the modelled CSRs by number; returns 0 when the CSR is modelled. */
int       rvo_peek_csr(const rvo_hart* h, unsigned num, uint64_t* v);
int       rvo_poke_csr(rvo_hart* h, unsigned num, uint64_t v);
unsigned  rvo_priv(const rvo_hart* h);
/* vector register i (vlenb bytes, element 0 first); returns vlenb */
unsigned  rvo_get_vreg(const rvo_hart* h, unsigned i, void* dst);
void      rvo_set_vreg(rvo_hart* h, unsigned i, const void* src);

/* Run until a stop condition or until `max_exec` instructions have been executed in total
 * (Hart::setInstructionCountLimit semantics: counts trapped instructions too).  If `log` is
 * non-NULL up to `log_cap` commit records are stored and *log_n updated. Returns status. */
int       rvo_run(rvo_hart* h, uint64_t max_exec, rvo_commit* log, size_t log_cap, size_t* log_n);

/* retCountLim_ and the stop address of Hart::untilAddress (Hart.cpp:6011, 6025-6026): max_ret = UINT64_MAX for none */
void      rvo_resume(rvo_hart* h);   /* a hart stopped by a count limit or its stop address runs again */
void      rvo_set_stop(rvo_hart* h, uint64_t max_ret, int stop_pc_valid, uint64_t stop_pc);

int       rvo_status(const rvo_hart* h);
uint64_t  rvo_stop_value(const rvo_hart* h);
uint64_t  rvo_exec_count(const rvo_hart* h);
uint64_t  rvo_retired_count(const rvo_hart* h);
uint64_t  rvo_trace_hash(const rvo_hart* h);
uint64_t  rvo_state_hash(const rvo_hart* h);
uint64_t  rvo_mem_hash(const rvo_hart* h);

/* The hash primitives the device kernels must reproduce bit-for-bit. */
/* PMP permission bits (r=1 w=2 x=4) of privilege pm at addr: test hook for pinning against the reference PmpManager */
unsigned  rvo_pmp_perm(const rvo_hart* h, unsigned pm, uint64_t addr);
/* VirtMem::translate (VirtMem.cpp:261) for privilege pm (0 U, 1 S) and access acc (1 read, 2 write, 4 exec) under the
 * hart's current satp / mstatus.SUM / MXR: returns 0 and *pa, or the exception cause.  Test hook for pinning against
 * the reference VirtMem + Tlb. */
unsigned  rvo_translate(rvo_hart* h, uint64_t va, unsigned pm, unsigned acc, uint64_t* pa);
unsigned  rvo_translate_noupdate(rvo_hart* h, uint64_t va, unsigned pm, unsigned acc, uint64_t* pa);   /* VirtMem::transAddrNoUpdate */
/* Tlb invalidation as execSfence_vma performs it (Hart.cpp:12139-12173): has_addr / has_asid = rs1 / rs2 is not x0 */
void      rvo_sfence(rvo_hart* h, int has_addr, uint64_t addr, int has_asid, uint64_t asid);
/* digest of the TLB contents (same packing as the device) */
uint64_t  rvo_tlb_hash(const rvo_hart* h);
/* Test hooks over the vector restatement (rv_oracle_vec.inc) for pinning against the reference's VecRegs / float-util.hpp:
 * what = 0 dest-active(vd, ix, log2 bytes, masked) -> active, *out = value to write when inactive; 1 valid-index(reg, ix,
 * groupx8, log2 bytes); 2 elemMax(eew); 3 vlmax; 4 mask bit (reg, ix); 5 write mask bit (reg, ix, v); 6 groupMultiplierX8(gm);
 * 7 legalConfig(sew, gm); 8 write element (reg, ix, log2 bytes, groupx8) = *out. */
int       rvo_vec_probe(rvo_hart* h, unsigned what, uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t* out);
uint64_t  rvo_vf_probe(unsigned kind, unsigned esz, uint64_t a, uint64_t b, uint64_t c, unsigned rm, unsigned* flags);
uint64_t  rvo_fold(uint64_t h, uint64_t w);
uint64_t  rvo_mix(uint64_t x);

/* test hook: Decoder::expandCompressedInst restated (0 = illegal) */
uint32_t rvo_expand16(uint32_t c, int rv32);

#ifdef __cplusplus
}
#endif
#endif
