/*
 * whisper_amd C ABI: batched RISC-V hart execution on MI355X (gfx950).
 *
 * The reference (tenstorrent/whisper) has no FFI for its execution path: every front end drives
 * the C++ classes System<URV>/Hart<URV> directly.  Each entry point below
 * names the reference call(s) it stands in for; file:line are into /root/reference.  A batch is
 * N independent single-hart systems, each with a private RAM arena [mem_base, mem_base+mem_size);
 * all run in lock step on one HIP device.  Plain C types only, caller-owned buffers, 0 or a
 * negative WSP_E* code returned, message via wsp_last_error() (mirrors the reference's
 * bool + "Error:" line on stderr).  No exceptions cross the ABI.
 *
 * The run entry points are HIP-only.  There is no CPU fallback: with no usable device
 * wsp_batch_create fails with WSP_ENODEV.
 */
#ifndef WHISPER_AMD_H
#define WHISPER_AMD_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wsp_batch wsp_batch;

enum { WSP_OK = 0, WSP_EINVAL = -1, WSP_ENODEV = -2, WSP_ENOMEM = -3, WSP_EIO = -4, WSP_EFORMAT = -5,
       WSP_ERANGE = -6, WSP_EUNSUPPORTED = -7, WSP_EHIP = -8 };

/* Hart stop states (wsp_result.status). */
enum { WSP_ST_RUNNING = 0,
       WSP_ST_STOP_TOHOST = 1,      /* store to tohost, bit 0 set (Hart.cpp:2665); success iff (value>>1)==0 (printTrace.cpp:943) */
       WSP_ST_STOP_ILLEGAL = 2,     /* >16 consecutive illegal/no-di exceptions (Hart.cpp:3329-3340), value 3 */
       WSP_ST_STOP_LIMIT = 3,       /* instruction-count limit reached (Hart.cpp:6025, setInstructionCountLimit) */
       WSP_ST_STOP_UNSUPPORTED = 4, /* left the modelled subset; stop_value holds the reason */
       WSP_ST_STOP_ADDR = 5         /* pc reached the stop address before executing it (Hart.cpp:6025, 6230) */ };

/* Resources for peek/poke (Hart::peekIntReg/pokeIntReg ... Hart.hpp:219-325, Hart.cpp:1345-1560). */
/* WSP_RES_PC takes an index: 0 pc; 1 executed-instruction count (the log's tag, the snapshot's "po"); 2 retired-instruction count
 * ("pr"); 3 privilege mode ("pm": 0 U, 1 S, 3 M; Hart::privilegeMode / setPrivilegeMode, snapshot.cpp:40, 270); 4 (peek only) the
 * hart's stop state WSP_ST_* (Hart::hasTargetProgramFinished is `== WSP_ST_STOP_TOHOST`). */
enum { WSP_RES_PC = 0, WSP_RES_X = 1, WSP_RES_F = 2, WSP_RES_CSR = 3, WSP_RES_MEM = 4, WSP_RES_V = 5 /* whole vector register, vlen_bytes bytes */ };

/* wsp_batch_run flags */
enum { WSP_RUN_TRACE_HASH = 1 /* fold every commit record (printTrace.cpp:302-517) into trace_hash */ };
/* wsp_batch_load_image flags */
enum { WSP_LOAD_EXEC = 1 /* bytes are code: pre-decode them */ };

typedef struct {
  uint32_t xlen;            /* 64, or 0 = from config_json / the ISA string (HartConfig::getXlen, HartConfig.cpp:3431) */
  uint32_t n_harts;         /* number of independent single-hart systems                                   */
  const char* isa;          /* e.g. "rv64imafdc_zicsr" (Hart::configIsa, Hart.cpp:5122; Isa.cpp:561); NULL = config tag "isa" */
  uint64_t mem_base;        /* arena base; accesses outside the arena take access faults (PMA no-access)   */
  uint64_t mem_size;        /* bytes per hart, multiple of 4096 (System ctor memSize, System.cpp:50); 0 = config tag memmap.size */
  uint32_t lanes_per_wave;  /* 0 = auto; else a power of two <= 64: harts per 64-lane wavefront           */
  int32_t  device;          /* HIP device ordinal, -1 = current                                            */
  uint32_t hart_id_base;    /* mhartid of hart 0 (core_hart_id_offset)                                     */
  uint32_t tlb_entries;     /* TLB entries per hart, a power of two; 0 = 2048 (Hart.cpp:106; config tag tlb_entries) */
  uint32_t vlen_bytes;      /* bytes per vector register: 16, 32 or 64; 0 = 64 (Hart.cpp:1081; config tag vector.bytes_per_vec) */
  uint32_t reserved;
  const char* config_json;  /* text of a Whisper JSON configuration file, or NULL (HartConfig::loadConfigFile + applyConfig,
                               HartConfig.cpp:61, 2085).  Tags that change what a hart executes are honoured; tags the device does
                               not model are rejected with WSP_EUNSUPPORTED, never ignored; non-zero descriptor fields override the
                               file (as command-line options do, Session.cpp:1373-1424).  See wsp_config_parse. */
} wsp_batch_desc;

/* What a configuration file asks for, as wsp_batch_create understands it (diagnostic / test surface: the same parser).
 * features: bit 16 enable_misaligned_data = false, 17 clear_mtval_on_illegal_instruction, 18 clear_mtval_on_ebreak,
 * 19 cancel_lr_on_trap, 20 clear_mprv_on_ret = false, 21 address_translation_modes without sv39. */
typedef struct {
  uint32_t xlen, features;
  uint64_t mem_size, reset_pc;
  uint32_t has_reset_pc, tlb_entries, vlen_bytes, abi_names, n_csr_resets, reserved;
  uint32_t csr_num[32]; uint64_t csr_value[32];
  char isa[256];
} wsp_config_info;
/* HartConfig::loadConfigFile + applyConfig (HartConfig.cpp:61, 2085) on a text: 0, or WSP_EFORMAT (not JSON), WSP_EINVAL (a value
 * of the wrong kind), WSP_EUNSUPPORTED (a tag the device does not model; wsp_last_error names it). */
int  wsp_config_parse(const char* json_text, wsp_config_info* out);

typedef struct {
  uint32_t status;          /* WSP_ST_*                                                                    */
  uint32_t priv;            /* final privilege mode (0 U, 1 S, 3 M)                                        */
  uint64_t stop_value;      /* tohost value / 3 / unsupported reason                                       */
  uint64_t pc;
  uint64_t exec_count;      /* Hart::getInstructionCount (execCount_: traps included)                      */
  uint64_t retired_count;   /* Hart::getRetiredInstructionCount                                            */
  uint64_t state_hash;      /* digest of pc, x1..x31, f0..f31, privilege and the modelled CSRs             */
  uint64_t mem_hash;        /* digest of the hart's whole RAM arena                                        */
  uint64_t trace_hash;      /* digest of all commit records (0 unless WSP_RUN_TRACE_HASH)                  */
} wsp_result;

/* One pre-decoded instruction as produced by the device decoder (test/diagnostic surface). */
typedef struct { uint32_t w0; int32_t imm; uint32_t pad; uint32_t raw; } wsp_decoded;

/* System ctor + HartConfig::configHarts/configMemory + Hart::configIsa + Hart::reset
 * (System.cpp:50, HartConfig.cpp:3282/3414, Hart.cpp:5122, 942). */
int  wsp_batch_create(const wsp_batch_desc* desc, wsp_batch** out);
void wsp_batch_destroy(wsp_batch* b);

/* System::loadElfFiles (System.cpp:288) for harts [hart_lo, hart_hi): PT_LOAD by physical address,
 * p_filesz bytes (Memory.cpp:450-489); pc=entry, gp=__global_pointer$, tp=.tdata (System.cpp:334-354);
 * tohost symbol -> stop address (System.cpp:323). */
int  wsp_batch_load_elf(wsp_batch* b, uint32_t hart_lo, uint32_t hart_hi, const char* path);
/* Same from an in-memory ELF image. */
int  wsp_batch_load_elf_mem(wsp_batch* b, uint32_t hart_lo, uint32_t hart_hi, const void* elf, size_t n);
/* System::loadElfFiles (System.cpp:288) for `count` systems at once: ELF image i goes to the harts_per_elf harts starting at
 * hart_lo + i * harts_per_elf.  Same result as `count` calls of wsp_batch_load_elf_mem, with one staging upload, one scatter
 * launch and one pre-decode launch for all of them (BASELINE configs[4]: 125 000 distinct programs per GPU). */
int  wsp_batch_load_elf_many(wsp_batch* b, uint32_t hart_lo, uint32_t count, const void* const* elfs, const size_t* sizes, uint32_t harts_per_elf);
/* System::loadHexFiles/loadBinaryFiles / Hart::pokeMemory (System.cpp:363,434; Hart.cpp:1345). */
int  wsp_batch_load_image(wsp_batch* b, uint32_t hart_lo, uint32_t hart_hi, uint64_t addr, const void* bytes, size_t n, uint32_t flags);
/* Address of an ELF symbol of the last ELF loaded into `hart` (Memory::findElfSymbol). */
int  wsp_batch_symbol(wsp_batch* b, uint32_t hart, const char* name, uint64_t* addr);

/* Hart::poke* / peek* for one hart. idx = register number, CSR number or byte address; size in
 * bytes (8 for PC/X/F/CSR, 1..n for MEM). */
int  wsp_batch_poke(wsp_batch* b, uint32_t hart, uint32_t res, uint64_t idx, const void* val, size_t size);
int  wsp_batch_peek(wsp_batch* b, uint32_t hart, uint32_t res, uint64_t idx, void* val, size_t size);
/* Poke a different 64-bit value into the same X register / memory dword of every hart in
 * [hart_lo, hart_hi) (per-instance seeds); vals has hart_hi-hart_lo entries. */
int  wsp_batch_poke_each(wsp_batch* b, uint32_t hart_lo, uint32_t hart_hi, uint32_t res, uint64_t idx, const uint64_t* vals);

/* Hart::setToHostAddress (Hart.cpp:1626) + setInstructionCountLimit / setRetiredInstructionCountLimit (Hart.hpp:1385) + the stop
 * address of Hart::untilAddress / setStopAddress (Hart.cpp:6003, 6025-6026, 6256): the loop condition
 * `pc != stop_pc and executed < max_inst and retired < max_ret`.  tohost = UINT64_MAX leaves it as it is; max_inst / max_ret =
 * UINT64_MAX mean no limit; stop_pc = UINT64_MAX means none.  Limits end a hart with WSP_ST_STOP_LIMIT, the address with
 * WSP_ST_STOP_ADDR (the instruction at stop_pc is not executed). */
int  wsp_batch_set_stop(wsp_batch* b, uint32_t hart_lo, uint32_t hart_hi, uint64_t tohost, uint64_t max_inst, uint64_t max_ret, uint64_t stop_pc);

/* Harts of [hart_lo, hart_hi) that ended with WSP_ST_STOP_LIMIT or WSP_ST_STOP_ADDR become runnable again (set new limits / a new
 * stop address first): Hart::run / untilAddress / singleStep called again on a hart that had reached its limit, as the interactive
 * "step" / "until" commands and a park-and-service loop do (Hart.cpp:6003, 6963; Interactive.cpp stepCommand / untilCommand).
 * Harts that wrote tohost, hit consecutive illegal instructions or left the modelled subset stay stopped. */
int  wsp_batch_resume(wsp_batch* b, uint32_t hart_lo, uint32_t hart_hi);

/* System::batchRun / Hart::run (System.cpp:1654, Hart.cpp:6963): step every running hart until it
 * stops or has executed max_steps more instructions. Returns the number of harts still running
 * (>= 0) or a negative error. */
int64_t wsp_batch_run(wsp_batch* b, uint64_t max_steps, uint32_t flags);
/* Seconds the device spent inside the step kernel during the last wsp_batch_run (HIP events on
 * the batch's stream) and the number of kernel launches it took. */
int  wsp_batch_last_run_stats(wsp_batch* b, double* kernel_seconds, uint32_t* launches);

/* Fill out[0 .. n_harts) (computes the state and memory digests on the device). */
int  wsp_batch_results(wsp_batch* b, wsp_result* out);
/* wsp_batch_results for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that runs
 * while another stream is being captured into a HIP graph).  It waits for the batch's stream before it returns, so the call
 * itself cannot be part of a capture.  No counterpart in the reference.  Calls no hipMalloc / hipFree or allocating library routine: the
 * digest's partial sums live in a static device array of the library (8 MiB per GPU, held until the process ends) and the
 * n_harts records are written into d_out, which is DEVICE memory supplied by the caller.  Limit: n_waves * 64 <= 1048576
 * hart slots, else WSP_ERANGE.  All calls on one GPU share that array: calls that overlap on one GPU race. */
int  wsp_batch_results_noalloc(wsp_batch* b, wsp_result* d_out);
/* Seconds spent in the memory-digest kernel during the last wsp_batch_results and its HBM bytes. */
int  wsp_batch_last_hash_stats(wsp_batch* b, double* kernel_seconds, uint64_t* bytes);

/* Commit ring (the binary side of the reference's --logfile, printTrace.cpp:380-495; Hart::printInstTrace).  With
 * words_per_hart > 0 every run started with WSP_RUN_TRACE_HASH also appends each hart's commit records to a ring of that
 * many 64-bit words (recording stops when it is full; the digest goes on; every step takes the general path while a ring
 * is on).  Record: pc, meta = inst | priv << 32 | trap << 40 | size << 48 | ldst << 55 | npairs << 56, kinds, then npairs x (id,
 * value): pair k has kind = (kinds >> 4k) & 15: 1 x-reg, 2 f-reg, 3 v-reg (value = digest of the register), 4 memory (id = the
 * whole 64-bit address), 5 CSR, in the order the reference's log prints them; then, outside the digest, what only the text log
 * needs: if ldst, the virtual and the physical address of the scalar load / store / atomic (2 words), and for every v-reg pair
 * the vlen_bytes / 8 words of the register.  0 switches it off.  Calling it again clears the ring.  Not part of snapshot /
 * restore. */
int wsp_batch_trace_ring(wsp_batch* b, uint32_t words_per_hart);
/* Copies the words recorded for one hart (at most cap); returns their count, or a negative error. */
int64_t wsp_batch_trace_read(wsp_batch* b, uint32_t hart, uint64_t* words, size_t cap);
/* The recorded commits of one hart as the reference's --logfile text (Hart::printDecodedInstTrace, printTrace.cpp:302-517, line
 * layouts printTrace.cpp:56-209 for RV32 / RV64; disassembly column as Disassembler.cpp prints it): one line per changed resource
 * in the order r f v m c, continuation lines joined by "  +" (" +" before a vector register line), "[va]" / "[va:pa]" after
 * scalar loads, stores and atomics, "(exception)" / "(interrupted)" marks, vector registers with their bytes.  first_tag is the
 * instruction number of the first recorded commit (1 for a ring switched on before the first run).  Writes at most cap - 1
 * characters and a NUL; returns the length of the whole text, or a negative error.  Not printed: the element address lists of
 * vector loads / stores and page-table-walk lines (--traceptw). */
int64_t wsp_batch_trace_text(wsp_batch* b, uint32_t hart, uint64_t first_tag, char* text, size_t cap);
/* Disassembler::disassembleInst (Disassembler.cpp:519): the log's disassembly column for one instruction word as fetched (a 16-bit
 * instruction in the low half); xlen 32 or 64.  Needs no device.  Returns the text length. */
int     wsp_disassemble(uint32_t inst, uint32_t xlen, char* text, size_t cap);

/* System::produceTestSignatureFile (System.cpp:1586-1633): 32-bit words from begin_signature to
 * end_signature of `hart`; writes up to cap words, returns the word count or a negative error. */
int64_t wsp_batch_signature(wsp_batch* b, uint32_t hart, uint32_t* words, size_t cap);

/* Checkpoint / resume of the whole batch on the device (the role of System::saveSnapshot and
 * --loadfrom, System.cpp:562, snapshot.cpp:17: here the device state arrays are the checkpoint).
 * snapshot keeps one device-resident copy; restore rewinds every hart to it. */
int  wsp_batch_snapshot(wsp_batch* b);
int  wsp_batch_restore(wsp_batch* b);

/* Run the device decoder (the one the step kernel uses) over n instruction words. */
int  wsp_decode_words(const char* isa, const uint32_t* words, size_t n, wsp_decoded* out);
/* The same decoder compiled for the host (no device needed; used by CPU-only tests). */
int  wsp_decode_words_host(const char* isa, const uint32_t* words, size_t n, wsp_decoded* out);
/* Run the device build of the softfloat (the arithmetic behind float.cpp / float-util.hpp:230-930, which the reference gets from
 * third_party/softfloat) over n operand triples on the GPU: out[i], flags[i] = op(a[i], b[i], c[i]) under rounding mode rm (0..4,
 * RISC-V order).  op & 0x100 selects double precision; low byte: 0 add 1 sub 2 mul 3 div 4 sqrt 5 fused multiply-add,
 * 6..9 float -> integer (bit 0 signed, bit 1 64-bit), 10..13 integer c -> float (same bits), 14 convert to the other format.
 * op & 0x200: evaluate the way the step kernel does for double precision under round-to-nearest-even - on the FP unit with
 * exact flags where the operands allow (flags[i] bit 31 set then), the integer routine otherwise. */
int  wsp_sf_eval(uint32_t op, uint32_t rm, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t n, uint64_t* out, uint32_t* flags);
/* Parse an ISA string into the internal option bit set (tests). */
uint32_t wsp_isa_flags(const char* isa);

const char* wsp_last_error(void);
const char* wsp_version(void);

#ifdef __cplusplus
}
#endif
#endif
