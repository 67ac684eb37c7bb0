// Whisper JSON configuration front end (host only): the part of HartConfig::loadConfigFile / applyConfig / configHarts /
// configMemory (/root/reference/HartConfig.cpp:61-92, 2085-2958, 3282-3430) that concerns the batched execution path.
//
// Every tag falls into one of three classes:
//   honoured  - changes what the device executes (isa, xlen, memmap.size, reset_vec, tlb_entries, vector.bytes_per_vec,
//               enable_misaligned_data, clear_mtval_on_illegal_instruction, clear_mtval_on_ebreak, cancel_lr_on_trap,
//               clear_mprv_on_ret, address_translation_modes, csr.<name>.reset, abi_names for the text log);
//   accepted  - has no effect on this path (log formatting, host-side features, NMI / debug plumbing that nothing in a batch
//               can trigger), or states the value the device models anyway;
//   rejected  - WSP_EUNSUPPORTED naming the tag: a semantic switch the device does not model.  Nothing is ignored silently.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <utility>

namespace wspcfg {

struct JVal {
  enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
  bool b = false; double num = 0; uint64_t u = 0; bool is_int = false; bool neg = false;
  std::string s;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal* find(const char* k) const { for (auto& kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
};

// JSON with // and /* */ comments (the reference parses with ignore_comments = true, HartConfig.cpp:77)
struct JParser {
  const char* p; const char* e; std::string err;
  void ws()
  {
    for (;;) {
      while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
      if (p + 1 < e && p[0] == '/' && p[1] == '/') { while (p < e && *p != '\n') ++p; continue; }
      if (p + 1 < e && p[0] == '/' && p[1] == '*') { p += 2; while (p + 1 < e && !(p[0] == '*' && p[1] == '/')) ++p; p = p + 1 < e ? p + 2 : e; continue; }
      return;
    }
  }
  bool fail(const char* m) { if (err.empty()) err = std::string("JSON: ") + m; return false; }
  bool str(std::string& out)
  {
    if (p >= e || *p != '"') return fail("expected string");
    ++p; out.clear();
    while (p < e && *p != '"') {
      if (*p == '\\') {
        if (++p >= e) return fail("bad escape");
        switch (*p) {
          case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break;
          case 'f': out += '\f'; break; case '/': out += '/'; break; case '\\': out += '\\'; break; case '"': out += '"'; break;
          case 'u': { if (e - p < 5) return fail("bad \\u escape"); char h[5] = { p[1], p[2], p[3], p[4], 0 }; unsigned c = (unsigned)strtoul(h, nullptr, 16);
                      if (c < 0x80) out += (char)c; else if (c < 0x800) { out += (char)(0xc0 | (c >> 6)); out += (char)(0x80 | (c & 0x3f)); }
                      else { out += (char)(0xe0 | (c >> 12)); out += (char)(0x80 | ((c >> 6) & 0x3f)); out += (char)(0x80 | (c & 0x3f)); }
                      p += 4; break; }
          default: return fail("bad escape");
        }
        ++p;
      } else out += *p++;
    }
    if (p >= e) return fail("unterminated string");
    ++p; return true;
  }
  bool val(JVal& v, int depth)
  {
    if (depth > 64) return fail("nesting too deep");
    ws();
    if (p >= e) return fail("unexpected end");
    if (*p == '{') {
      v.t = JVal::OBJ; ++p; ws();
      if (p < e && *p == '}') { ++p; return true; }
      for (;;) {
        ws(); std::string k; if (!str(k)) return false;
        ws(); if (p >= e || *p != ':') return fail("expected ':'"); ++p;
        JVal c; if (!val(c, depth + 1)) return false;
        bool dup = false;
        for (auto& kv : v.obj) if (kv.first == k) { kv.second = c; dup = true; }
        if (!dup) v.obj.emplace_back(k, c);
        ws(); if (p < e && *p == ',') { ++p; continue; }
        if (p < e && *p == '}') { ++p; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (*p == '[') {
      v.t = JVal::ARR; ++p; ws();
      if (p < e && *p == ']') { ++p; return true; }
      for (;;) {
        JVal c; if (!val(c, depth + 1)) return false;
        v.arr.push_back(c);
        ws(); if (p < e && *p == ',') { ++p; continue; }
        if (p < e && *p == ']') { ++p; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (*p == '"') { v.t = JVal::STR; return str(v.s); }
    if (e - p >= 4 && !strncmp(p, "true", 4)) { v.t = JVal::BOOL; v.b = true; p += 4; return true; }
    if (e - p >= 5 && !strncmp(p, "false", 5)) { v.t = JVal::BOOL; v.b = false; p += 5; return true; }
    if (e - p >= 4 && !strncmp(p, "null", 4)) { v.t = JVal::NUL; p += 4; return true; }
    if (*p == '-' || (*p >= '0' && *p <= '9')) {
      const char* q = p; bool isint = true;
      if (*q == '-') ++q;
      while (q < e && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) { if (*q == '.' || *q == 'e' || *q == 'E') isint = false; ++q; }
      std::string t(p, q);
      v.t = JVal::NUM; v.is_int = isint; v.neg = t[0] == '-';
      v.num = strtod(t.c_str(), nullptr);
      v.u = isint ? (v.neg ? (uint64_t)strtoll(t.c_str(), nullptr, 10) : strtoull(t.c_str(), nullptr, 10)) : (uint64_t)v.num;
      p = q; return true;
    }
    return fail("unexpected character");
  }
};

inline bool parse_json(const char* text, JVal& out, std::string& err)
{
  JParser ps; ps.p = text; ps.e = text + strlen(text);
  if (!ps.val(out, 0)) { err = ps.err; return false; }
  ps.ws();
  if (ps.p != ps.e) { err = "JSON: trailing characters"; return false; }
  return true;
}

// getJsonUnsigned / getJsonBoolean (HartConfig.cpp:101-137, 203-238): numbers, or strings with an optional 0x prefix
inline bool get_unsigned(const JVal& v, uint64_t& out)
{
  if (v.t == JVal::NUM) { out = v.u; return true; }
  if (v.t == JVal::STR) { char* end = nullptr; out = strtoull(v.s.c_str(), &end, 0); return !(end && *end) && !v.s.empty(); }
  return false;
}
inline bool get_bool(const JVal& v, bool& out)
{
  if (v.t == JVal::BOOL) { out = v.b; return true; }
  if (v.t == JVal::NUM) { out = v.u != 0; return true; }
  if (v.t == JVal::STR) {
    if (v.s == "0" || v.s == "false" || v.s == "False") { out = false; return true; }
    if (v.s == "1" || v.s == "true" || v.s == "True") { out = true; return true; }
  }
  return false;
}

struct CsrReset { uint32_t num; uint64_t value; };

struct Parsed {
  uint32_t xlen = 0;              // 0: not given
  std::string isa;                // "": not given
  uint64_t mem_size = 0;          // memmap.size; 0: not given
  bool has_reset_pc = false; uint64_t reset_pc = 0;
  uint32_t tlb_entries = 0, vlen_bytes = 0;
  uint32_t features = 0;          // wsp::CFG_* bits
  bool abi_names = false;
  std::vector<CsrReset> csr;
};

enum { R_OK = 0, R_EINVAL = -1, R_EUNSUPPORTED = -7 };   // = WSP_OK / WSP_EINVAL / WSP_EUNSUPPORTED

// CSR names of the modelled set (CsRegs.cpp defineMachineRegs / defineSupervisorRegs / defineUserRegs / defineVectorRegs)
inline int csr_by_name(const std::string& n)
{
  static const struct { const char* nm; int num; } tab[] = {
    {"fflags", 0x001}, {"frm", 0x002}, {"fcsr", 0x003}, {"vstart", 0x008}, {"vxsat", 0x009}, {"vxrm", 0x00a}, {"vcsr", 0x00f},
    {"sstatus", 0x100}, {"sie", 0x104}, {"stvec", 0x105}, {"scounteren", 0x106}, {"senvcfg", 0x10a}, {"sscratch", 0x140},
    {"sepc", 0x141}, {"scause", 0x142}, {"stval", 0x143}, {"sip", 0x144}, {"satp", 0x180},
    {"mstatus", 0x300}, {"misa", 0x301}, {"medeleg", 0x302}, {"mideleg", 0x303}, {"mie", 0x304}, {"mtvec", 0x305},
    {"mcounteren", 0x306}, {"menvcfg", 0x30a}, {"mcountinhibit", 0x320}, {"mscratch", 0x340}, {"mepc", 0x341},
    {"mcause", 0x342}, {"mtval", 0x343}, {"mip", 0x344}, {"mcycle", 0xb00}, {"minstret", 0xb02},
    {"mvendorid", 0xf11}, {"marchid", 0xf12}, {"mimpid", 0xf13}, {"mhartid", 0xf14}, {"mconfigptr", 0xf15},
  };
  for (auto& t : tab) if (n == t.nm) return t.num;
  if (n.rfind("pmpcfg", 0) == 0 && n.size() > 6) { char* end = nullptr; long k = strtol(n.c_str() + 6, &end, 10); if (!*end && k >= 0 && k < 16) return 0x3a0 + (int)k; }
  if (n.rfind("pmpaddr", 0) == 0 && n.size() > 7) { char* end = nullptr; long k = strtol(n.c_str() + 7, &end, 10); if (!*end && k >= 0 && k < 64) return 0x3b0 + (int)k; }
  return -1;
}

// feature bits: keep in step with wsp::CFG_* (wsp_core.h)
enum : uint32_t { F_NO_MISAL = 1u << 16, F_CLR_MTVAL_ILL = 1u << 17, F_CLR_MTVAL_EBREAK = 1u << 18, F_CANCEL_LR_TRAP = 1u << 19,
                  F_KEEP_MPRV = 1u << 20, F_NO_SV39 = 1u << 21 };

inline int apply(const JVal& root, Parsed& out, std::string& err)
{
  if (root.t != JVal::OBJ) { err = "config: top level must be an object"; return R_EINVAL; }
  auto bad = [&](const std::string& tag, const char* what) { err = "config tag '" + tag + "': " + what; return R_EINVAL; };
  auto unsup = [&](const std::string& tag, const char* what) { err = "config tag '" + tag + "' is not supported: " + what; return R_EUNSUPPORTED; };
  // tags with no effect on the batched execution path
  static const char* const no_effect[] = {
    "trace_ptw", "print_load_store_address", "trace_pmp", "trace_pma", "enable_csv_log", "snapshot_periods",   // logging / host features
    "nmi_vec", "nmi_exception_vec", "indexed_nmi", "non_maskable_interrupts",                                  // nothing in a batch can raise an NMI
    "debug_park_loop", "debug_trap_address", "cancel_lr_on_debug",                                              // debug mode is unreachable
    "perf_count_atomic_load_store", "perf_count_fp_load_store", "perf_count_fp_load_store_as_fp", "perf_count_vec_load_store_as_vec",
    "enable_performance_counters", "page_fault_on_first_access",                                                // deprecated no-ops in the reference too
    "enable_zba", "enable_zbb", "enable_zbc", "enable_zbs", "enable_zfh", "enable_zfhmin", "enable_zknd", "enable_zkne", "enable_zknh",
    "enable_zbkb", "enable_zbkx", "enable_zksed", "enable_zksh", "enable_zbe", "enable_zbf", "enable_zbm", "enable_zbp", "enable_zbr", "enable_zbt",
    "core_hart_id_offset",                                                                                      // one hart per system: mhartid comes from hart_id_base
    "merge_buffer_line_size", "merge_buffer_check_all", "enable_mcm_cache", "enable_memory_consistency", "enable_tso", "enable_ppo",   // MCM checker (host side)
  };
  // boolean tags accepted only at the value the device models (the reference's default)
  static const struct { const char* tag; bool modelled; } fixed_bool[] = {
    {"enable_pmp_tor", true}, {"enable_pmp_na4", true}, {"misaligned_has_priority", true}, {"in_sequence_misaligned", false},
    {"keep_reservation_on_sc_exception", false}, {"enable_translation_pbmt", false}, {"enable_translation_napot", false},
    {"enable_svinval", false}, {"enable_supervisor_time_compare", false}, {"enable_aia", false}, {"enable_smstateen", false},
    {"enable_triggers", false}, {"enable_counter_overflow", false}, {"coherent_icache", false}, {"enable_semihosting", false},
    {"hfence_gvma_ignores_gpa", false}, {"mark_dirty_gstage_for_vs_nonleaf_pte", false}, {"auto_increment_timer", false},
    {"can_receive_interrupts", true}, {"clear_tinst_on_cbo_inval", false}, {"clear_tinst_on_cbo_flush", false}, {"align_cbo_address", false},
    {"trap_non_zero_vstart", true},
  };
  for (auto& kv : root.obj) {
    const std::string& tag = kv.first; const JVal& v = kv.second;
    uint64_t u = 0; bool f = false;
    bool done = false;
    for (const char* t : no_effect) if (tag == t) { done = true; break; }
    if (done) continue;
    for (auto& fb : fixed_bool) if (tag == fb.tag) {
      if (!get_bool(v, f)) return bad(tag, "expecting a boolean");
      if (f != fb.modelled) return unsup(tag, "only the reference's default value is modelled");
      done = true; break;
    }
    if (done) continue;
    if (tag == "isa") { if (v.t != JVal::STR) return bad(tag, "expecting a string"); out.isa = v.s; }
    else if (tag == "xlen") { if (!get_unsigned(v, u) || (u != 32 && u != 64)) return bad(tag, "expecting 32 or 64"); out.xlen = (uint32_t)u; }
    else if (tag == "cores" || tag == "harts") {
      if (!get_unsigned(v, u)) return bad(tag, "expecting a number");
      if (u != 1) return unsup(tag, "a batch is N independent single-hart systems; shared-memory multi-hart systems are out of scope");
    }
    else if (tag == "reset_vec") { if (!get_unsigned(v, u)) return bad(tag, "expecting a number"); out.has_reset_pc = true; out.reset_pc = u; }
    else if (tag == "abi_names") { if (!get_bool(v, f)) return bad(tag, "expecting a boolean"); out.abi_names = f; }
    else if (tag == "tlb_entries") {
      if (!get_unsigned(v, u)) return bad(tag, "expecting a number");
      if (u == 0 || (u & (u - 1)) || u > (1u << 20)) return bad(tag, "TLB size must be a power of 2");
      out.tlb_entries = (uint32_t)u;
    }
    else if (tag == "enable_misaligned_data") { if (!get_bool(v, f)) return bad(tag, "expecting a boolean"); if (!f) out.features |= F_NO_MISAL; else out.features &= ~F_NO_MISAL; }
    else if (tag == "clear_mtval_on_illegal_instruction") { if (!get_bool(v, f)) return bad(tag, "expecting a boolean"); if (f) out.features |= F_CLR_MTVAL_ILL; }
    else if (tag == "clear_mtval_on_ebreak") { if (!get_bool(v, f)) return bad(tag, "expecting a boolean"); if (f) out.features |= F_CLR_MTVAL_EBREAK; }
    else if (tag == "cancel_lr_on_trap" || tag == "cancel_lr_on_ret") { if (!get_bool(v, f)) return bad(tag, "expecting a boolean"); if (f) out.features |= F_CANCEL_LR_TRAP; }
    else if (tag == "clear_mprv_on_ret") { if (!get_bool(v, f)) return bad(tag, "expecting a boolean"); if (!f) out.features |= F_KEEP_MPRV; }
    else if (tag == "reservation_bytes") {
      if (!get_unsigned(v, u) || (u & (u - 1)) || u == 0) return bad(tag, "expecting a power of 2");
      if (u > 8) return unsup(tag, "reservation sets larger than 8 bytes are not modelled");
    }
    else if (tag == "physical_memory_protection_grain") { if (!get_unsigned(v, u)) return bad(tag, "expecting a number"); if (u > 4) return unsup(tag, "only grain 4 (G = 0) is modelled"); }
    else if (tag == "time_down_sample") { if (!get_unsigned(v, u)) return bad(tag, "expecting a number"); if (u != 0) return unsup(tag, "only 0 is modelled"); }
    else if (tag == "guest_interrupt_count" || tag == "wfi_timeout") { if (!get_unsigned(v, u)) return bad(tag, "expecting a number"); if (u != 0) return unsup(tag, "only 0 is modelled"); }
    else if (tag == "address_translation_modes") {
      if (v.t != JVal::ARR) return bad(tag, "expecting an array of strings");
      bool sv39 = false;
      for (auto& it : v.arr) {
        if (it.t != JVal::STR) return bad(tag, "expecting an array of strings");
        if (it.s == "sv39") sv39 = true;
        else if (it.s != "bare" && it.s != "sv48" && it.s != "sv57" && it.s != "sv32") return bad(tag, "no such address translation mode");
      }
      if (!sv39) out.features |= F_NO_SV39;          // sv48 / sv57, where listed, stop a hart that selects them (UNSUPPORTED 0x180)
    }
    else if (tag == "memmap") {
      if (v.t != JVal::OBJ) return bad(tag, "expecting an object");
      for (auto& mk : v.obj) {
        if (mk.first == "size") { if (!get_unsigned(mk.second, u) || u == 0) return bad("memmap.size", "expecting a number"); out.mem_size = u; }
        else if (mk.first == "page_size") { if (!get_unsigned(mk.second, u)) return bad("memmap.page_size", "expecting a number"); }
        else return unsup("memmap." + mk.first, "only size and page_size are modelled (one RWX arena per hart)");
      }
    }
    else if (tag == "vector") {
      if (v.t != JVal::OBJ) return bad(tag, "expecting an object");
      const JVal* bpv = v.find("bytes_per_vec");
      if (!bpv) return bad("vector.bytes_per_vec", "missing");                                   // HartConfig.cpp:834-838
      if (!get_unsigned(*bpv, u) || u == 0 || (u & (u - 1))) return bad("vector.bytes_per_vec", "expecting a power of 2");
      if (u != 16 && u != 32 && u != 64) return unsup("vector.bytes_per_vec", "16, 32 and 64 are modelled");
      out.vlen_bytes = (uint32_t)u;
      if (!v.find("max_bytes_per_elem")) return bad("vector.max_bytes_per_elem", "missing");     // HartConfig.cpp:866-872
      for (auto& vk : v.obj) {
        const std::string t = "vector." + vk.first;
        if (vk.first == "bytes_per_vec") continue;
        if (vk.first == "min_bytes_per_elem") { if (!get_unsigned(vk.second, u)) return bad(t, "expecting a number"); if (u != 1) return unsup(t, "only 1 is modelled"); }
        else if (vk.first == "max_bytes_per_elem") { if (!get_unsigned(vk.second, u)) return bad(t, "expecting a number"); if (u != 8) return unsup(t, "only 8 (ELEN 64) is modelled"); }
        else if (vk.first == "mask_agnostic_policy" || vk.first == "tail_agnostic_policy") {
          if (vk.second.t != JVal::STR || (vk.second.s != "ones" && vk.second.s != "undisturb")) return bad(t, "must be 'ones' or 'undisturb'");
          if (vk.second.s != "ones") return unsup(t, "only 'ones' is modelled");
        }
        else if (vk.first == "trap_non_zero_vstart") { if (!get_bool(vk.second, f)) return bad(t, "expecting a boolean"); if (!f) return unsup(t, "only true is modelled"); }
        else if (vk.first == "trap_out_of_bounds_vstart" || vk.first == "update_whole_mask" || vk.first == "trap_invalid_vtype" ||
                 vk.first == "legalize_vsetvl_avl" || vk.first == "legalize_vsetvli_avl" || vk.first == "legalize_for_egs" ||
                 vk.first == "partial_segment_update" || vk.first == "always_mark_dirty" || vk.first == "always_mark_dirty_covers_load" ||
                 vk.first == "vmvr_ignore_vill" || vk.first == "tt_clear_tval_vl_egs") {
          if (!get_bool(vk.second, f)) return bad(t, "expecting a boolean");
          if (f) return unsup(t, "only false (the reference's default) is modelled");
        }
        else if (vk.first == "tt_fp_usum_tree_reduction" || vk.first == "fp_usum_nan_canonicalize") {
          if (vk.second.t != JVal::ARR) return bad(t, "expecting an array"); if (!vk.second.arr.empty()) return unsup(t, "only the sequential sum is modelled");
        }
        else return unsup(t, "not modelled");
      }
    }
    else if (tag == "csr") {
      if (v.t != JVal::OBJ) return bad(tag, "expecting an object");
      for (auto& ck : v.obj) {
        const int num = csr_by_name(ck.first);
        if (ck.second.t != JVal::OBJ) return bad("csr." + ck.first, "expecting an object");
        for (auto& fk : ck.second.obj) {
          const std::string t = "csr." + ck.first + "." + fk.first;
          if (fk.first == "reset") {
            if (!get_unsigned(fk.second, u)) return bad(t, "expecting a number");
            if (num < 0) return unsup(t, "CSR is not in the modelled set");
            if (num == 0x301 || (num >= 0xf11 && num <= 0xf15)) return unsup(t, "read-only identification CSRs keep the reference's default values");
            out.csr.push_back({ (uint32_t)num, u });
          }
          else if (fk.first == "exists") { if (!get_bool(fk.second, f)) return bad(t, "expecting a boolean"); if (f != (num >= 0)) return unsup(t, "the implemented CSR set is fixed"); }
          else return unsup(t, "per-CSR write masks, sharing and new CSR definitions are not modelled");
        }
      }
    }
    else return unsup(tag, "not modelled on the batched path");
  }
  return R_OK;
}

}  // namespace wspcfg
