// Test tool: compares whisper_amd/csrc/wsp_softfloat.h (compiled for the host) against the
// reference's vendored Berkeley SoftFloat (oracle/_ref/libsoftfloat_ref.so) on directed and
// random operands, all five rounding modes, results AND exception flags.
// usage: sf_check <n_random_per_op> ; prints mismatches, exit code = min(#mismatch, 255)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../whisper_amd/csrc/wsp_softfloat.h"
extern "C" {
#include "softfloat.h"
}
using namespace wsf;
static u64 rng_s = 0x1234567;
static u64 rnd() { rng_s += 0x9E3779B97F4A7C15ULL; u64 z = rng_s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static const uint8_t rm_sf[5] = { softfloat_round_near_even, softfloat_round_minMag, softfloat_round_min, softfloat_round_max, softfloat_round_near_maxMag };
static long bad = 0, total = 0;
static void chk(const char* op, int rm, u64 a, u64 b, u64 c, u64 got, unsigned gf, u64 exp, unsigned ef)
{
  ++total;
  if (got != exp || gf != ef) { if (bad < 40) printf("MISMATCH %s rm=%d a=%016lx b=%016lx c=%016lx got=%016lx/%02x exp=%016lx/%02x\n", op, rm, a, b, c, got, gf, exp, ef); ++bad; }
}
static std::vector<u64> specials64() {
  std::vector<u64> v = {0, 1, 2, 0x000fffffffffffffULL, 0x0010000000000000ULL, 0x0010000000000001ULL, 0x001fffffffffffffULL,
    0x3ff0000000000000ULL, 0x3ff0000000000001ULL, 0x3fefffffffffffffULL, 0x4000000000000000ULL, 0x3fe0000000000000ULL, 0x7fefffffffffffffULL, 0x7fe0000000000000ULL,
    0x7ff0000000000000ULL, 0x7ff8000000000000ULL, 0x7ff0000000000001ULL, 0x7ff4000000000000ULL, 0x7fffffffffffffffULL, 0x4340000000000000ULL, 0x433fffffffffffffULL,
    0x43e0000000000000ULL, 0x43dfffffffffffffULL, 0x43f0000000000000ULL, 0x41dfffffffc00000ULL, 0x41e0000000000000ULL, 0x41efffffffe00000ULL, 0x41f0000000000000ULL,
    0x3fdfffffffffffffULL, 0x3fe0000000000001ULL, 0x3ff8000000000000ULL, 0x4004000000000000ULL, 0x36a0000000000000ULL, 0x3690000000000000ULL, 0x380fffffffffffffULL, 0x3810000000000000ULL, 0x47efffffe0000000ULL, 0x47effffff0000000ULL};
  size_t n = v.size(); for (size_t i = 0; i < n; ++i) v.push_back(v[i] | 0x8000000000000000ULL); return v;
}
static std::vector<u64> specials32() {
  std::vector<u64> v = {0, 1, 2, 0x007fffff, 0x00800000, 0x00800001, 0x00ffffff, 0x3f800000, 0x3f800001, 0x3f7fffff, 0x40000000, 0x3f000000, 0x7f7fffff, 0x7f000000,
    0x7f800000, 0x7fc00000, 0x7f800001, 0x7fa00000, 0x7fffffff, 0x4b000000, 0x4affffff, 0x4f000000, 0x4effffff, 0x4f800000, 0x5f000000, 0x5effffff, 0x5f800000, 0x3effffff, 0x3f000001, 0x3fc00000, 0x40200000, 0x33800000, 0x34000000, 0x0c000000};
  size_t n = v.size(); for (size_t i = 0; i < n; ++i) v.push_back(v[i] | 0x80000000ULL); return v;
}
static u64 rand64f() { u64 r = rnd(); switch (rnd() % 8) { case 0: return r; case 1: return (r & 0x800fffffffffffffULL) | ((u64)(1023 + (int)(rnd() % 120) - 60) << 52);
  case 2: return (r & 0x800fffffffffffffULL) | ((u64)(rnd() % 4) << 52); case 3: return (r & 0x800fffffffffffffULL) | ((u64)(2043 + rnd() % 4) << 52);
  case 4: return (r & 0xfff0000000000000ULL) | (rnd() % 2 ? 0 : (1ULL << (rnd() % 52))); case 5: return (r & 0x800fffffffffffffULL) | ((u64)(1023 + (int)(rnd() % 8) - 4) << 52);
  case 6: return (r & 0x800fffffffffffffULL) | ((u64)(1023 - 1080 + rnd() % 70) << 52) ; default: return (r & 0x800fffffffffffffULL) | ((u64)(1023 + 40 + rnd() % 30) << 52); } }
static u64 rand32f() { u64 r = rnd() & 0xffffffffULL; switch (rnd() % 8) { case 0: return r; case 1: return (r & 0x807fffff) | ((u64)(127 + (int)(rnd() % 60) - 30) << 23);
  case 2: return (r & 0x807fffff) | ((u64)(rnd() % 4) << 23); case 3: return (r & 0x807fffff) | ((u64)(251 + rnd() % 4) << 23);
  case 4: return (r & 0xff800000) | (rnd() % 2 ? 0 : (1ULL << (rnd() % 23))); case 5: return (r & 0x807fffff) | ((u64)(127 + (int)(rnd() % 8) - 4) << 23);
  case 6: return (r & 0x807fffff) | ((u64)(1 + rnd() % 40) << 23); default: return (r & 0x807fffff) | ((u64)(127 + 20 + rnd() % 45) << 23); } }

template <class F> struct Ref;
template <> struct Ref<F64> { typedef float64_t T;
  static u64 add(u64 a, u64 b) { return f64_add({a}, {b}).v; } static u64 sub(u64 a, u64 b) { return f64_sub({a}, {b}).v; }
  static u64 mul(u64 a, u64 b) { return f64_mul({a}, {b}).v; } static u64 div(u64 a, u64 b) { return f64_div({a}, {b}).v; }
  static u64 sqrt(u64 a) { return f64_sqrt({a}).v; } static u64 fma(u64 a, u64 b, u64 c) { return f64_mulAdd({a}, {b}, {c}).v; }
  static u64 toi(u64 a, int rm, bool sg, bool w) { float64_t x = {a}; return sg ? (w ? (u64)f64_to_i64(x, rm, true) : (u64)(i64)(i32)f64_to_i32(x, rm, true)) : (w ? (u64)f64_to_ui64(x, rm, true) : (u64)(u32)f64_to_ui32(x, rm, true)); }
  static u64 fromi(u64 v, bool sg, bool w) { return sg ? (w ? i64_to_f64((i64)v).v : i32_to_f64((i32)v).v) : (w ? ui64_to_f64(v).v : ui32_to_f64((u32)v).v); } };
template <> struct Ref<F32> { typedef float32_t T;
  static u64 add(u64 a, u64 b) { return f32_add({(u32)a}, {(u32)b}).v; } static u64 sub(u64 a, u64 b) { return f32_sub({(u32)a}, {(u32)b}).v; }
  static u64 mul(u64 a, u64 b) { return f32_mul({(u32)a}, {(u32)b}).v; } static u64 div(u64 a, u64 b) { return f32_div({(u32)a}, {(u32)b}).v; }
  static u64 sqrt(u64 a) { return f32_sqrt({(u32)a}).v; } static u64 fma(u64 a, u64 b, u64 c) { return f32_mulAdd({(u32)a}, {(u32)b}, {(u32)c}).v; }
  static u64 toi(u64 a, int rm, bool sg, bool w) { float32_t x = {(u32)a}; return sg ? (w ? (u64)f32_to_i64(x, rm, true) : (u64)(i64)(i32)f32_to_i32(x, rm, true)) : (w ? (u64)f32_to_ui64(x, rm, true) : (u64)(u32)f32_to_ui32(x, rm, true)); }
  static u64 fromi(u64 v, bool sg, bool w) { return sg ? (w ? i64_to_f32((i64)v).v : i32_to_f32((i32)v).v) : (w ? ui64_to_f32(v).v : ui32_to_f32((u32)v).v); } };

template <class F> static void one(const char* tag, u64 a, u64 b, u64 c)
{
  char nm[32];
  for (int rm = 0; rm < 5; ++rm) {
    softfloat_roundingMode = rm_sf[rm]; unsigned f; u64 g, e;
#define T2(NAME, MINE, REF) f = 0; g = MINE; softfloat_exceptionFlags = 0; e = REF; snprintf(nm, 32, "%s.%s", tag, NAME); chk(nm, rm, a, b, c, g, f, e, softfloat_exceptionFlags);
    T2("add", f_add<F>(a, b, rm, f), Ref<F>::add(a, b));
    T2("sub", f_sub<F>(a, b, rm, f), Ref<F>::sub(a, b));
    T2("mul", f_mul<F>(a, b, rm, f), Ref<F>::mul(a, b));
    T2("div", f_div<F>(a, b, rm, f), Ref<F>::div(a, b));
    T2("sqrt", f_sqrt<F>(a, rm, f), Ref<F>::sqrt(a));
    T2("fma", f_fma<F>(a, b, c, rm, f), Ref<F>::fma(a, b, c));
    for (int k = 0; k < 4; ++k) { bool sg = k & 1, w = k & 2;
      T2(sg ? (w ? "to_i64" : "to_i32") : (w ? "to_u64" : "to_u32"), f_to_int<F>(a, rm, sg, w, f), Ref<F>::toi(a, rm_sf[rm], sg, w));
      T2(sg ? (w ? "from_i64" : "from_i32") : (w ? "from_u64" : "from_u32"), int_to_f<F>(c, sg, w, rm, f), Ref<F>::fromi(c, sg, w)); }
  }
}
// The FP-unit paths (hw_* in the header: round to nearest even, ordinary operands) against the real library, wherever they apply.
static long hw_hits = 0;
static void one_hw(u64 a, u64 b, u64 c)
{
  softfloat_roundingMode = softfloat_round_near_even; unsigned f; u64 g, e;
#define TH(NAME, CALL, REF) f = 0; g = 0; if (CALL) { ++hw_hits; softfloat_exceptionFlags = 0; e = REF; chk("hw." NAME, 0, a, b, c, g, f, e, softfloat_exceptionFlags); }
  TH("add", hw_add64(a, b, g, f), Ref<F64>::add(a, b));
  TH("sub", hw_add64(a, b ^ F64::SIGN, g, f), Ref<F64>::sub(a, b));
  TH("mul", hw_mul64(a, b, g, f), Ref<F64>::mul(a, b));
  TH("div", hw_div64(a, b, g, f), Ref<F64>::div(a, b));
  TH("fma", hw_fma64(a, b, c, g, f), Ref<F64>::fma(a, b, c));
  for (int k = 0; k < 4; ++k) { bool sg = k & 1, w = k & 2; u64 v = c >> (rnd() % 64); if (rnd() & 1) v = (u64)0 - v;
    f = 0; g = hw_int_to_f64(v, sg, w, f); softfloat_exceptionFlags = 0; e = Ref<F64>::fromi(v, sg, w); chk("hw.fromi", 0, v, 0, 0, g, f, e, softfloat_exceptionFlags); }
}
static void one_hw32(u64 a, u64 b, u64 c)
{
  softfloat_roundingMode = softfloat_round_near_even; unsigned f; u64 g, e;
  TH("add32", hw_add32(a, b, g, f), Ref<F32>::add(a, b));
  TH("sub32", hw_add32(a, b ^ F32::SIGN, g, f), Ref<F32>::sub(a, b));
  TH("mul32", hw_mul32(a, b, g, f), Ref<F32>::mul(a, b));
  TH("div32", hw_div32(a, b, g, f), Ref<F32>::div(a, b));
  TH("fma32", hw_fma32(a, b, c, g, f), Ref<F32>::fma(a, b, c));
}
// operands the way programs have them: moderate exponents, few significant bits, exact quotients and products
static u64 rand64o() { u64 r = rnd(); u64 e = (u64)(1023 + (int)(rnd() % 40) - 20) << 52; u64 m = r & 0xfffffffffffffULL;
  switch (rnd() % 4) { case 0: m &= ~((1ULL << (rnd() % 53)) - 1); break; case 1: m = 0; break; default: break; }
  return (r & 0x8000000000000000ULL) | e | m; }

int main(int argc, char** argv)
{
  long n = argc > 1 ? atol(argv[1]) : 200000;
  auto s64 = specials64(); auto s32 = specials32();
  for (u64 a : s64) for (u64 b : s64) { one<F64>("f64", a, b, s64[rnd() % s64.size()]); }
  for (u64 a : s64) for (u64 b : s64) { one_hw(a, b, s64[rnd() % s64.size()]); }
  for (u64 a : s32) for (u64 b : s32) { one<F32>("f32", a, b, s32[rnd() % s32.size()]); }
  for (u64 a : s32) for (u64 b : s32) for (u64 c : s32) one_hw32(a, b, c);
  for (u64 a : s64) for (u64 b : s64) for (int k = 0; k < 4; ++k) one<F64>("f64", a, b, s64[rnd() % s64.size()]);
  for (long i = 0; i < n; ++i) {
    u64 a = rand64f(), b = rand64f(), c = rand64f();
    if (rnd() % 4 == 0) { // near-cancellation fma: c ~ -(a*b)
      softfloat_roundingMode = 0; c = f64_mul({a}, {b}).v ^ 0x8000000000000000ULL; if (rnd() & 1) c += (rnd() % 5) - 2; }
    if (rnd() % 8 == 0) b = a ^ (rnd() % 2 ? 0x8000000000000000ULL : 0) ^ (rnd() % 4);
    one<F64>("f64", a, b, c);
    one_hw(a, b, c);
    { u64 p = rand64o(), q = rand64o(), w = rand64o(); one_hw(p, q, w);
      softfloat_roundingMode = 0; one_hw(f64_mul({p}, {q}).v, q, f64_mul({p}, {q}).v ^ 0x8000000000000000ULL); }   // often an exact quotient; fma cancels
    u64 x = rand32f(), y = rand32f(), z = rand32f();
    if (rnd() % 4 == 0) { softfloat_roundingMode = 0; z = f32_mul({(u32)x}, {(u32)y}).v ^ 0x80000000u; if (rnd() & 1) z = (u32)(z + (rnd() % 5) - 2); }
    if (rnd() % 8 == 0) y = (x ^ (rnd() % 2 ? 0x80000000u : 0) ^ (rnd() % 4)) & 0xffffffffu;
    one<F32>("f32", x, y, z);
    one_hw32(x, y, z);
    { // few significant bits, moderate exponents: exact sums / products / quotients, fma ties
      auto ro = [] { u64 r = rnd(); u64 m = r & 0x7fffff; switch (rnd() % 4) { case 0: m &= ~((1ULL << (rnd() % 24)) - 1); break; case 1: m = 0; break; default: break; }
                     return ((r >> 32) & 0x80000000ULL) | ((u64)(127 + (int)(rnd() % 40) - 20) << 23) | m; };
      u64 p = ro(), q = ro(), w = ro(); one_hw32(p, q, w);
      softfloat_roundingMode = 0; const u64 pq = f32_mul({(u32)p}, {(u32)q}).v; one_hw32(pq, q, pq ^ 0x80000000u);
      one_hw32(p, q, (u64)(u32)((pq ^ 0x80000000u) + (rnd() % 5) - 2)); }
    // conversions between formats
    for (int rm = 0; rm < 5; ++rm) { softfloat_roundingMode = rm_sf[rm]; unsigned f = 0; u64 g = f64_to_f32(a, rm, f); softfloat_exceptionFlags = 0; u64 e = f64_to_f32({a}).v; chk("f64_to_f32", rm, a, 0, 0, g, f, e, softfloat_exceptionFlags); }
    { unsigned f = 0; u64 g = f32_to_f64(x, f); softfloat_exceptionFlags = 0; u64 e = ::f32_to_f64({(u32)x}).v; chk("f32_to_f64", 0, x, 0, 0, g, f, e, softfloat_exceptionFlags); }
  }
  for (u64 a : s64) for (int rm = 0; rm < 5; ++rm) { softfloat_roundingMode = rm_sf[rm]; unsigned f = 0; u64 g = f64_to_f32(a, rm, f); softfloat_exceptionFlags = 0; u64 e = f64_to_f32({a}).v; chk("f64_to_f32", rm, a, 0, 0, g, f, e, softfloat_exceptionFlags); }
  for (u64 x : s32) { unsigned f = 0; u64 g = f32_to_f64(x, f); softfloat_exceptionFlags = 0; u64 e = ::f32_to_f64({(u32)x}).v; chk("f32_to_f64", 0, x, 0, 0, g, f, e, softfloat_exceptionFlags); }
  printf("checked %ld cases (%ld on the FP-unit paths), %ld mismatches\n", total, hw_hits, bad);
  return bad > 255 ? 255 : (int)bad;
}
