// Bit-exact IEEE-754 binary32/binary64 arithmetic in integer registers, for one hart per lane.
//
// Stands in for the reference's vendored Berkeley SoftFloat 3e as configured for RISC-V
// (/root/reference/third_party/softfloat/source/RISCV/specialize.h: canonical NaN on every NaN
// result, invalid on signaling NaNs, tininess detected AFTER rounding, RISC-V integer saturation
// values), which the reference reaches through float-util.hpp:230-930.  Written from the IEEE
// behaviour, not from the library's code: every operation funnels an exact (sign, exponent,
// 128-bit significand + sticky) value through one rounding routine.  oracle/sf_check.cpp (host build)
// and tests/test_gpu_softfloat.py (the gfx950 build, through wsp_sf_eval) compare each entry point against the real library
// (oracle/_ref) on directed + random operands.
//
// All functions are WSP_HD so the same source is compiled into the HIP kernels and into the
// host-side unit-test shim; there is no separate CPU implementation.
#pragma once
#include <stdint.h>

#ifndef WSP_HD
#  if defined(__HIPCC__)
#    define WSP_HD __host__ __device__ __forceinline__
#    define WSP_HDN __host__ __device__ __attribute__((noinline))     /* large, rarely taken paths: kept out of their callers' register budget */
#  else
#    define WSP_HD inline
#    define WSP_HDN inline
#  endif
#endif

namespace wsf {

typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t  i64;
typedef int32_t  i32;

enum { FL_NX = 1, FL_UF = 2, FL_OF = 4, FL_DZ = 8, FL_NV = 16 };
enum { RM_RNE = 0, RM_RTZ = 1, RM_RDN = 2, RM_RUP = 3, RM_RMM = 4, RM_ODD = 6 };   // RM_ODD: SoftFloat's round_odd (jam), used by vfncvt.rod only

struct U128 { u64 hi, lo; };

WSP_HD int clz64(u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return x ? __clzll((long long)x) : 64;
#else
  return x ? __builtin_clzll(x) : 64;
#endif
}

WSP_HD U128 mul64x64(u64 a, u64 b)
{
  U128 r;
#if defined(__HIP_DEVICE_COMPILE__)
  r.hi = __umul64hi(a, b);
  r.lo = a * b;
#else
  unsigned __int128 p = (unsigned __int128)a * b;
  r.hi = (u64)(p >> 64);
  r.lo = (u64)p;
#endif
  return r;
}

WSP_HD U128 shl128(U128 a, unsigned n)   // n in [0,127]
{
  U128 r;
  if (n == 0) return a;
  if (n >= 64) { r.hi = a.lo << (n - 64); r.lo = 0; }
  else { r.hi = (a.hi << n) | (a.lo >> (64 - n)); r.lo = a.lo << n; }
  return r;
}

WSP_HD U128 shr128_jam(U128 a, unsigned n)   // any n; bits shifted out are OR-ed into bit 0
{
  U128 r;
  if (n == 0) return a;
  if (n >= 128) { r.hi = 0; r.lo = (a.hi | a.lo) != 0; return r; }
  if (n >= 64) {
    unsigned m = n - 64;
    u64 lost = a.lo | (m ? (a.hi << (64 - m)) : 0);
    r.hi = 0; r.lo = (m ? (a.hi >> m) : a.hi) | (lost != 0);
    return r;
  }
  u64 lost = a.lo << (64 - n);
  r.hi = a.hi >> n;
  r.lo = (a.hi << (64 - n)) | (a.lo >> n) | (lost != 0);
  return r;
}

WSP_HD U128 add128(U128 a, U128 b) { U128 r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi + (r.lo < a.lo); return r; }
WSP_HD U128 sub128(U128 a, U128 b) { U128 r; r.lo = a.lo - b.lo; r.hi = a.hi - b.hi - (a.lo < b.lo); return r; }
WSP_HD bool lt128(U128 a, U128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
WSP_HD bool eq128(U128 a, U128 b) { return a.hi == b.hi && a.lo == b.lo; }

// floor((hi:lo) / d) for hi < d and d normalised (bit 63 set); remainder in *rem.
WSP_HD u64 div128by64(u64 hi, u64 lo, u64 d, u64* rem)
{
#if !defined(__HIP_DEVICE_COMPILE__)
  unsigned __int128 n = ((unsigned __int128)hi << 64) | lo;
  *rem = (u64)(n % d);
  return (u64)(n / d);
#else
  // schoolbook base-2^32 long division with two quotient digits (d has its top bit set)
  const u64 b = 1ULL << 32;
  u64 dh = d >> 32, dl = d & 0xffffffffULL;
  u64 l1 = lo >> 32, l0 = lo & 0xffffffffULL;
  u64 q1 = hi / dh, r = hi - q1 * dh;
  while (q1 >= b || q1 * dl > ((r << 32) | l1)) { --q1; r += dh; if (r >= b) break; }
  u64 un = ((hi << 32) | l1) - q1 * d;
  u64 q0 = un / dh; r = un - q0 * dh;
  while (q0 >= b || q0 * dl > ((r << 32) | l0)) { --q0; r += dh; if (r >= b) break; }
  *rem = ((un << 32) | l0) - q0 * d;
  return (q1 << 32) | q0;
#endif
}

// ------------------------------------------------------------------------------------------
template <int EB_, int MB_> struct Fmt {
  static constexpr int EB = EB_, MB = MB_;
  static constexpr int BIAS = (1 << (EB_ - 1)) - 1;
  static constexpr int EMAX = (1 << EB_) - 1;             // all-ones exponent field
  static constexpr u64 MANT = (1ULL << MB_) - 1;
  static constexpr u64 SIGN = 1ULL << (EB_ + MB_);
  static constexpr u64 QNAN = ((u64)EMAX << MB_) | (1ULL << (MB_ - 1));   // canonical quiet NaN
  static constexpr u64 INF = (u64)EMAX << MB_;
};
typedef Fmt<8, 23> F32;
typedef Fmt<11, 52> F64;

enum { K_ZERO = 0, K_FIN = 1, K_INF = 2, K_QNAN = 3, K_SNAN = 4 };
struct Unp { int kind; bool sign; int exp; u64 sig; };   // finite: value = sig * 2^(exp-63), sig bit 63 set

template <class F> WSP_HD Unp unpack(u64 a)
{
  Unp u;
  u.sign = (a & F::SIGN) != 0;
  int e = (int)((a >> F::MB) & (u64)F::EMAX);
  u64 m = a & F::MANT;
  u.exp = 0; u.sig = 0;
  if (e == F::EMAX) { u.kind = m ? ((m >> (F::MB - 1)) ? K_QNAN : K_SNAN) : K_INF; return u; }
  if (e == 0) {
    if (!m) { u.kind = K_ZERO; return u; }
    int lz = clz64(m);
    u.kind = K_FIN; u.sig = m << lz;
    u.exp = 1 - F::BIAS - F::MB + (63 - lz);     // subnormal: value = m * 2^(1-bias-MB)
    return u;
  }
  u.kind = K_FIN; u.sig = ((1ULL << F::MB) | m) << (63 - F::MB); u.exp = e - F::BIAS;
  return u;
}

// Round the exact value (-1)^sign * (sig.hi:sig.lo) * 2^(exp-127) (sig != 0) to format F.
template <class F> WSP_HD u64 round_pack(bool sign, int exp, U128 sig, int rm, unsigned& fl)
{
  // normalise so the leading one sits at bit 126 (bit 62 of the high word)
  int lz = sig.hi ? clz64(sig.hi) : 64 + clz64(sig.lo);
  if (lz == 0) { sig = shr128_jam(sig, 1); exp += 1; }
  else if (lz > 1) { sig = shl128(sig, (unsigned)(lz - 1)); exp -= lz - 1; }
  exp -= 1;                                 // value = 1.f * 2^exp with the one at bit 126
  u64 s = sig.hi | (sig.lo != 0);           // 63-bit significand with sticky in bit 0
  const int RB = 62 - F::MB;                // number of bits to round away
  const u64 rmask = (1ULL << RB) - 1, half = 1ULL << (RB - 1);
  u64 inc = (rm == RM_RNE || rm == RM_RMM) ? half
          : (rm == RM_RTZ || rm == RM_ODD) ? 0
          : ((rm == RM_RDN) == sign) ? rmask : 0;   // RDN rounds negative magnitudes up, RUP positive
  i64 e = (i64)exp + F::BIAS;               // biased exponent of the unrounded value
  if (e <= 0) {
    // tininess after rounding: tiny unless rounding at full precision reaches the smallest normal
    bool tiny = e < 0 || (s + inc) < 0x8000000000000000ULL;
    u64 sh = (u64)(1 - e);
    s = sh >= 63 ? (u64)(s != 0) : ((s >> sh) | ((s << (64 - sh)) != 0));
    if (tiny && (s & rmask)) fl |= FL_UF;
    e = 0;
  } else if (e >= F::EMAX || (e == F::EMAX - 1 && (s + inc) >= 0x8000000000000000ULL)) {
    fl |= FL_OF | FL_NX;
    bool to_inf = inc == half || inc == rmask;
    u64 mag = to_inf ? F::INF : (F::INF - 1);
    return (sign ? F::SIGN : 0) | mag;
  }
  u64 rb = s & rmask;
  if (rb) fl |= FL_NX;
  u64 r = (s + inc) >> RB;
  if (rm == RM_RNE && rb == half) r &= ~1ULL;
  if (rm == RM_ODD && rb) r |= 1;           // round to odd: an inexact result gets its last bit set
  // r holds the hidden bit at position MB for normals; adding it to the exponent field below
  // accounts for it (and for a carry out of the significand).
  u64 ef = e > 0 ? (u64)(e - 1) : 0;
  if (e == 0) return (sign ? F::SIGN : 0) | r;
  return (sign ? F::SIGN : 0) | ((ef << F::MB) + r);
}

template <class F> WSP_HD u64 nan_result(const Unp& a, const Unp& b, unsigned& fl)
{ if (a.kind == K_SNAN || b.kind == K_SNAN) fl |= FL_NV; return F::QNAN; }

// exact sum of two finite non-zero magnitudes with signs; result rounded to F.
template <class F> WSP_HD u64 add_fin(bool sa, int ea, U128 xa, bool sb, int eb, U128 xb, int rm, unsigned& fl)
{
  // operands have their leading one at bit 126 of the 128-bit frame
  if (ea < eb || (ea == eb && lt128(xa, xb))) { bool ts = sa; sa = sb; sb = ts; int te = ea; ea = eb; eb = te; U128 tx = xa; xa = xb; xb = tx; }
  xb = shr128_jam(xb, (unsigned)(ea - eb > 200 ? 200 : ea - eb));
  U128 r;
  if (sa == sb) r = add128(xa, xb);
  else {
    r = sub128(xa, xb);
    if (!(r.hi | r.lo)) return rm == RM_RDN ? F::SIGN : 0;     // exact cancellation
  }
  return round_pack<F>(sa, ea + 1, r, rm, fl);
}

WSP_HD U128 frame(u64 sig) { U128 r; r.hi = sig >> 1; r.lo = sig << 63; return r; }

template <class F> WSP_HD u64 f_add(u64 a, u64 b, int rm, unsigned& fl, bool negate_b = false)
{
  Unp x = unpack<F>(a), y = unpack<F>(b);
  if (x.kind >= K_QNAN || y.kind >= K_QNAN) return nan_result<F>(x, y, fl);
  if (negate_b) { y.sign = !y.sign; b ^= F::SIGN; }
  if (x.kind == K_INF || y.kind == K_INF) {
    if (x.kind == K_INF && y.kind == K_INF && x.sign != y.sign) { fl |= FL_NV; return F::QNAN; }
    return x.kind == K_INF ? a : b;
  }
  if (x.kind == K_ZERO && y.kind == K_ZERO) return (x.sign == y.sign) ? a : (rm == RM_RDN ? F::SIGN : 0);
  if (x.kind == K_ZERO) return b;
  if (y.kind == K_ZERO) return a;
  return add_fin<F>(x.sign, x.exp, frame(x.sig), y.sign, y.exp, frame(y.sig), rm, fl);
}

template <class F> WSP_HD u64 f_sub(u64 a, u64 b, int rm, unsigned& fl) { return f_add<F>(a, b, rm, fl, true); }

template <class F> WSP_HD u64 f_mul(u64 a, u64 b, int rm, unsigned& fl)
{
  Unp x = unpack<F>(a), y = unpack<F>(b);
  if (x.kind >= K_QNAN || y.kind >= K_QNAN) return nan_result<F>(x, y, fl);
  bool s = x.sign != y.sign;
  if (x.kind == K_INF || y.kind == K_INF) {
    if (x.kind == K_ZERO || y.kind == K_ZERO) { fl |= FL_NV; return F::QNAN; }
    return (s ? F::SIGN : 0) | F::INF;
  }
  if (x.kind == K_ZERO || y.kind == K_ZERO) return s ? F::SIGN : 0;
  return round_pack<F>(s, x.exp + y.exp + 1, mul64x64(x.sig, y.sig), rm, fl);
}

// a*b + c with a single rounding
template <class F> WSP_HD u64 f_fma(u64 a, u64 b, u64 c, int rm, unsigned& fl)
{
  Unp x = unpack<F>(a), y = unpack<F>(b), z = unpack<F>(c);
  bool sp = x.sign != y.sign;
  if (x.kind >= K_QNAN || y.kind >= K_QNAN || z.kind >= K_QNAN) {
    if (x.kind == K_SNAN || y.kind == K_SNAN || z.kind == K_SNAN) fl |= FL_NV;
    // inf * 0 is invalid even when the addend is a quiet NaN
    if ((x.kind == K_INF && y.kind == K_ZERO) || (x.kind == K_ZERO && y.kind == K_INF)) fl |= FL_NV;
    return F::QNAN;
  }
  if (x.kind == K_INF || y.kind == K_INF) {
    if (x.kind == K_ZERO || y.kind == K_ZERO) { fl |= FL_NV; return F::QNAN; }
    if (z.kind == K_INF && z.sign != sp) { fl |= FL_NV; return F::QNAN; }
    return (sp ? F::SIGN : 0) | F::INF;
  }
  if (z.kind == K_INF) return c;
  if (x.kind == K_ZERO || y.kind == K_ZERO) {
    if (z.kind == K_ZERO) return (sp == z.sign) ? c : (rm == RM_RDN ? F::SIGN : 0);
    return c;
  }
  U128 p = mul64x64(x.sig, y.sig);          // leading one at bit 127 or 126
  int ep = x.exp + y.exp;
  if (p.hi >> 63) { p = shr128_jam(p, 1); ep += 1; }   // exact: at most 106 significant bits... low bit is 0
  if (z.kind == K_ZERO) return round_pack<F>(sp, ep + 1, p, rm, fl);
  return add_fin<F>(sp, ep, p, z.sign, z.exp, frame(z.sig), rm, fl);
}

template <class F> WSP_HD u64 f_div(u64 a, u64 b, int rm, unsigned& fl)
{
  Unp x = unpack<F>(a), y = unpack<F>(b);
  if (x.kind >= K_QNAN || y.kind >= K_QNAN) return nan_result<F>(x, y, fl);
  bool s = x.sign != y.sign;
  if (x.kind == K_INF) { if (y.kind == K_INF) { fl |= FL_NV; return F::QNAN; } return (s ? F::SIGN : 0) | F::INF; }
  if (y.kind == K_INF) return s ? F::SIGN : 0;
  if (y.kind == K_ZERO) {
    if (x.kind == K_ZERO) { fl |= FL_NV; return F::QNAN; }
    fl |= FL_DZ; return (s ? F::SIGN : 0) | F::INF;
  }
  if (x.kind == K_ZERO) return s ? F::SIGN : 0;
  u64 rem, q; int e = x.exp - y.exp;
  if (x.sig < y.sig) { q = div128by64(x.sig, 0, y.sig, &rem); e -= 1; }
  else q = div128by64(x.sig >> 1, x.sig << 63, y.sig, &rem);
  U128 sg; sg.hi = q; sg.lo = rem != 0;     // value = q * 2^(e-63) = (q*2^64) * 2^(e-127)
  return round_pack<F>(s, e, sg, rm, fl);
}

// floor(sqrt(n)) for a 128-bit n >= 2^126, with *inexact set when n is not a perfect square
WSP_HD u64 isqrt128(U128 n, bool* inexact)
{
  U128 rem = n, res = { 0, 0 };
  U128 bit = { 1ULL << 62, 0 };             // highest power of four <= 2^126
  for (int i = 0; i < 64; ++i) {
    U128 t = add128(res, bit);
    res = shr128_jam(res, 1);                // low bit of res is always 0 here: no jamming occurs
    if (!lt128(rem, t)) { rem = sub128(rem, t); res = add128(res, bit); }
    bit.lo = (bit.lo >> 2) | (bit.hi << 62); bit.hi >>= 2;
  }
  *inexact = (rem.hi | rem.lo) != 0;
  return res.lo;
}

template <class F> WSP_HD u64 f_sqrt(u64 a, int rm, unsigned& fl)
{
  Unp x = unpack<F>(a);
  if (x.kind >= K_QNAN) { if (x.kind == K_SNAN) fl |= FL_NV; return F::QNAN; }
  if (x.kind == K_ZERO) return a;
  if (x.sign) { fl |= FL_NV; return F::QNAN; }
  if (x.kind == K_INF) return a;
  U128 n; int e;
  if (x.exp & 1) { n.hi = x.sig; n.lo = 0; e = (x.exp - 1) / 2; }           // sig*2^64, in [2^127,2^128)
  else { n.hi = x.sig >> 1; n.lo = x.sig << 63; e = x.exp / 2; }            // sig*2^63, in [2^126,2^127)
  bool inex; u64 r = isqrt128(n, &inex);                                     // r in [2^63, 2^64)
  U128 sg; sg.hi = r; sg.lo = inex;
  return round_pack<F>(false, e, sg, rm, fl);
}

// ---- conversions -------------------------------------------------------------------------
// Round |value| to an integer: returns magnitude in *mag, false if it does not fit 64 bits.
WSP_HD bool round_to_mag(const Unp& x, int rm, u64* mag, bool* inexact)
{
  if (x.exp > 63) return false;
  int sh = 63 - x.exp;                      // >= 0
  u64 ip, frac_hi; bool frac_nz, ge_half, gt_half;
  if (sh == 0) { ip = x.sig; frac_nz = ge_half = gt_half = false; }
  else if (sh < 64) {
    ip = x.sig >> sh; u64 rem = x.sig << (64 - sh);     // fraction left-aligned
    frac_nz = rem != 0; ge_half = (rem >> 63) != 0; gt_half = ge_half && (rem << 1) != 0;
  } else if (sh == 64) { ip = 0; frac_nz = true; ge_half = true; gt_half = (x.sig << 1) != 0; }
  else { ip = 0; frac_nz = true; ge_half = gt_half = false; }
  (void)frac_hi;
  bool up = rm == RM_RNE ? (gt_half || (ge_half && (ip & 1)))
          : rm == RM_RMM ? ge_half
          : rm == RM_RTZ ? false
          : rm == RM_RDN ? (x.sign && frac_nz) : (!x.sign && frac_nz);
  if (up) { ip += 1; if (ip == 0) return false; }
  *mag = ip; *inexact = frac_nz;
  return true;
}

template <class F> WSP_HD u64 f_to_int(u64 a, int rm, bool is_signed, bool is64, unsigned& fl)
{
  Unp x = unpack<F>(a);
  const u64 pos_max = is_signed ? (is64 ? 0x7fffffffffffffffULL : 0x7fffffffULL) : (is64 ? ~0ULL : 0xffffffffULL);
  const u64 neg_ovf = is_signed ? (is64 ? 0x8000000000000000ULL : 0xffffffff80000000ULL) : 0;
  if (x.kind >= K_QNAN) { fl |= FL_NV; return pos_max; }
  if (x.kind == K_INF) { fl |= FL_NV; return x.sign ? neg_ovf : pos_max; }
  if (x.kind == K_ZERO) return 0;
  u64 mag; bool inex;
  if (!round_to_mag(x, rm, &mag, &inex)) { fl |= FL_NV; return x.sign ? neg_ovf : pos_max; }
  if (x.sign) {
    if (is_signed) {
      u64 lim = is64 ? 0x8000000000000000ULL : 0x80000000ULL;
      if (mag > lim) { fl |= FL_NV; return neg_ovf; }
      if (inex) fl |= FL_NX;
      return (u64)(-(i64)mag);
    }
    if (mag != 0) { fl |= FL_NV; return neg_ovf; }
    if (inex) fl |= FL_NX;
    return 0;
  }
  if (mag > pos_max) { fl |= FL_NV; return pos_max; }
  if (inex) fl |= FL_NX;
  return mag;
}

template <class F> WSP_HD u64 int_to_f(u64 v, bool is_signed, bool is64, int rm, unsigned& fl)
{
  bool sign = false; u64 mag;
  if (is64) { if (is_signed && (i64)v < 0) { sign = true; mag = (u64)(-(i64)v); } else mag = v; }
  else { if (is_signed && (i32)v < 0) { sign = true; mag = (u64)(-(i64)(i32)v); } else mag = (u32)v; }
  if (!mag) return 0;
  U128 sg; sg.hi = mag; sg.lo = 0;          // mag * 2^64 * 2^(63-127)
  return round_pack<F>(sign, 63, sg, rm, fl);
}

WSP_HD u64 f32_to_f64(u64 a, unsigned& fl)
{
  Unp x = unpack<F32>(a);
  if (x.kind >= K_QNAN) { if (x.kind == K_SNAN) fl |= FL_NV; return F64::QNAN; }
  u64 s = x.sign ? F64::SIGN : 0;
  if (x.kind == K_INF) return s | F64::INF;
  if (x.kind == K_ZERO) return s;
  return s | ((u64)(x.exp + F64::BIAS) << 52) | ((x.sig << 1) >> 12);
}

WSP_HD u64 f64_to_f32(u64 a, int rm, unsigned& fl)
{
  Unp x = unpack<F64>(a);
  if (x.kind >= K_QNAN) { if (x.kind == K_SNAN) fl |= FL_NV; return F32::QNAN; }
  u64 s = x.sign ? F32::SIGN : 0;
  if (x.kind == K_INF) return s | F32::INF;
  if (x.kind == K_ZERO) return s;
  U128 sg; sg.hi = x.sig; sg.lo = 0;
  return round_pack<F32>(x.sign, x.exp, sg, rm, fl);
}

// ---- non-arithmetic helpers (the reference uses host float ops here: float.cpp:499-829) ----
template <class F> WSP_HD bool is_nan(u64 a) { return (a & F::INF) == F::INF && (a & F::MANT); }
template <class F> WSP_HD bool is_snan(u64 a) { return is_nan<F>(a) && !((a >> (F::MB - 1)) & 1); }
template <class F> WSP_HD bool lt_ord(u64 a, u64 b)   // a < b for non-NaN operands
{
  const u64 mag = F::SIGN - 1;
  bool sa = (a & F::SIGN) != 0, sb = (b & F::SIGN) != 0;
  if (!((a | b) & mag)) return false;
  if (sa != sb) return sa;
  return sa ? ((a & mag) > (b & mag)) : ((a & mag) < (b & mag));
}
template <class F> WSP_HD bool eq_ord(u64 a, u64 b) { const u64 mag = F::SIGN - 1; return a == b || !((a | b) & mag); }
template <class F> WSP_HD unsigned fclass(u64 a)
{
  bool s = (a & F::SIGN) != 0; int e = (int)((a >> F::MB) & (u64)F::EMAX); u64 m = a & F::MANT;
  if (e == F::EMAX) return m ? (((m >> (F::MB - 1)) & 1) ? 0x200u : 0x100u) : (s ? 1u : 0x80u);
  if (e == 0) return m ? (s ? 4u : 0x20u) : (s ? 8u : 0x10u);
  return s ? 2u : 0x40u;
}

// ---- binary64 on the machine's own FP unit, with the exact flags ----------------------------------------------------------
// add / sub / mul / fma / div / int -> double for round-to-nearest-even when every operand is zero or a normal number and the
// exact result stays well inside the normal range: there the machine's correctly rounded IEEE result IS the SoftFloat result,
// the only flag that can be raised is inexact, and inexact is decided by an error-free transformation (2Sum for sums, the fma
// residual for products and quotients, Boldo & Muller's ErrFma for fused multiply-add: each yields the exact rounding error as
// floating-point numbers, which is non-zero exactly when the operation was inexact).  Every function returns false without
// touching its outputs when the operands are outside that domain (NaN, infinity, subnormal, results near overflow or
// underflow, a zero divisor): the caller then runs the integer routine above, which handles everything.  Contraction of a
// product into a following sum must stay off in here (the device compiler's default fuses u1 + a1 below into an fma and the
// transformation is no longer error-free: measured, spurious inexact flags): WSP_FP_STRICT at the top of each body.
// Checked against the integer routines on directed and random operands: oracle/sf_check.cpp (host), tests/test_gpu_softfloat.py
// (gfx950, through wsp_sf_eval).
#if defined(__clang__)
#  define WSP_FP_STRICT _Pragma("clang fp contract(off)")
#else
#  define WSP_FP_STRICT
#  if defined(__FMA__)
#    pragma GCC optimize("fp-contract=off")
#  endif
#endif
WSP_HD double hw_f64(u64 v) { double d; __builtin_memcpy(&d, &v, 8); return d; }
WSP_HD u64 hw_bits(double d) { u64 v; __builtin_memcpy(&v, &d, 8); return v; }
WSP_HD bool hw_zero(u64 a) { return !(a << 1); }
WSP_HD int hw_exp(u64 a) { return (int)((a >> 52) & 0x7ff); }
WSP_HD bool hw_normal(u64 a) { const int e = hw_exp(a); return e >= 1 && e <= 0x7fe; }
enum { HW_ELO = 0x080, HW_EHI = 0x77e };                 // biased exponents of the "well inside" range: 2^-895 .. 2^+895

WSP_HD bool hw_add64(u64 a, u64 b, u64& z, unsigned& fl)
{
  WSP_FP_STRICT
  if (!((hw_zero(a) || hw_normal(a)) && (hw_zero(b) || hw_normal(b)))) return false;
  const double x = hw_f64(a), y = hw_f64(b), s = x + y;
  const u64 r = hw_bits(s);
  if (hw_exp(r) == 0x7ff) return false;                  // overflow
  const double bv = s - x, av = s - bv, e = (x - av) + (y - bv);      // 2Sum: x + y = s + e exactly
  z = r; if (e != 0.0) fl |= FL_NX;
  return true;
}

// the product a * b has its exponent well inside the range (or is an exact zero)
WSP_HD bool hw_prod_ok(u64 a, u64 b)
{
  const bool za = hw_zero(a), zb = hw_zero(b);
  if (!((za || hw_normal(a)) && (zb || hw_normal(b)))) return false;
  const int e = hw_exp(a) + hw_exp(b) - 1023;
  return za || zb || (e >= HW_ELO && e <= HW_EHI);
}

WSP_HD bool hw_mul64(u64 a, u64 b, u64& z, unsigned& fl)
{
  WSP_FP_STRICT
  if (!hw_prod_ok(a, b)) return false;
  const double x = hw_f64(a), y = hw_f64(b), p = x * y;
  z = hw_bits(p); if (__builtin_fma(x, y, -p) != 0.0) fl |= FL_NX;
  return true;
}

WSP_HD bool hw_fma64(u64 a, u64 b, u64 c, u64& z, unsigned& fl)      // a * b + c
{
  WSP_FP_STRICT
  if (!hw_prod_ok(a, b) || !(hw_zero(c) || (hw_exp(c) >= HW_ELO && hw_exp(c) <= HW_EHI))) return false;
  const double x = hw_f64(a), y = hw_f64(b), w = hw_f64(c);
  const double r1 = __builtin_fma(x, y, w);
  const double u1 = x * y, u2 = __builtin_fma(x, y, -u1);             // x * y = u1 + u2
  const double a1 = w + u2, t1 = a1 - w, a2 = (w - (a1 - t1)) + (u2 - t1);      // w + u2 = a1 + a2
  const double b1 = u1 + a1, t2 = b1 - u1, b2 = (u1 - (b1 - t2)) + (a1 - t2);   // u1 + a1 = b1 + b2
  const double g = (b1 - r1) + b2;                                    // x * y + w = r1 + g + a2, both terms exact
  z = hw_bits(r1); if (g + a2 != 0.0) fl |= FL_NX;
  return true;
}

WSP_HD bool hw_div64(u64 a, u64 b, u64& z, unsigned& fl)
{
  WSP_FP_STRICT
  if (!hw_normal(b) || !(hw_zero(a) || hw_normal(a))) return false;
  if (!hw_zero(a)) {
    const int e = hw_exp(a) - hw_exp(b) + 1023;
    if (e < HW_ELO || e > HW_EHI || hw_exp(a) < HW_ELO) return false; // the residual a - q * b must be a normal number or zero
  }
  const double x = hw_f64(a), y = hw_f64(b), q = x / y;
  z = hw_bits(q); if (__builtin_fma(-q, y, x) != 0.0) fl |= FL_NX;
  return true;
}

WSP_HD u64 hw_int_to_f64(u64 v, bool is_signed, bool is64, unsigned& fl)         // always in the domain
{
  if (!is64) return hw_bits(is_signed ? (double)(i32)v : (double)(u32)v);       // 32 bits fit the significand
  const u64 mag = (is_signed && (i64)v < 0) ? (u64)0 - v : v;
  if (mag && ((mag << clz64(mag)) << 53)) fl |= FL_NX;                          // more than 53 significant bits
  return hw_bits(is_signed ? (double)(i64)v : (double)v);
}

// ---- binary32 on the FP unit.  Operands and results ordinary binary32 numbers (zero or normal, result away from overflow and
// underflow).  add / sub / mul / div are computed in binary64 and rounded once more to binary32: with 53 >= 2 * 24 + 2 significand
// bits the second rounding is innocuous for these four operations, so the result is the correctly rounded binary32 one; inexact is
// decided exactly (2Sum in binary64 for sums, the exact 48-bit product, the fma residual for quotients).  fma is not covered by that
// theorem: the result is the unit's own single-rounding binary32 fma, inexact from the exact product plus a 2Sum in binary64.
WSP_HD float hw_f32(u64 v) { const u32 w = (u32)v; float f; __builtin_memcpy(&f, &w, 4); return f; }
WSP_HD u32 hw_bits32(float f) { u32 v; __builtin_memcpy(&v, &f, 4); return v; }
WSP_HD bool hw_zero32(u64 a) { return !((u32)a << 1); }
WSP_HD int hw_exp32(u64 a) { return (int)((a >> 23) & 0xff); }
WSP_HD bool hw_ord32(u64 a) { const int e = hw_exp32(a); return hw_zero32(a) || (e >= 1 && e <= 0xfe); }
// the binary32 result r of an operation whose exact value is v (a binary64 value standing for it): in range when it is an exact
// zero or a normal number above the smallest binade (a value just under 2^-126 that rounds up to it is still tiny: underflow); a
// zero or subnormal result of a non-zero value underflowed
WSP_HD bool hw_res32_ok(u32 r, bool exact_zero) { const int e = (int)((r >> 23) & 0xff); return e == 0xff ? false : e == 0 ? (exact_zero && !(r << 1)) : e >= 2; }

WSP_HD bool hw_add32(u64 a, u64 b, u64& z, unsigned& fl)
{
  WSP_FP_STRICT
  if (!(hw_ord32(a) && hw_ord32(b))) return false;
  const double x = (double)hw_f32(a), y = (double)hw_f32(b), s = x + y;
  const double bv = s - x, av = s - bv, e = (x - av) + (y - bv);        // x + y = s + e exactly
  const float r = (float)s;
  const u32 rb = hw_bits32(r);
  if (!hw_res32_ok(rb, s == 0.0 && e == 0.0)) return false;
  z = rb; if ((double)r != s || e != 0.0) fl |= FL_NX;
  return true;
}

WSP_HD bool hw_mul32(u64 a, u64 b, u64& z, unsigned& fl)
{
  WSP_FP_STRICT
  if (!(hw_ord32(a) && hw_ord32(b))) return false;
  const double p = (double)hw_f32(a) * (double)hw_f32(b);               // 48 significant bits: exact
  const float r = (float)p;
  const u32 rb = hw_bits32(r);
  if (!hw_res32_ok(rb, p == 0.0)) return false;
  z = rb; if ((double)r != p) fl |= FL_NX;
  return true;
}

WSP_HD bool hw_div32(u64 a, u64 b, u64& z, unsigned& fl)
{
  WSP_FP_STRICT
  if (!(hw_ord32(a) && hw_ord32(b)) || hw_zero32(b)) return false;
  const double x = (double)hw_f32(a), y = (double)hw_f32(b), q = x / y;
  const float r = (float)q;
  const u32 rb = hw_bits32(r);
  if (!hw_res32_ok(rb, x == 0.0)) return false;
  z = rb; if (__builtin_fma(-(double)r, y, x) != 0.0) fl |= FL_NX;      // r * y has 48 significant bits: the residual is exact
  return true;
}

WSP_HD bool hw_fma32(u64 a, u64 b, u64 c, u64& z, unsigned& fl)          // a * b + c
{
  WSP_FP_STRICT
  if (!(hw_ord32(a) && hw_ord32(b) && hw_ord32(c))) return false;
  const float xf = hw_f32(a), yf = hw_f32(b), wf = hw_f32(c);
  const double p = (double)xf * (double)yf, w = (double)wf, s = p + w;  // p exact
  const double bv = s - p, av = s - bv, e = (p - av) + (w - bv);        // p + w = s + e exactly
  // one rounding of the exact value to binary32: through binary64 unless that lands on a binary32 tie (then the error term decides)
  const u64 sb = hw_bits(s);
  float r;
  if (e == 0.0 || (sb & 0x1fffffffULL) != 0x10000000ULL) r = (float)s;  // s is not a tie, or is the exact value
  else r = __builtin_fmaf(xf, yf, wf);
  const u32 rb = hw_bits32(r);
  if (!hw_res32_ok(rb, s == 0.0 && e == 0.0)) return false;
  z = rb; if ((double)r != s || e != 0.0) fl |= FL_NX;
  return true;
}

WSP_HD u64 hw_int_to_f32(u64 v, bool is_signed, unsigned& fl)                     // 32-bit integer -> binary32: always in the domain
{
  const u32 w = (u32)v;
  const u32 mag = (is_signed && (i32)w < 0) ? 0u - w : w;
  if (mag && (u32)((mag << __builtin_clz(mag)) << 24)) fl |= FL_NX;               // more than 24 significant bits
  return hw_bits32(is_signed ? (float)(i32)w : (float)w);
}

}  // namespace wsf
