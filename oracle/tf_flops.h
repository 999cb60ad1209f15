/*
 * tf_flops.h - TEST INFRASTRUCTURE, developer build only (make -C oracle flops).
 *
 * Instrumented build of the oracle: tf_oracle.c is compiled as C++ with every `float` replaced by the wrapper below, whose operators
 * do the same IEEE fp32 arithmetic and count what they do.  The count is the exact number of fp32 operations the scalar restatement
 * executes per env-step - the figure SURVEY.md section 8(d) asks for ("to be replaced by an exact count from the CPU restatement's
 * instrumented build") and bench.py reports as `flops_per_env_step` (tools/count_flops.py writes it to profiles/).
 *
 * Categories: add (add and subtract), mul, fma (one operation here, two FLOP by the usual convention), div, sqrt, cmp (ordered
 * comparisons and the min / max / abs helpers built from them), cvt (int <-> float).  Negation is free (a source modifier on the GPU).
 * The library it builds has the same C ABI as the oracle plus tf_flop_counts / tf_flop_reset; single-threaded.
 */
#pragma once
#ifndef __cplusplus
#error "the counting build compiles tf_oracle.c as C++ (see oracle/Makefile, target flops)"
#endif
#include <math.h>
#include <stdint.h>
#include <type_traits>

enum { TFC_ADD = 0, TFC_MUL, TFC_FMA, TFC_DIV, TFC_SQRT, TFC_CMP, TFC_CVT, TFC_N };
static uint64_t tf_flop_counter[TFC_N];

struct cf32 {
    float v;
    cf32() = default;
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
    cf32(T x) : v((float)x) { if (std::is_integral<T>::value) ++tf_flop_counter[TFC_CVT]; }
    explicit operator float() const { return v; }
    explicit operator double() const { return (double)v; }
    explicit operator int() const { ++tf_flop_counter[TFC_CVT]; return (int)v; }
    explicit operator unsigned() const { ++tf_flop_counter[TFC_CVT]; return (unsigned)v; }
    explicit operator long() const { ++tf_flop_counter[TFC_CVT]; return (long)v; }
    explicit operator unsigned long() const { ++tf_flop_counter[TFC_CVT]; return (unsigned long)v; }
    explicit operator long long() const { ++tf_flop_counter[TFC_CVT]; return (long long)v; }
    explicit operator unsigned char() const { ++tf_flop_counter[TFC_CVT]; return (unsigned char)v; }
    explicit operator bool() const { return v != 0.0f; }
    cf32 operator-() const { cf32 r; r.v = -v; return r; }
    cf32 operator+() const { return *this; }
    cf32& operator+=(cf32 o) { ++tf_flop_counter[TFC_ADD]; v = v + o.v; return *this; }
    cf32& operator-=(cf32 o) { ++tf_flop_counter[TFC_ADD]; v = v - o.v; return *this; }
    cf32& operator*=(cf32 o) { ++tf_flop_counter[TFC_MUL]; v = v * o.v; return *this; }
    cf32& operator/=(cf32 o) { ++tf_flop_counter[TFC_DIV]; v = v / o.v; return *this; }
};
static_assert(sizeof(cf32) == 4 && std::is_trivially_copyable<cf32>::value && std::is_standard_layout<cf32>::value, "same ABI as float");

static inline cf32 cf_raw(float x) { cf32 r; r.v = x; return r; }
#define TFC_BIN(op, slot)                                                                                         \
    static inline cf32 operator op(cf32 a, cf32 b) { ++tf_flop_counter[slot]; return cf_raw(a.v op b.v); }        \
    static inline cf32 operator op(cf32 a, float b) { ++tf_flop_counter[slot]; return cf_raw(a.v op b); }         \
    static inline cf32 operator op(float a, cf32 b) { ++tf_flop_counter[slot]; return cf_raw(a op b.v); }         \
    static inline cf32 operator op(cf32 a, int b) { ++tf_flop_counter[slot]; return cf_raw(a.v op (float)b); }    \
    static inline cf32 operator op(int a, cf32 b) { ++tf_flop_counter[slot]; return cf_raw((float)a op b.v); }    \
    static inline double operator op(cf32 a, double b) { return (double)a.v op b; }                               \
    static inline double operator op(double a, cf32 b) { return a op (double)b.v; }
TFC_BIN(+, TFC_ADD)
TFC_BIN(-, TFC_ADD)
TFC_BIN(*, TFC_MUL)
TFC_BIN(/, TFC_DIV)
#define TFC_CMPOP(op)                                                                                             \
    static inline bool operator op(cf32 a, cf32 b) { ++tf_flop_counter[TFC_CMP]; return a.v op b.v; }             \
    static inline bool operator op(cf32 a, float b) { ++tf_flop_counter[TFC_CMP]; return a.v op b; }              \
    static inline bool operator op(float a, cf32 b) { ++tf_flop_counter[TFC_CMP]; return a op b.v; }              \
    static inline bool operator op(cf32 a, int b) { ++tf_flop_counter[TFC_CMP]; return a.v op (float)b; }         \
    static inline bool operator op(cf32 a, double b) { ++tf_flop_counter[TFC_CMP]; return (double)a.v op b; }
TFC_CMPOP(<)
TFC_CMPOP(>)
TFC_CMPOP(<=)
TFC_CMPOP(>=)
TFC_CMPOP(==)
TFC_CMPOP(!=)

static inline cf32 cf_fma(cf32 a, cf32 b, cf32 c) { ++tf_flop_counter[TFC_FMA]; return cf_raw(__builtin_fmaf(a.v, b.v, c.v)); }
static inline cf32 sqrtf(cf32 a) { ++tf_flop_counter[TFC_SQRT]; return cf_raw(__builtin_sqrtf(a.v)); }
static inline cf32 rintf(cf32 a) { ++tf_flop_counter[TFC_CVT]; return cf_raw(__builtin_rintf(a.v)); }
static inline cf32 fabsf(cf32 a) { ++tf_flop_counter[TFC_CMP]; return cf_raw(__builtin_fabsf(a.v)); }
static inline cf32 floorf(cf32 a) { ++tf_flop_counter[TFC_CVT]; return cf_raw(__builtin_floorf(a.v)); }
static inline double sqrt(cf32 a) { return sqrt((double)a.v); }
static inline double sin(cf32 a) { return sin((double)a.v); }
static inline double cos(cf32 a) { return cos((double)a.v); }
static inline double exp(cf32 a) { return exp((double)a.v); }
static inline double log(cf32 a) { return log((double)a.v); }
static inline double asin(cf32 a) { return asin((double)a.v); }
static inline double fabs(cf32 a) { return fabs((double)a.v); }
static inline bool cf_isnan(cf32 a) { return a.v != a.v; }

extern "C" {
/* counters since the last reset: add, mul, fma, div, sqrt, cmp, cvt */
__attribute__((visibility("default"))) void tf_flop_counts(uint64_t out[TFC_N]) { for (int i = 0; i < TFC_N; ++i) out[i] = tf_flop_counter[i]; }
__attribute__((visibility("default"))) void tf_flop_reset(void) { for (int i = 0; i < TFC_N; ++i) tf_flop_counter[i] = 0; }
}

#define float cf32
