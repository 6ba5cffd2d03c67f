// optik_oracle_flops.cpp -- the oracle compiled with a COUNTING double: how many f64 operations the reference
// algorithm performs per restart ("measure with the oracle's counters").
//
// THIS IS TEST / MEASUREMENT INFRASTRUCTURE, like the oracle itself.  optik_oracle.c is included unchanged with
// `double` replaced by a one-member struct whose arithmetic operators count: every + - * / and sqrt the restatement
// executes -- FK, Jacobian, log maps, weighting, the software sin / cos / atan2, NLopt's SLSQP down to the
// Householder transformations of NNLS, the ChaCha seed's u64 -> f64 map -- is counted exactly once, as the CPU
// executes it.  Same arithmetic as the plain build (the struct holds one double; results are bit-identical,
// checked by tests/test_oracle_flops.py).  Counters are process-wide and not thread-safe: run with n_threads = 1.
//
// What is NOT a flop here: comparisons, negation, fabs, min / max, int <-> double conversions (counted apart).
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

namespace {
unsigned long long g_cnt[8];  // 0 add/sub, 1 mul, 2 div, 3 sqrt, 4 compare, 5 neg/abs/min/max, 6 conversions
}

struct cdouble {
    double v;
    cdouble() = default;
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
    cdouble(T x) : v((double)x) {}
    template <class T, class = typename std::enable_if<std::is_integral<T>::value>::type>
    explicit operator T() const { ++g_cnt[6]; return (T)v; }
    friend cdouble operator+(cdouble a, cdouble b) { ++g_cnt[0]; return cdouble(a.v + b.v); }
    friend cdouble operator-(cdouble a, cdouble b) { ++g_cnt[0]; return cdouble(a.v - b.v); }
    friend cdouble operator*(cdouble a, cdouble b) { ++g_cnt[1]; return cdouble(a.v * b.v); }
    friend cdouble operator/(cdouble a, cdouble b) { ++g_cnt[2]; return cdouble(a.v / b.v); }
    friend cdouble operator-(cdouble a) { ++g_cnt[5]; return cdouble(-a.v); }
    friend cdouble operator+(cdouble a) { return a; }
    cdouble &operator+=(cdouble b) { ++g_cnt[0]; v += b.v; return *this; }
    cdouble &operator-=(cdouble b) { ++g_cnt[0]; v -= b.v; return *this; }
    cdouble &operator*=(cdouble b) { ++g_cnt[1]; v *= b.v; return *this; }
    cdouble &operator/=(cdouble b) { ++g_cnt[2]; v /= b.v; return *this; }
    friend bool operator<(cdouble a, cdouble b) { ++g_cnt[4]; return a.v < b.v; }
    friend bool operator>(cdouble a, cdouble b) { ++g_cnt[4]; return a.v > b.v; }
    friend bool operator<=(cdouble a, cdouble b) { ++g_cnt[4]; return a.v <= b.v; }
    friend bool operator>=(cdouble a, cdouble b) { ++g_cnt[4]; return a.v >= b.v; }
    friend bool operator==(cdouble a, cdouble b) { ++g_cnt[4]; return a.v == b.v; }
    friend bool operator!=(cdouble a, cdouble b) { ++g_cnt[4]; return a.v != b.v; }
    bool operator!() const { return v == 0.0; }
    explicit operator bool() const { return v != 0.0; }
};
static_assert(sizeof(cdouble) == sizeof(double) && std::is_trivially_copyable<cdouble>::value, "same layout as double");

static inline cdouble cd_sqrt(cdouble a) { ++g_cnt[3]; return cdouble(::sqrt(a.v)); }
static inline cdouble cd_fabs(cdouble a) { ++g_cnt[5]; return cdouble(::fabs(a.v)); }
static inline cdouble cd_fmin(cdouble a, cdouble b) { ++g_cnt[5]; return cdouble(::fmin(a.v, b.v)); }
static inline cdouble cd_fmax(cdouble a, cdouble b) { ++g_cnt[5]; return cdouble(::fmax(a.v, b.v)); }
static inline int cd_isinf(cdouble a) { return ::isinf(a.v); }
static inline int cd_isnan(cdouble a) { return ::isnan(a.v); }
static inline int cd_isfinite(cdouble a) { return ::isfinite(a.v); }
// (platform libm: only pose_from_rpy -- URDF loading, outside the restart loop -- uses these)
static inline cdouble cd_sin(cdouble a) { return cdouble(::sin(a.v)); }
static inline cdouble cd_cos(cdouble a) { return cdouble(::cos(a.v)); }
static inline cdouble cd_atan2(cdouble a, cdouble b) { return cdouble(::atan2(a.v, b.v)); }
static inline cdouble cd_floor(cdouble a) { return cdouble(::floor(a.v)); }
static inline void cd_sincos(cdouble a, cdouble *s, cdouble *c) { double ss, cc; ::sincos(a.v, &ss, &cc); s->v = ss; c->v = cc; }

#undef isinf
#undef isnan
#undef isfinite
#define sqrt cd_sqrt
#define fabs cd_fabs
#define fmin cd_fmin
#define fmax cd_fmax
#define isinf cd_isinf
#define isnan cd_isnan
#define isfinite cd_isfinite
#define sin cd_sin
#define cos cd_cos
#define atan2 cd_atan2
#define floor cd_floor
#define sincos cd_sincos
#define double cdouble
#define restrict
#include "optik_oracle.c"
#undef double

extern "C" {
// out8: adds/subs, muls, divs, sqrts, compares, sign / abs / min / max, int <-> f64 conversions, 0
void ok_flop_counts(unsigned long long *out8) { for (int i = 0; i < 8; ++i) out8[i] = g_cnt[i]; }
void ok_flop_reset(void) { for (int i = 0; i < 8; ++i) g_cnt[i] = 0; }
}
