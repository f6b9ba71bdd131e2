// rr_sim.hpp -- the phases of the batched lockstep simulator for gfx950 (CDNA4): GameEnv.step of RoboRugby's SimpleDuel3.
//
// A 64-lane wavefront is cut into 64 / VW "virtual waves" of VW lanes and each of them owns one arena (VW = 8 for the 2+2 / 4+4
// shape, 2 for the 1+0 / 1+0 training shape: one lane per entity and nothing idle; VW = 64 is literally one wavefront per arena).
// The arena's persistent record (SoA over entities) is pulled from HBM into its slice of LDS with lane-strided loads, the 12
// physics sub-steps of GameEnv.step (reference RR_EnvBase.py:275-287) run out of LDS, and the record is written back.  A sub-step
// is two fused lane-parallel phases (hooks + moves + broad phase | roll + broad phase) on the common, contact-free path; the
// reference-shaped loops run only when a ballot reports something close: narrow phases one lane per (pair, diameter, side),
// hits reduced with the arena's share of the wavefront ballot, and the contact RESPONSES -- which mutate state the next response
// reads (RR_EnvBase.py:373-388) -- applied in the reference's list order per entity: bounces of different balls side by side in
// slots of lanes, each response on a lane pair (x | y component, first | second chain; "lane pairs" below).
//
// The source is written against a few macros (RR_FOR_LANES / RR_FOR_PAIRS / RR_SYNC / RR_VOTE / RR_LANE_VAR / RR_XOR1) so that the
// very same phases also compile with g++ as lane loops: tests/emu builds that variant to check the phase logic against the CPU
// oracle without a GPU.  It is a test harness only -- the product library is the HIP build and nothing else.
//
// Real = double is the parity mode (the reference computes in Python floats = fp64); Real = float is the fast mode.  Incremental
// edge bookkeeping (MyUtils.py:141-148), int-truncated wall rects (RR_Ball.py:8-15), Python's float %, FloatRect.copy()'s
// re-derived centres and the list order of the responses are reproduced so that knife-edge branches agree with the reference.
// The default build deviates from it in two places -- its own sin / cos (< 1 ulp) and the scratch rect placed exactly on the
// ball -- which the parity build (-DRR_EXACT_TRIG=1: double-double sin / cos + the scratch-rect carry) removes; DESIGN.md section 2.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define RR_GPU 1
#else
#define RR_GPU 0
#endif

#if defined(__HIPCC__)
#define RR_HD __host__ __device__ __forceinline__
// rarely-taken paths (contact responses, reset).  Measured on MI355X: inlining them too and capping registers
// with __launch_bounds__ (4 waves/SIMD) beats real calls, whose ABI pins values in high callee-saved VGPRs.
#ifdef RR_OUTLINE_RARE
#define RR_HDN __host__ __device__ __noinline__
#else
#define RR_HDN __host__ __device__ __forceinline__
#endif
#else
#define RR_HD inline
#define RR_HDN
#endif

// A wavefront is split into 64/VW "virtual waves" of VW lanes (C::VW, a power of two); each owns one arena.
// Everything below is ordinary SIMT code in which `l` is the lane's index inside its virtual wave: control flow
// is uniform inside a virtual wave by construction (it only depends on that arena's ballots and LDS), so its
// lanes stay converged, while different virtual waves of one wavefront may diverge on the rare response paths.
// VW = 64 is the plain one-wavefront-per-arena mapping; smaller VW packs several small arenas per wavefront.
#if RR_GPU
#define RR_LANE_ID() ((int)(threadIdx.x & (C::VW - 1)))
#define RR_FOR_LANES(l) for (int l = RR_LANE_ID(), l##_o = 0; l##_o < 1; ++l##_o)
#define RR_IS_LANE0 (RR_LANE_ID() == 0)
// LDS is only shared inside the wave: a wavefront-scope release/acquire pair orders it
#define RR_SYNC()                                                    \
    do {                                                             \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
        __builtin_amdgcn_wave_barrier();                             \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
    } while (0)
// ballot restricted to this lane's virtual wave (inactive / diverged lanes contribute 0)
#define RR_VOTE(mask, l, pred) (mask) = rr::vballot<C::VW>(pred)
// a per-lane value that lives across two lane-parallel phases: a plain register here, an array in the host emulation
#define RR_LANE_VAR(type, name) type name
#define RR_LV(name, l) name
// the same variable in the neighbouring lane (l ^ 1): one DPP move per dword, no LDS.  Every lane of the pair must
// execute it (DPP does not read lanes that EXEC has switched off), so it is only used at converged points.
#define RR_XOR1(name, l) rr::lane_xor1(name)
#else
#define RR_FOR_LANES(l) for (int l = 0; l < C::VW; ++l)
#define RR_IS_LANE0 true
#define RR_SYNC() do { } while (0)
#define RR_VOTE(mask, l, pred) (mask) |= ((uint64_t)((pred) ? 1 : 0)) << (l)
#define RR_LANE_VAR(type, name) type name[C::VW]
#define RR_LV(name, l) name[l]
#define RR_XOR1(name, l) name[(l) ^ 1]
#endif

// Diagnostic build only (-DRR_PROFILE_PHASES): per-phase cycle totals of the leader lane of every 64th wavefront,
// accumulated with atomics into a device array that tools/phase_profile.py reads (sampled: stamping every wave puts
// ~600 k same-address atomics into each launch and measures mostly those).  Never compiled into the product library.
#if defined(__HIPCC__) && defined(RR_PROFILE_PHASES)
__device__ unsigned long long g_rr_prof[32];
__device__ unsigned long long g_rr_wave_t[2 * 65536]; // [start, end] s_memtime of every wavefront of the last k_step launch
#endif
#if RR_GPU && defined(RR_PROFILE_PHASES)
#define RR_T0() unsigned long long rr_t0_ = __builtin_amdgcn_s_memtime()
#define RR_TR() rr_t0_ = __builtin_amdgcn_s_memtime()
#define RR_STAMP(id)                                                                          \
    do {                                                                                      \
        unsigned long long rr_t1_ = __builtin_amdgcn_s_memtime();                             \
        if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0) atomicAdd(&g_rr_prof[id], rr_t1_ - rr_t0_);              \
        rr_t0_ = rr_t1_;                                                                      \
    } while (0)
#else
#define RR_T0() do { } while (0)
#define RR_TR() do { } while (0)
#define RR_STAMP(id) do { } while (0)
#endif

// host emulation only: print the response branches taken (tests bisect mismatches against the oracle with it)
#if !RR_GPU && defined(RR_EMU_TRACE)
#include <stdio.h>
#define RR_TRACE(...) do { if (RR_EMU_TRACE) fprintf(stderr, __VA_ARGS__); } while (0)
#else
#define RR_TRACE(...) do { } while (0)
#endif

// The frozen island's "witness" tests (substep): default build only -- in the parity build every extra detection sweep would walk the
// reference's scratch rect.  -DRR_NO_WITNESS: A/B builds.
#if defined(RR_NO_WITNESS)
#define RR_WITNESS 0
#else
#define RR_WITNESS (!RR_CARRY)
#endif
// rare branches (contact paths, reset, frozen islands): tells the register allocator where spilling is cheap
#define RR_UNLIKELY(x) __builtin_expect(!!(x), 0)

#ifdef RR_NO_OBS_STAGE // A/B builds only: observation rows written by one lane, value by value
#define RR_OBS_STAGE 0
#else
#define RR_OBS_STAGE 1
#endif
#ifndef RR_NUM_SUBSTEPS
#define RR_NUM_SUBSTEPS 12 // MOVES_PER_FRAME (RR_Constants.py:13); only the emulation harness overrides it to bisect
#endif

namespace rr {

#if RR_GPU
template <int VW> __device__ __forceinline__ uint64_t vballot(bool pred) {
    uint64_t m = __ballot(pred);
    if (VW == 64) return m;
    const int sh = (int)(threadIdx.x & 63 & ~(VW - 1));
    return (m >> sh) & ((1ull << (VW & 63)) - 1ull);
}
#endif

#if RR_GPU
__device__ __forceinline__ int lane_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true); }
__device__ __forceinline__ float lane_xor1(float v) { return __int_as_float(lane_xor1(__float_as_int(v))); }
__device__ __forceinline__ double lane_xor1(double v) {
    return __hiloint2double(lane_xor1(__double2hiint(v)), lane_xor1(__double2loint(v)));
}
#endif

enum : int {
    ST_BOT_RESOLVE_FAIL = 1, ST_BOT_STUCK = 2, ST_UNDO_MOVE_FAIL = 4, ST_UNDO_FAIL = 8, ST_SAME_SPOT = 16,
    ST_DIV0 = 32, ST_STEP_AFTER_DONE = 64, ST_BAD_ACTION = 128, ST_UNDO_WARN = 256, ST_RESET_GAVE_UP = 512,
    ST_WAS_RESET = 1024,
    ST_GOAL_H_DESTROYED = 2048, ST_GOAL_G_DESTROYED = 4096, ST_NO_BALLS = 8192, // opt-in goal scoring (rr_extras.hpp: goal_step)
    ST_NOT_READY = 16384 // budgeted step: the arena's step is still in progress (parked mid-step); its outputs were not written
};

// ------------------------------------------------------------------------------------------------ math
RR_HD double m_atan(double x) { return ::atan(x); }
RR_HD double m_sqrt(double x) { return ::sqrt(x); }
RR_HD double m_fmod(double a, double b) { return ::fmod(a, b); }
RR_HD double m_abs(double x) { return ::fabs(x); }
RR_HD double m_rint(double x) { return ::rint(x); }
RR_HD float m_atan(float x) { return ::atanf(x); }
RR_HD float m_sqrt(float x) { return ::sqrtf(x); }
RR_HD float m_fmod(float a, float b) { return ::fmodf(a, b); }
RR_HD float m_abs(float x) { return ::fabsf(x); }
RR_HD float m_rint(float x) { return ::rintf(x); }

// sin & cos of one angle.  fp64: Cody-Waite reduction by pi/2 + the fdlibm kernel polynomials (|x| stays below
// a few pi here: angles are degrees in [-90, 810) converted to radians).  Absolute error < 1e-16, which is what
// matters for corner offsets of length ~22; much cheaper in registers and instructions than the generic ocml
// path with its Payne-Hanek fallback, and identical on the host emulation and on the GPU.
#ifndef RR_EXACT_TRIG
#define RR_EXACT_TRIG 0
#endif
// The parity build also carries the centre of the reference's module-global scratch rect (_rectBallInner, RR_TrashyPhysics.py:26-36)
// from one use to the next: see "scratch-rect carry" below.  (Every instantiation of that build carries it; only its fp64
// configurations are parity modes -- BatchedRoboRugbyEnv refuses exact_trig with dtype f32.)
#ifndef RR_CARRY
#define RR_CARRY RR_EXACT_TRIG
#endif
// phase 1's broad tests: after the radius bounds, also the separating-axis / robot-frame bounds (with the motion slack), so that fewer
// wavefronts enter the reference-shaped loops for pairs that merely pass each other.  A/B only: G -4 %, T -1 % (the extra tests
// cost the hot phase more than the skipped loops save; profiles/r03/broad_tight_ab.txt) -- the frozen variants always use them.
#ifndef RR_BROAD_TIGHT
#define RR_BROAD_TIGHT 0
#endif
#if RR_EXACT_TRIG
// The "exact trig" build (libroborugby_amd_exact.so, BatchedRoboRugbyEnv(exact_trig=True)): sin & cos to ~2^-63 of their true values
// before the one final rounding, i.e. correctly rounded except on ~1 argument in 10^4.  Why: the reference's math.sin / math.cos
// (glibc) are within 0.55 ulp and return the correctly rounded value for 99.86 % of this path's arguments; the fast routine below
// (Cody-Waite + fdlibm kernels, < 1 ulp) agrees with glibc on 97.0 % of them, this one on 99.81 % -- and a 1-ulp difference in
// a robot's move is what eventually makes a free-running episode leave the reference's trajectory (DESIGN.md section 2,
// profiles/r03/divergence_attribution.txt: the first cause in 21 of 25 departures).  Double-double arithmetic from exactly specified
// operations only (+, *, fma), so GPU and host emulation still agree bit for bit.  ~4x the instructions of the fast routine:
// an opt-in parity build, not the default.
struct DD { double h, l; };
RR_HD DD dd_two_sum(double a, double b) { const double s = a + b, bb = s - a; DD r = { s, (a - (s - bb)) + (b - bb) }; return r; }
RR_HD DD dd_fast_two_sum(double a, double b) { const double s = a + b; DD r = { s, b - (s - a) }; return r; }
RR_HD DD dd_two_prod(double a, double b) { const double p = a * b; DD r = { p, ::fma(a, b, -p) }; return r; }
RR_HD DD dd_add(DD a, DD b) { const DD s = dd_two_sum(a.h, b.h); return dd_fast_two_sum(s.h, a.l + b.l + s.l); }
RR_HD DD dd_add_d(DD a, double b) { const DD s = dd_two_sum(a.h, b); return dd_fast_two_sum(s.h, s.l + a.l); }
RR_HD DD dd_mul(DD a, DD b) { const DD p = dd_two_prod(a.h, b.h); return dd_fast_two_sum(p.h, ::fma(a.h, b.l, ::fma(a.l, b.h, p.l))); }
RR_HD DD dd_mul_d(DD a, double b) { const DD p = dd_two_prod(a.h, b); return dd_fast_two_sum(p.h, ::fma(a.l, b, p.l)); }
RR_HD void m_sincos(double x, double &s, double &c) {
    const double fn = ::rint(x * 6.36619772367581382433e-01);
    const int n = (int)fn;
    // y = x - fn * pi/2 with pi/2 = P1 + P2 + P3 (33 + 33 + 53 bits; fn * P1 and fn * P2 are exact for the few turns seen here)
    const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11, P3 = 2.02226624879595063154e-21;
    DD y = dd_add(dd_two_sum(::fma(-fn, P1, x), -fn * P2), dd_two_prod(-fn, P3));
    const DD z = dd_mul(y, y);
    // sin y = y + y^3 (S1 + z (S2 + z Q(z))), cos y = 1 - z/2 + z^2 (C1 + z (C2 + z R(z))): Taylor coefficients, the two leading ones of
    // each series in double-double (the tails contribute < 2^-66 of the result each)
    const DD S1 = { -1.66666666666666657415e-01, -9.25185853854297065662e-18 }, S2 = { 8.33333333333333321769e-03, 1.15648231731787138300e-19 };
    const double S3 = -1.98412698412698412526e-04, S4 = 2.75573192239858925110e-06, S5 = -2.50521083854417202239e-08,
                 S6 = 1.60590438368216133188e-10, S7 = -7.64716373181981641101e-13, S8 = 2.81145725434552059500e-15;
    const DD C1 = { 4.16666666666666643537e-02, 2.31296463463574266163e-18 }, C2 = { -1.38888888888888894189e-03, 5.30054395437357706435e-20 };
    const double C3 = 2.48015873015873015658e-05, C4 = -2.75573192239858882758e-07, C5 = 2.08767569878680989792e-09,
                 C6 = -1.14707455977297245138e-11, C7 = 4.77947733238738524956e-14, C8 = -1.56192069685862252710e-16;
    const double zh = z.h;
    const double qs = ::fma(zh, ::fma(zh, ::fma(zh, ::fma(zh, ::fma(zh, S8, S7), S6), S5), S4), S3);
    const double qc = ::fma(zh, ::fma(zh, ::fma(zh, ::fma(zh, ::fma(zh, C8, C7), C6), C5), C4), C3);
    const DD sn = dd_add(y, dd_mul(dd_mul(y, z), dd_add(S1, dd_mul(z, dd_add_d(S2, zh * qs)))));
    const DD cs = dd_add(dd_add_d(dd_mul_d(z, -0.5), 1.0), dd_mul(dd_mul(z, z), dd_add(C1, dd_mul(z, dd_add_d(C2, zh * qc)))));
    const double ks = sn.h + sn.l, kc = cs.h + cs.l;
    const int q = n & 3;
    s = (q == 0) ? ks : (q == 1) ? kc : (q == 2) ? -ks : -kc;
    c = (q == 0) ? kc : (q == 1) ? -ks : (q == 2) ? -kc : ks;
}
#else
RR_HD void m_sincos(double x, double &s, double &c) {
    // explicit fused multiply-adds (the build runs with -ffp-contract=off so that the REFERENCE arithmetic is never
    // contracted; this routine is ours, and fma is exactly specified, so GPU and host emulation still agree bit for bit)
    const double fn = ::rint(x * 6.36619772367581382433e-01);
    const int n = (int)fn;
    const double z0 = ::fma(-fn, 1.57079632673412561417e+00, x); // pio2_1: first 33 bits of pi/2 (exact product)
    const double w0 = fn * 6.07710050650619224932e-11;             // pio2_1t
    const double y = z0 - w0, yl = (z0 - y) - w0;
    const double z = y * y;
    // __kernel_sin(y, yl, 1)
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z * y;
    const double rs = ::fma(z, ::fma(z, ::fma(z, ::fma(z, S6, S5), S4), S3), S2);
    const double ks = y - ::fma(-v, S1, ::fma(z, ::fma(0.5, yl, -v * rs), -yl));
    // __kernel_cos(y, yl)
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double w = z * z;
    const double rc = ::fma(w * w, ::fma(z, ::fma(z, C6, C5), C4), z * ::fma(z, ::fma(z, C3, C2), C1));
    const double hz = 0.5 * z, wc = 1.0 - hz;
    const double kc = wc + (((1.0 - wc) - hz) + ::fma(z, rc, -y * yl));
    const int q = n & 3;
    s = (q == 0) ? ks : (q == 1) ? kc : (q == 2) ? -ks : -kc;
    c = (q == 0) ? kc : (q == 1) ? -ks : (q == 2) ? -kc : ks;
}
#endif
RR_HD void m_sincos(float x, float &s, float &c) { s = ::sinf(x); c = ::cosf(x); }

template <typename R> RR_HD R inf_() { return (R)INFINITY; }
template <typename R> RR_HD bool is_inf(R x) { return x == inf_<R>() || x == -inf_<R>(); }
template <typename R> RR_HD bool is_nan(R x) { return x != x; }
template <typename R> RR_HD R pi_() { return (R)3.14159265358979323846; }
template <typename R> RR_HD R radians(R x) { return x * (pi_<R>() / (R)180.0); } // math.radians
template <typename R> RR_HD R degrees(R x) { return x * ((R)180.0 / pi_<R>()); } // math.degrees
// sin & cos of an angle given in degrees, as the reference takes them: math.sin(math.radians(deg)).  fp64: exactly that.
// fp32 fast mode: the angles here reach 810 degrees = 14 rad, where one fp32 ulp of the ARGUMENT is 1e-6 rad -- 1.5e-5 px on
// the 16-px pivot arm, every sub-step.  Bringing the angle into [-180, 180] first is exact in fp32 (Sterbenz) and shrinks
// that fourfold; the function value is the same.
template <typename R> RR_HD void sincos_deg(R deg, R &s, R &c) { m_sincos(radians<R>(deg), s, c); }
template <> RR_HD void sincos_deg<float>(float deg, float &s, float &c) {
    float d = deg;
    d = d >= 360.0f ? d - 360.0f : d;
    d = d >= 360.0f ? d - 360.0f : d;
    d = d > 180.0f ? d - 360.0f : d;
    d = d < -180.0f ? d + 360.0f : d;
    m_sincos(radians<float>(d), s, c);
}
template <typename R> RR_HD R py_mod(R a, R b) {                                 // Python float %
    R m = m_fmod(a, b);
    if (m != (R)0) {
        if ((b < (R)0) != (m < (R)0)) m += b;
    } else {
        m = (R)0;
    }
    return m;
}
// Every float % on this path is `x % 360` with |x| of a few thousand at most.  fmod's result is exact, so it can be had
// without libm's iterative reduction: q = floor(x / 360) up to one unit (the quotient comes from a rounded reciprocal
// product), r = x - q * 360 in ONE rounding (fma; the true remainder is representable, so that rounding is exact),
// and the off-by-one cases fixed by adding / subtracting 360 -- exact again for the same reason.  Bit-identical to
// Python's % (checked against fmod on 2e8 values incl. the neighbours of every multiple of 360).  Domain: |x| < 2^40.
template <> RR_HD double py_mod<double>(double a, double b) {
    if (b != 360.0) {
        double m = m_fmod(a, b);
        if (m != 0.0) { if ((b < 0.0) != (m < 0.0)) m += b; } else m = 0.0;
        return m;
    }
    double r;
    if (a >= 0.0) {
        const double q = ::floor(a * 2.77777777777777788e-03);
        r = ::fma(-q, 360.0, a);
        r = (r < 0.0) ? r + 360.0 : ((r >= 360.0) ? r - 360.0 : r);
    } else { // never on the step path (angles are normalised before they get here); a correctly rounded quotient can
             // only be one too large, and then r + 360 is Python's own single rounding (a tiny negative x gives 360.0)
        const double q = ::floor(a / 360.0);
        r = ::fma(-q, 360.0, a);
        r = (r < 0.0) ? r + 360.0 : r;
    }
    return r == 0.0 ? 0.0 : r;
}
// `(x + 720) % 360`, the reference's angle normalisation (MyUtils.py:279, :98).  fp64: exactly that, rounding of x + 720
// included (parity).  fp32 fast mode: the sum would drop the bits of x below ulp(1024) = 6e-5 degrees on EVERY turn, and 16 px
// of pivot arm turn that into ~1e-4 px per step; the float instance evaluates the same expression in fp64 and rounds once, so
// an angle already in [0, 360) comes back unchanged, as it does in exact arithmetic.
template <typename R> RR_HD R norm360(R x) { return py_mod<R>(x + (R)720, (R)360); }
template <> RR_HD float norm360<float>(float x) {
    const float r = (float)py_mod<double>((double)x + 720.0, 360.0);
    return r >= 360.0f ? 0.0f : r; // 359.99999999 rounds up to the float 360 = 0 on the circle
}
template <typename R> RR_HD R py_max(R a, R b) { return (b > a) ? b : a; } // first maximal wins
template <typename R> RR_HD R py_min(R a, R b) { return (b < a) ? b : a; }

// lowest set bit first: walks a hit mask in the reference's list order without visiting the empty positions
RR_HD int low_bit(uint32_t m) { return __builtin_ctz(m); }
RR_HD int low_bit(uint64_t m) { return __builtin_ctzll(m); }

template <typename R> struct V2 { R x, y; };
template <typename R> struct Seg { V2<R> a, b; };

// MyUtils.py:17-25.  0/0 raises in the reference -> status bit + 0
template <typename R> RR_HD R div0(R n, R d, int &st) {
    if (d != (R)0) return n / d;
    if (n > (R)0) return inf_<R>();
    if (n < (R)0) return -inf_<R>();
    st |= ST_DIV0;
    return (R)0;
}
// MyUtils.py:40-41
template <typename R> RR_HD R dist(V2<R> a, V2<R> b) {
    R dx = b.x - a.x, dy = b.y - a.y;
    return m_sqrt(dx * dx + dy * dy);
}
// MyUtils.py:44-58
template <typename R> RR_HD void slope_yint(V2<R> a, V2<R> b, R &m, R &c, int &st) {
    m = div0<R>(b.y - a.y, b.x - a.x, st);
    if (m == inf_<R>()) c = -inf_<R>();
    else if (m == -inf_<R>()) c = inf_<R>();
    else c = a.y - a.x * m;
}
// MyUtils.py:61-85 with the two slope/intercept pairs already known.  Written with selects instead of
// branches (the values are the reference's in every case; lanes in a wave take different cases).
template <typename R> RR_HD V2<R> intersect_mb(R m1, R b1, R x1, R m2, R b2, R x2) {
    const bool i1 = is_inf(m1), i2 = is_inf(m2);
    const bool par = (m1 == m2) | (i1 & i2);
    R gx = (b1 - b2) / (m2 - m1);
    R x = i1 ? x1 : (i2 ? x2 : gx);
    const bool use1 = i2 | (!i1 & (m_abs(b1) < m_abs(b2))); // which line formula gives y
    R y = use1 ? (m1 * x + b1) : (m2 * x + b2);
    V2<R> r;
    r.x = par ? inf_<R>() : x;
    r.y = par ? inf_<R>() : y;
    return r;
}
template <typename R> RR_HD V2<R> line_intersection(Seg<R> l1, Seg<R> l2, int &st) {
    R m1, b1, m2, b2;
    slope_yint(l1.a, l1.b, m1, b1, st);
    slope_yint(l2.a, l2.b, m2, b2, st);
    return intersect_mb<R>(m1, b1, l1.a.x, m2, b2, l2.a.x);
}
// MyUtils.py:88-94 (non-short-circuit form: same truth table, no branches)
template <typename R> RR_HD bool within(V2<R> p, Seg<R> l, R buf) {
    const bool wx = ((l.a.x - buf <= p.x) & (p.x <= l.b.x + buf)) | ((l.b.x - buf <= p.x) & (p.x <= l.a.x + buf));
    const bool wy = ((l.a.y - buf <= p.y) & (p.y <= l.b.y + buf)) | ((l.b.y - buf <= p.y) & (p.y <= l.a.y + buf));
    return wx & wy;
}
// Can `within(q, s, 0) & within(q, d, buf)` hold for ANY point q?  Only if the two accepted boxes meet -- the very bounds within()
// compares against (rounded the same way), so a "no" here is the conjunction's "no" whatever the intersection point is, inf and
// NaN included.  The contact paths of the narrow virtual waves (fewer than eight lanes per arena: a lane sweeps several candidates one
// after the other) use it to skip the two divisions of a candidate that cannot hit -- a ball touches one side with one diameter,
// seven of the eight (side, diameter) candidates stop here: T +2.5 %, T chase +7 %.  With eight lanes and more every candidate
// has its own lane, the division runs anyway for the one that can hit, and the test only adds instructions (G chase -3 %).
template <typename R> RR_HD bool boxes_meet(Seg<R> s, Seg<R> d, R buf) {
    const R sx0 = py_min<R>(s.a.x, s.b.x), sx1 = py_max<R>(s.a.x, s.b.x), sy0 = py_min<R>(s.a.y, s.b.y), sy1 = py_max<R>(s.a.y, s.b.y);
    const R dx0 = py_min<R>(d.a.x - buf, d.b.x - buf), dx1 = py_max<R>(d.a.x + buf, d.b.x + buf);
    const R dy0 = py_min<R>(d.a.y - buf, d.b.y - buf), dy1 = py_max<R>(d.a.y + buf, d.b.y + buf);
    return (sx0 <= dx1) & (dx0 <= sx1) & (sy0 <= dy1) & (dy0 <= sy1);
}
// MyUtils.py:97-110
template <typename R> RR_HD R angle_degrees(V2<R> a, V2<R> b, int &st) {
    R dy = b.y - a.y, dx = b.x - a.x;
    R ang = m_atan(div0<R>(dy, dx, st));
    if (dx < (R)0) ang += pi_<R>();
    return norm360<R>(degrees<R>((R)2 * pi_<R>() - ang));
}

// ------------------------------------------------------------------------------------------------ config
// Precision policy of a configuration: `double` / `float` = state and arithmetic in that type; F32State = the arena's record in HBM is
// fp32 ("fp32 state", BASELINE config 2) while a step computes in fp64 from the moment the record is loaded into LDS to the moment it
// is written back -- one rounding per stored value and step, which is what keeps a contact step within 1e-5 of the fp64 reference
// (the all-fp32 mode's cancellations in the contact responses do not: DESIGN.md section 3).
struct F32State {};
template <typename T> struct Precision { using Real = T; using Store = T; };
template <> struct Precision<F32State> { using Real = double; using Store = float; };
template <int NRH_, int NRG_, int NBP_, int NBN_, typename Prec_, int VW_ = 64> struct Cfg {
    static constexpr int NRH = NRH_, NRG = NRG_, NBP = NBP_, NBN = NBN_;
    static constexpr int NR = NRH_ + NRG_, NB = NBP_ + NBN_;
    static constexpr int NPR = NR * (NR - 1) / 2; // robot pairs
    static constexpr int NPB = NB * (NB - 1) / 2; // ball pairs
    static constexpr int VW = VW_;                // lanes per arena (virtual wave width)
    using Real = typename Precision<Prec_>::Real;   // arithmetic, and the arena's LDS image
    using Store = typename Precision<Prec_>::Store; // the persistent record in HBM
    static constexpr bool MIXED = sizeof(Store) != sizeof(Real);
    static_assert(VW == 2 || VW == 4 || VW == 8 || VW == 16 || VW == 32 || VW == 64, "VW must divide the wavefront");
    static_assert(NR >= 1 && NR <= VW && NB >= 1 && NB <= VW, "one lane per entity");
    static_assert(NB * NR <= 32 && NPB <= 64 && NB <= 16 && NR <= 16, "pair masks are 32/64-bit, per-ball masks 16-bit");
};

template <typename R> struct SimParams {
    R W, H;
    R mult_ball, mult_robot;     // RR_Constants.py:46,50
    R rob_cdist;                 // FloatRect._corner_dist of a 20x40 rect (MyUtils.py:138)
    R inner_h, inner_cdist;      // half side / corner dist of _rectBallInner (RR_TrashyPhysics.py:29-35)
    int32_t game_len, game_mode, time_limit, auto_reset, reset_on_fault;
    int32_t memo;                // stop the sub-step loop at a bitwise fixed point (exact shortcut, see step_arena)
    int32_t acc_external;        // a non-default keeper program owns the episode-return accumulators (k_extras_end), not step_arena
    uint64_t seed, arena_offset;
};

// corner / side numbering of FloatRect (MyUtils.py:326-336, :209-229)
enum { TL = 0, TR = 1, BL = 2, BR = 3 };
RR_HD int side_a(int s) { return s == 0 ? TR : s == 1 ? TL : s == 2 ? BL : BR; } // RIGHT,TOP,LEFT,BOTTOM
RR_HD int side_b(int s) { return s == 0 ? BR : s == 1 ? TR : s == 2 ? TL : BL; }

// ------------------------------------------------------------------------------------------------ arena (LDS image)
template <class C> struct ArenaBody {
    using R = typename C::Real;
    static constexpr int NR = C::NR, NB = C::NB;
    // ---- persistent: identical order to the HBM record (field-major = SoA over the entities)
    struct P {
        R rcx[NR], rcy[NR], rl[NR], rrt[NR], rt[NR], rb[NR], rrot[NR], px[NR], py[NR], prot[NR];
        R bcx[NB], bcy[NB], bl[NB], brt[NB], bt[NB], bb[NB], bvx[NB], bvy[NB];
        R acc[4]; // running return happy/grumpy, last finished return happy/grumpy
#if RR_CARRY
        R ic[2];  // centre of the reference's scratch rect _rectBallInner as its last user left it ("scratch-rect carry")
#endif
#ifdef RR_REL_IN_RECORD // A/B builds only: the corner offsets travel with the record instead of being rebuilt by derive()
        R rel[NR][9];
#endif
    } p;
    struct I {
        int32_t mc[NR], thl[NR], thr[NR];
        int32_t step, episode, ep_len, ep_count, last_len, fault;
        int32_t fzp, fexc; // the island that was still frozen when the last step ended (step_arena: "island freeze across steps")
    } i;
    // ---- per-step scratch
    // (rows of 9, not 8: a 16-word row stride would put every robot's row of every arena of a bank group on the same 4 banks)
#ifndef RR_REL_IN_RECORD
    R rel[NR][9];   // corner offsets TL,TR,BL,BR (x,y) for the current rotation
#endif
    R irot[NR];     // rotation irel was built for (NaN = stale)
    R sm[NR][4], sc[NR][4]; // slope / y-intercept of the four sides (get_slope_yint, MyUtils.py:44-58) for the current pose
    R ax[NR], ay[NR], arot[NR]; // pose at frame begin (= ring entry written this frame)
    R psx[NR], psy[NR]; // robot centre at step begin (rectDblPriorStep)
    R bfx[NB], bfy[NB], pfx[NB], pfy[NB];
#if RR_CARRY
    uint8_t bmass[NB];         // (1..3; bytes, so that the two reals of P::ic do not cost the G slice its eighth workgroup per CU)
#else
    int32_t bmass[NB];
#endif
    // (parity build: bbm doubles as the per-ball mask of the robots with a corner inside the ball -- the pairs of a ball-robot sweep
    // that return before touching the scratch rect; the ball-pair sweep's use of it is over by then)
    uint16_t bbm[NB], brc[NB]; // per-ball hit / close bit masks of the contact sweeps (one lane per ball writes its own; 16 bits:
                               // NB <= 11, NR <= 8 -- the G slice has no byte to spare, see wm below)
    R exc[NB];                // how far (L1) the contact responses of this sub-step have carried the ball from its frame-begin centre
    R reach[NB];              // 14.04 + the most the ball can travel in this sub-step's roll: the ball-ball bound of the fused roll phase
    int32_t sides_ok; // sm/sc match the current robot poses (rebuilt lazily by the first phase that needs them)
    uint8_t wm[NR];   // how the robot's last move met the walls: bit 0 = blocked (reverted), bit 1 = clamped (the island freeze
                      // compares it); bytes, so that it fits the alignment hole behind sides_ok (the slice is at its 8-waves-per-CU limit)
    union { // the lidar candidates are only alive inside observe(), the inner-square offsets only inside a sub-step
        R irel[NR][9];                    // corner offsets of the ball's inner square at rot+45 (diameter end points)
        R lidar[2][3 * NR];               // [front|back][ray, rect] minima over the rect's four sides
    } u;
    R lid[6];                             // capped minima: front/back per ray
#ifdef RR_LDS_EXTRA
    R extra_[NR > 1 ? RR_LDS_EXTRA : 1]; // occupancy experiments only
#endif
    static constexpr int P_REALS = (int)(sizeof(P) / sizeof(R));
    static constexpr int I_INTS = (int)(sizeof(I) / sizeof(int32_t));
    // HBM record of one arena: the P reals, the I ints right behind them, padded to a 64-B multiple (so that the lane-strided
    // runs of the arenas sharing a wavefront start on request boundaries).  G/fp64: 108 reals + 20 ints = 944 -> 960 B.
    // (the record holds `Store` values -- fp32 under the F32State policy --, everything else here is in arithmetic reals)
    static constexpr int WR = (int)(sizeof(R) / 4);                       // 32-bit words per real
    static constexpr int WS = (int)(sizeof(typename C::Store) / 4);       // 32-bit words per stored real
    static constexpr int P_ALIGN = 64 / (int)sizeof(typename C::Store);
    static constexpr int P_STRIDE = (P_REALS + (I_INTS + WS - 1) / WS + P_ALIGN - 1) / P_ALIGN * P_ALIGN; // stored reals per record
    static constexpr int I_STRIDE = P_STRIDE * WS;                         // the same stride seen from the int part (irecs = recs + P_REALS)
    // fixed-point snapshot (its own buffer, touched only by arenas that ran the expensive contact paths): P + ax, ay, arot
    static constexpr int SNAP_WORDS = (P_REALS + 3 * NR) * WR, ISNAP_WORDS = 2 * NR;
    // parked mid-step state of the budgeted step (its own buffer, touched only by arenas that park): PARK_INTS ints, then
    // ax, ay, arot, psx, psy [NR each], dist_sum0, exc [NB]
    // (+ when it parks between two passes of a resolve loop: the loop's registers, bmass [NB], and bfx, bfy, pfx, pfy [NB each])
    static constexpr int PARK_INTS = 20 + NB + ((20 + NB) & 1), PARK_REALS = 5 * NR + 1 + 5 * NB, PARK_WORDS = PARK_INTS + PARK_REALS * WR;
};
// LDS bank spreading.  The 64/VW arenas of a wavefront sit in consecutive LDS slices and their lanes touch the SAME
// field at the same time, so the slice stride decides the banking: lane-strided accesses of one arena cover
// win = VW * (words per Real) consecutive banks, and the arenas sharing a bank group (32 lanes for ds_read_b32/b64 with
// 32/64 banks, 16 lanes for ds_write_b64 / ds_read2_b64 with 32 banks: MI355X_MICROARCH.md, LDS) tile the banks exactly
// when stride = win * (odd number).  G/fp64/VW=8 was 576 words = 9 * 64: every arena on the same banks, 4-way
// conflicts on every access (SQ_LDS_BANK_CONFLICT = 4x the LDS instruction cycles).  VW >= 32: one arena per group.
constexpr int lds_pad_words(int body_words, int vw, int real_words) {
#ifdef RR_NO_LDS_PAD // A/B builds only
    return 0;
#endif
    if (vw >= 32) return 0;
    const int win = vw * real_words;
    int pad = (win - body_words % win) % win;
    if (((body_words + pad) / win) % 2 == 0) pad += win;
    return pad;
}
template <class C, int PAD> struct ArenaPadded : ArenaBody<C> { uint32_t lds_pad_[PAD]; };
template <class C> struct ArenaPadded<C, 0> : ArenaBody<C> {};
template <class C> using Arena = ArenaPadded<C, lds_pad_words((int)(sizeof(ArenaBody<C>) / 4), C::VW, (int)(sizeof(typename C::Real) / 4))>;

#ifdef RR_REL_IN_RECORD
#define RR_REL(A) (A).p.rel
#else
#define RR_REL(A) (A).rel
#endif
// ------------------------------------------------------------------------------------------------ FloatRect in registers
template <typename R> struct FR {
    R cx, cy, l, r, t, b, rot;
    R rel[8];
};
template <typename R> RR_HD void fr_move(FR<R> &f, R dx, R dy) { // MyUtils.py:141-148
    f.cx += dx; f.l += dx; f.r += dx;
    f.cy += dy; f.t += dy; f.b += dy;
}
template <typename R> RR_HD void fr_set_left(FR<R> &f, R v) { fr_move<R>(f, v - f.l, (R)0); }
template <typename R> RR_HD void fr_set_right(FR<R> &f, R v) { fr_move<R>(f, v - f.r, (R)0); }
template <typename R> RR_HD void fr_set_top(FR<R> &f, R v) { fr_move<R>(f, (R)0, v - f.t); }
template <typename R> RR_HD void fr_set_bottom(FR<R> &f, R v) { fr_move<R>(f, (R)0, v - f.b); }
template <typename R> RR_HD void fr_set_cx(FR<R> &f, R v) { fr_move<R>(f, v - f.cx, (R)0); }
template <typename R> RR_HD void fr_set_cy(FR<R> &f, R v) { fr_move<R>(f, (R)0, v - f.cy); }

// body of the rotation setter (MyUtils.py:284-316): rotate the initial corners (+-hw, +-hh) by
// 360-rot degrees and renormalise them to the corner distance
// one corner of the rotation setter (MyUtils.py:284-316): the initial corner (ix, iy) rotated by (s, c) = sin/cos of
// radians(360 - rot) and renormalised to the corner distance; rot == 0 keeps the initial corner (MyUtils.py:298-300)
template <typename R> RR_HD void corner_from_sc(R rot, R s, R c, R ix, R iy, R cdist, R &ax, R &ay) {
    R qx = ix * c - iy * s, qy = ix * s + iy * c;
    R d = m_sqrt(qx * qx + qy * qy);
    ax = qx * cdist / d; ay = qy * cdist / d;
    const bool z = rot == (R)0;
    ax = z ? ix : ax; ay = z ? iy : ay;
}
template <typename R> RR_HD void corners_from_sc(R rot, R s, R c, R hw, R hh, R cdist, R *rel) {
    // initial corners TL(-hw,-hh) TR(hw,-hh) BL(-hw,hh) BR(hw,hh): BR = -TL and BL = -TR, and every operation
    // is odd-symmetric in (x,y), so two corners are computed and two are exact negations
    const R ix[2] = { -hw, hw }, iy = -hh; // TL, TR
    for (int k = 0; k < 2; k++) {
        R ax, ay;
        corner_from_sc<R>(rot, s, c, ix[k], iy, cdist, ax, ay);
        rel[2 * k] = ax; rel[2 * k + 1] = ay;                  // TL / TR
        rel[2 * (3 - k)] = -ax; rel[2 * (3 - k) + 1] = -ay;    // BR / BL
    }
}
template <typename R> RR_HD void corners_for(R rot, R hw, R hh, R cdist, R *rel) {
    R c, s;
    sincos_deg<R>((R)360 - rot, s, c);
    corners_from_sc<R>(rot, s, c, hw, hh, cdist, rel);
}
RR_HD double m_min(double a, double b) { return ::fmin(a, b); }
RR_HD double m_max(double a, double b) { return ::fmax(a, b); }
RR_HD float m_min(float a, float b) { return ::fminf(a, b); }
RR_HD float m_max(float a, float b) { return ::fmaxf(a, b); }
template <typename R> RR_HD void fr_edges_from_rel(FR<R> &f) { // MyUtils.py:318-322
    // min / max of the four corner offsets as v_min / v_max (one instruction each instead of compare + two selects: G +0.4 %, T +1 %,
    // profiles/r03/micro_ab.txt).  The same values as the reference's `if x < mn: mn = x` chain: the offsets are never NaN and never
    // +-0 (axis-aligned poses have offsets of exactly +-10 / +-20, every other pose a nonzero renormalised product).
    const R mnx = m_min(m_min(f.rel[0], f.rel[2]), m_min(f.rel[4], f.rel[6])), mxx = m_max(m_max(f.rel[0], f.rel[2]), m_max(f.rel[4], f.rel[6]));
    const R mny = m_min(m_min(f.rel[1], f.rel[3]), m_min(f.rel[5], f.rel[7])), mxy = m_max(m_max(f.rel[1], f.rel[3]), m_max(f.rel[5], f.rel[7]));
    f.l = mnx + f.cx; f.r = mxx + f.cx; f.t = mny + f.cy; f.b = mxy + f.cy;
}
template <typename R> RR_HD void fr_set_rot(FR<R> &f, R nr, R cdist) { // robot rect: 20 x 40
    nr = norm360<R>(nr);
    if (nr == f.rot) return;
    f.rot = nr;
    corners_for<R>(nr, (R)10, (R)20, cdist, f.rel);
    fr_edges_from_rel<R>(f);
}

template <class C> RR_HD FR<typename C::Real> load_robot(const Arena<C> &A, int r) {
    FR<typename C::Real> f;
    f.cx = A.p.rcx[r]; f.cy = A.p.rcy[r]; f.l = A.p.rl[r]; f.r = A.p.rrt[r]; f.t = A.p.rt[r]; f.b = A.p.rb[r];
    f.rot = A.p.rrot[r];
    for (int k = 0; k < 8; k++) f.rel[k] = RR_REL(A)[r][k];
    return f;
}
template <class C> RR_HD void store_robot(Arena<C> &A, int r, const FR<typename C::Real> &f) {
    A.p.rcx[r] = f.cx; A.p.rcy[r] = f.cy; A.p.rl[r] = f.l; A.p.rrt[r] = f.r; A.p.rt[r] = f.t; A.p.rb[r] = f.b;
    A.p.rrot[r] = f.rot;
    for (int k = 0; k < 8; k++) RR_REL(A)[r][k] = f.rel[k];
    A.sides_ok = 0;
}
template <class C> RR_HD V2<typename C::Real> robot_corner(const Arena<C> &A, int r, int c) {
    V2<typename C::Real> v = { A.p.rcx[r] + RR_REL(A)[r][2 * c], A.p.rcy[r] + RR_REL(A)[r][2 * c + 1] };
    return v;
}
template <class C> RR_HD Seg<typename C::Real> robot_side(const Arena<C> &A, int r, int s) {
    Seg<typename C::Real> g = { robot_corner(A, r, side_a(s)), robot_corner(A, r, side_b(s)) };
    return g;
}

// ball rect helpers (rotation is always 0 for balls): incremental edges like the reference
template <class C> RR_HD void ball_shift(Arena<C> &A, int b, typename C::Real dx, typename C::Real dy) {
    A.p.bcx[b] += dx; A.p.bl[b] += dx; A.p.brt[b] += dx;
    A.p.bcy[b] += dy; A.p.bt[b] += dy; A.p.bb[b] += dy;
}

// ------------------------------------------------------------------------------------------------ robot kinematics (RR_Robot.py:139-234)
template <typename R> RR_HD bool rob_hit_wall(const FR<R> &f, const SimParams<R> &sp) {
    return f.l < (R)0 || f.r > sp.W || f.t <= (R)0 || f.b >= sp.H;
}
template <typename R> RR_HD bool rob_clamp(FR<R> &f, const SimParams<R> &sp) {
    const R buffer = (R).5;
    bool any = false;
    if (f.l < (R)0) { fr_set_left<R>(f, buffer); any = true; }
    if (f.r > sp.W) { fr_set_right<R>(f, sp.W - buffer); any = true; }
    if (f.t <= (R)0) { fr_set_top<R>(f, buffer); any = true; }
    if (f.b >= sp.H) { fr_set_bottom<R>(f, sp.H - buffer); any = true; }
    return any;
}
// Robot.move (RR_Robot.py:106-108,139-234).  Lanes of one wavefront drive robots with different thrust patterns; the
// three trig evaluations a move can need (heading or pivot direction, the rotation setter's 360-rot, the re-centring
// angle) are independent of each other, so they are issued up front as straight-line code -- one latency instead
// of three serialised divergent paths -- and the cheap bookkeeping then selects what its move type uses.  Every
// value is computed from the same operands in the same order as Robot._move_linear / _move_angular
// (RR_Robot.py:181-234): wall test on the incrementally kept edges, revert of centre (and rotation), 0.5 clamp.
// thrust pattern -> kind of move and the three angles whose sin/cos it can need
template <typename R> struct MovePlan { bool idle, lin, spin; int L; R off, nrot, a1, a2, a3; };
template <class C> RR_HD MovePlan<typename C::Real> robot_move_plan(const Arena<C> &A, int r) {
    using R = typename C::Real;
    MovePlan<R> m;
    const int L = A.i.thl[r], Rt = A.i.thr[r];
    const R rot = A.p.rrot[r];
    m.L = L;
    m.idle = (L == Rt && L == 0);
    m.lin = (L == Rt);
    m.spin = !m.lin && (L + Rt == 0);
    const R w = m.lin ? (R)0 : m.spin ? (Rt > 0 ? (R)1.2 : (R)-1.2) : ((Rt > 0 || L < 0) ? (R).6 : (R)-.6);
    m.off = (Rt != 0) ? (R)90 : (R)-90; // pivot = left track (rot+90) when the right one drives
    m.nrot = m.lin ? rot : norm360<R>(rot + w);
    m.a1 = m.lin ? rot : rot + m.off;   // heading (linear) / direction of the pivot (track) centre
    m.a2 = (R)360 - m.nrot;             // rotation setter
    m.a3 = m.nrot + -m.off;             // robot centre as seen from the pivot after the turn
    return m;
}
// the move itself once the trigonometry is there: (s1,c1), (s3,c3) and the new TL / TR corner offsets for nrot
template <class C>
RR_HD void robot_move_finish(Arena<C> &A, const SimParams<typename C::Real> &sp, int r, const MovePlan<typename C::Real> &m,
                             typename C::Real s1, typename C::Real c1, typename C::Real s3, typename C::Real c3,
                             typename C::Real tlx, typename C::Real tly, typename C::Real trx, typename C::Real try_) {
    using R = typename C::Real;
    FR<R> f = load_robot(A, r);
    const R rot_prior = f.rot, px = f.cx, py = f.cy;
    bool blocked;
    if (m.lin) {
        const R vel = m.L < 0 ? (R)-1 : (R)1;
        fr_set_left<R>(f, f.l + c1 * vel);
        fr_set_top<R>(f, f.t + s1 * vel * (R)-1);
        blocked = rob_hit_wall<R>(f, sp);
        if (blocked) { fr_set_cx<R>(f, px); fr_set_cy<R>(f, py); }
    } else {
        if (m.nrot != f.rot) {
            f.rot = m.nrot;
            f.rel[0] = tlx; f.rel[1] = tly; f.rel[2] = trx; f.rel[3] = try_;       // TL, TR
            f.rel[4] = -trx; f.rel[5] = -try_; f.rel[6] = -tlx; f.rel[7] = -tly;   // BL = -TR, BR = -TL
            fr_edges_from_rel<R>(f);
        }
        if (!m.spin) {
            const R cpx = px + (R)16 * c1, cpy = py - (R)16 * s1; // tplCenterRot, from the pose before the turn
            fr_set_cx<R>(f, cpx + (R)16 * c3);
            fr_set_cy<R>(f, cpy - (R)16 * s3);
        }
        blocked = rob_hit_wall<R>(f, sp);
        if (blocked) {
            fr_set_cx<R>(f, px); fr_set_cy<R>(f, py);
            // `self.rectDbl.rotation = dblPriorRot` (RR_Robot.py:226): the setter on (rot_prior + 720) % 360.  The
            // normalised value is almost always rot_prior itself (it came out of this very normalisation one move ago),
            // and then the corners the setter would rebuild are the ones still sitting in LDS: A.rel is always
            // corners_for(rrot) and store_robot has not run yet.  Only a value the +720 really perturbs pays the trig.
            const R nr = norm360<R>(rot_prior);
            if (nr != f.rot) {
                f.rot = nr;
                if (nr == rot_prior) {
                    for (int k = 0; k < 8; k++) f.rel[k] = RR_REL(A)[r][k];
                } else {
                    corners_for<R>(nr, (R)10, (R)20, sp.rob_cdist, f.rel);
                }
                fr_edges_from_rel<R>(f);
            }
        }
    }
    const bool clamped = rob_clamp<R>(f, sp);
    store_robot(A, r, f);
    A.wm[r] = (uint8_t)((blocked ? 1 : 0) | (clamped ? 2 : 0));
}
// one robot's whole move on one lane (thaw / edge replay of the island freeze, and configurations without a lane pair
// per robot); substep_phase1 spreads the same arithmetic over a pair of lanes
template <class C> RR_HD void robot_move_lane(Arena<C> &A, const SimParams<typename C::Real> &sp, int r) {
    using R = typename C::Real;
    const MovePlan<R> m = robot_move_plan(A, r);
    A.i.mc[r] += 1;
    if (m.idle) { A.wm[r] = 0; return; }
    R s1, c1, s2, c2, s3, c3;
    sincos_deg<R>(m.a1, s1, c1);
    sincos_deg<R>(m.a2, s2, c2);
    sincos_deg<R>(m.a3, s3, c3);
    R tlx, tly, trx, try_;
    corner_from_sc<R>(m.nrot, s2, c2, (R)-10, (R)-20, sp.rob_cdist, tlx, tly);
    corner_from_sc<R>(m.nrot, s2, c2, (R)10, (R)-20, sp.rob_cdist, trx, try_);
    robot_move_finish(A, sp, r, m, s1, c1, s3, c3, tlx, tly, trx, try_);
}
// Robot.undo_move (RR_Robot.py:110-137): back to the pose stored at frame begin
template <class C> RR_HD void robot_undo_lane(Arena<C> &A, const SimParams<typename C::Real> &sp, int r) {
    using R = typename C::Real;
    FR<R> f = load_robot(A, r);
    fr_set_cx<R>(f, A.ax[r]);
    fr_set_cy<R>(f, A.ay[r]);
    fr_set_rot<R>(f, A.arot[r], sp.rob_cdist); // no-op unless the undone move rotated (rare path: trig is fine here)
    store_robot(A, r, f);
    A.i.mc[r] -= 1;
}
// inner-square corner offsets for robot r at rot+45 (RR_TrashyPhysics.py:54-55,93)
template <class C> RR_HD void refresh_inner_lane(Arena<C> &A, const SimParams<typename C::Real> &sp, int r) {
    using R = typename C::Real;
    R rot = A.p.rrot[r];
    if (A.irot[r] == rot) return;
    A.irot[r] = rot;
    corners_for<R>(norm360<R>(rot + (R)45), sp.inner_h, sp.inner_h, sp.inner_cdist, A.u.irel[r]);
}

// side slope/intercept cache: one lane per (robot, side), rebuilt on demand after robot poses changed
template <class C> RR_HD void ensure_sides(Arena<C> &A) {
    using R = typename C::Real;
    if (A.sides_ok) return; // same value in every lane of the arena (read after a sync)
    for (int base = 0; base < 4 * C::NR; base += C::VW) {
        RR_FOR_LANES(l) {
            int t = base + l;
            if (t < 4 * C::NR) {
                int r = t >> 2, sd = t & 3, st = 0;
                Seg<R> g = robot_side(A, r, sd);
                R m, c;
                slope_yint<R>(g.a, g.b, m, c, st);
                A.sm[r][sd] = m; A.sc[r][sd] = c;
            }
        }
    }
    RR_SYNC();
    if (RR_IS_LANE0) A.sides_ok = 1;
    RR_SYNC();
}
// Exact broad phase.  A hit of robots_collided / ball_robot_collided needs an intersection point inside the
// bounding boxes of BOTH segments (MyUtils.py:88-94): every side point is within |corner| = 22.36 of its robot
// centre, every diameter point within 7*sqrt(2) = 9.9 of the ball centre, a corner hit needs |corner-ball| < 7.
// So centres farther apart than 2*22.36 (robots) or 22.36+9.9 (ball-robot) can never hit; the constants below
// leave > 0.7 px of slack, far above any rounding.  Arenas where nothing is close skip the narrow phase.
template <typename R> RR_HD R cull_rr2() { return (R)(45.5 * 45.5); }
template <typename R> RR_HD R cull_br2() { return (R)(33.0 * 33.0); }

// Second, tighter exact bound in the robot's own frame.  ball_robot_collided hits only if a robot corner is within 7 of
// the ball centre or one of the two diameters -- 14 long, centred on the ball, parallel / perpendicular to the robot's
// sides -- really crosses a side segment (an intersection point inside the bounding boxes of both segments lies on
// both segments).  Either way the ball centre is within 7 of the 20 x 40 rectangle along each robot axis:
// |x_local| <= 10 + 7, |y_local| <= 20 + 7.  The axes come from the corner offsets (unit to ~1e-15); 0.05 px of slack.
template <class C> RR_HD bool ball_near_robot(const Arena<C> &A, int b, int r) {
    using R = typename C::Real;
    const R dx = A.p.bcx[b] - A.p.rcx[r], dy = A.p.bcy[b] - A.p.rcy[r];
    if (!(dx * dx + dy * dy <= cull_br2<R>())) return false;
    const R *q = RR_REL(A)[r];
    const R ux = (q[2] - q[0]) * (R)0.05, uy = (q[3] - q[1]) * (R)0.05;     // TL -> TR, 20 long
    const R vx = (q[4] - q[0]) * (R)0.025, vy = (q[5] - q[1]) * (R)0.025;   // TL -> BL, 40 long
    const R lx = dx * ux + dy * uy, ly = dx * vx + dy * vy;
    // Beyond BOTH side lines (the four 7 x 7 squares off the corners) no diameter reaches a side segment -- the one parallel
    // to a side passes it outside its end points, the other is parallel to the other side -- so only that corner's radius test
    // can hit there: within 7 of the corner.  This is where a ball pushed by a corner comes to rest (7.2-7.8 px from it), and it
    // used to send every sub-step of such an arena through the narrow phase twice.
    const R ax = m_abs(lx), ay = m_abs(ly), cx = ax - (R)10, cy = ay - (R)20;
    const bool corner_zone = (cx > (R)0.05) & (cy > (R)0.05);
    return (ax <= (R)17.05) & (ay <= (R)27.05) & (!corner_zone | (cx * cx + cy * cy <= (R)(7.05 * 7.05)));
}

// Third exact cull, robot against robot: separating-axis test of the two 20 x 40 rectangles.  robots_collided hits only if
// the intersection of two side LINES lies inside the bounding boxes of both SEGMENTS, i.e. (to rounding, ~1e-13) on both
// segments: the rectangles touch.  If some side direction of either robot separates them by more than 0.05 px, they do
// not.  (Nearly parallel sides make the intersection ill-conditioned, but then its x or y is off by far more than a
// bounding box is wide unless the lines coincide to that precision -- in which case no axis separates them.)  Most
// robot pairs that come within 45.5 px never overlap -- a colliding move is undone -- so this keeps the side-slope
// cache and the 16-test narrow phase for the real collisions.
template <class C>
RR_HD bool robots_separated(const Arena<C> &A, int i, int j, typename C::Real dx, typename C::Real dy, typename C::Real m = (typename C::Real)0.05) {
    using R = typename C::Real;
    const R *p = RR_REL(A)[i], *q = RR_REL(A)[j];
    // unit axes (to ~1e-15) from the corner offsets: TL -> TR is 20 long, TL -> BL 40
    const R uix = (p[2] - p[0]) * (R)0.05, uiy = (p[3] - p[1]) * (R)0.05, vix = (p[4] - p[0]) * (R)0.025, viy = (p[5] - p[1]) * (R)0.025;
    const R ujx = (q[2] - q[0]) * (R)0.05, ujy = (q[3] - q[1]) * (R)0.05, vjx = (q[4] - q[0]) * (R)0.025, vjy = (q[5] - q[1]) * (R)0.025;
    const R uu = m_abs(uix * ujx + uiy * ujy), uv = m_abs(uix * vjx + uiy * vjy), vu = m_abs(vix * ujx + viy * ujy), vv = m_abs(vix * vjx + viy * vjy);
    const bool s0 = m_abs(dx * uix + dy * uiy) > (R)10 + ((R)10 * uu + (R)20 * uv) + m; // axis u_i
    const bool s1 = m_abs(dx * vix + dy * viy) > (R)20 + ((R)10 * vu + (R)20 * vv) + m; // axis v_i
    const bool s2 = m_abs(dx * ujx + dy * ujy) > (R)10 + ((R)10 * uu + (R)20 * vu) + m; // axis u_j
    const bool s3 = m_abs(dx * vjx + dy * vjy) > (R)20 + ((R)10 * uv + (R)20 * vv) + m; // axis v_j
    return s0 | s1 | s2 | s3;
}

// ------------------------------------------------------------------------------------------------ contact predicates, one task per lane
// robots_collided (RR_TrashyPhysics.py:18-24): task = (pair, side of bot1, side of bot2)
template <class C> RR_HD void pair_of(int p, int n, int &i, int &j) { // p-th (i<j) pair in nested-loop order
    i = 0;
    while (p >= n - 1 - i) { p -= n - 1 - i; i++; }
    j = i + 1 + p;
}
template <class C> RR_HD uint32_t detect_robot_pairs(Arena<C> &A) {
    using R = typename C::Real;
    if (C::NPR == 0) return 0;
    // broad phase: one lane per pair
    uint32_t close = 0;
    for (int base = 0; base < C::NPR; base += C::VW) {
        uint64_t m = 0;
        RR_FOR_LANES(l) {
            bool c = false;
            int t = base + l;
            if (t < C::NPR) {
                int i, j;
                pair_of<C>(t, C::NR, i, j);
                R dx = A.p.rcx[j] - A.p.rcx[i], dy = A.p.rcy[j] - A.p.rcy[i];
                c = (dx * dx + dy * dy <= cull_rr2<R>()) && !robots_separated(A, i, j, dx, dy);
            }
            RR_VOTE(m, l, c);
        }
        close |= (uint32_t)(m << base);
    }
    if (!RR_UNLIKELY(close)) return 0;
    ensure_sides(A);
    // narrow phase: the 16 (side, side) tests of each close pair, VW of them per round
    uint32_t pairs = 0;
#pragma unroll 1
    for (uint32_t todo = close; todo; todo &= todo - 1) {
        const int p = low_bit(todo);
        int i, j;
        pair_of<C>(p, C::NR, i, j);
        uint64_t any = 0;
        for (int base = 0; base < 16; base += C::VW) {
            uint64_t m = 0;
            RR_FOR_LANES(l) {
                bool hit = false;
                int t = base + l;
                if (t < 16) {
                    int s1 = t >> 2, s2 = t & 3;
                    Seg<R> g1 = robot_side(A, i, s1), g2 = robot_side(A, j, s2);
                    V2<R> q = intersect_mb<R>(A.sm[i][s1], A.sc[i][s1], g1.a.x, A.sm[j][s2], A.sc[j][s2], g2.a.x);
                    hit = within<R>(q, g1, (R)0) & within<R>(q, g2, (R)0);
                }
                RR_VOTE(m, l, hit);
            }
            any |= m;
        }
        if (any) pairs |= 1u << p;
    }
    return pairs;
}
#if RR_CARRY
// ---- scratch-rect carry (parity build only).
// The reference keeps ONE module-global FloatRect, _rectBallInner (RR_TrashyPhysics.py:26-36), whose corners give the ball diameters
// parallel / perpendicular to a robot's sides.  ball_robot_collided (:53), apply_force_to_ball (:94) and bounce_ball_off_bot (:165)
// each "move" it onto the ball with `_rectBallInner.center = ball.center`, and FloatRect's centre setters are relative moves
// (MyUtils.py:266-275: `_move_linear(new - old)`, i.e. c <- c + (x - c)) -- so the centre the diameters are built from is
// fl(c_old + fl(x - c_old)), which differs from x in the last bits whenever c_old is far away (a different ball), and that c_old
// is whatever the previous user left: the rect carries state from pair to pair, sweep to sweep, step to step and (one module,
// many envs) episode to episode.  The default build places the rect exactly on the ball -- a <= 1e-13 px difference in a diameter's
// end points that only a free-running episode ever notices.  Here the carry is reproduced:
//   * within one collision sweep (collision_pairs: ball-major, every robot for every ball, RR_TrashyPhysics.py:352-362) ball b's
//     FIRST pair that reaches the setter -- its lowest robot without a corner inside the ball (:49-51 return before it) --
//     sees c1(b) = carry(c_in(b), x_b); every later pair of the ball sees x_b exactly (x_b - c1(b) is exact by Sterbenz and so is
//     the sum); the rect leaves ball b at x_b (>= 2 setter calls), c1(b) (one) or c_in(b) (none), which is c_in(b + 1);
//   * c_in(0) of a sweep and the c_old of a response is A.p.ic, the centre the last user left; it lives in the arena's record
//     (rr_set_scratch_rect / rr_get_scratch_rect seed and read it), since a reset moves the balls but not the rect.
template <typename R> RR_HD R carry1(R c_old, R x) { return c_old + (x - c_old); }
template <class C> RR_HD int corner_free_robots(const Arena<C> &A, int b) { return C::NR - __builtin_popcount((unsigned)A.bbm[b]); }
// centre of the scratch rect when the sweep reaches ball `upto` (= what ball upto - 1 left behind); the corner masks (A.bbm) must be current
template <class C> RR_HD V2<typename C::Real> carry_chain(const Arena<C> &A, int upto) {
    using R = typename C::Real;
    V2<R> c = { A.p.ic[0], A.p.ic[1] };
    for (int b = 0; b < upto; b++) {
        const int nset = corner_free_robots(A, b);
        const V2<R> c1 = { carry1<R>(c.x, A.p.bcx[b]), carry1<R>(c.y, A.p.bcy[b]) };
        if (nset >= 2) { c.x = A.p.bcx[b]; c.y = A.p.bcy[b]; }
        else if (nset == 1) c = c1;
    }
    return c;
}
// the centre pair (b, r) builds its diameters from
template <class C> RR_HD V2<typename C::Real> carry_centre(const Arena<C> &A, int b, int r) {
    using R = typename C::Real;
    const unsigned cm = A.bbm[b], low = (1u << r) - 1u;
    V2<R> x = { A.p.bcx[b], A.p.bcy[b] };
    if ((cm & low) != low) return x; // a lower robot already moved the rect onto this ball
    const V2<R> c = carry_chain(A, b);
    V2<R> c1 = { carry1<R>(c.x, x.x), carry1<R>(c.y, x.y) };
    return c1;
}
// The two sweeps the fast path of a sub-step skips (the push's, RR_EnvBase.py:335, and the resolve loop's one pass, :372) find no ball
// near any robot: every pair reaches the setter.  With two or more robots the rect therefore ends exactly ON THE LAST BALL whatever it
// held before; with one robot (one ball) it ends at carry(c, x), which is x again whenever the rect already sat on the ball's previous
// position and the roll is an exact difference (always, but for a ball flung across half its coordinate in one sub-step).  So on the
// quiet path nothing is written: `icm` = 1 says "the rect is on the last ball's frame-begin centre (A.pfx/pfy: pre-roll, and the same
// value as the centre at the end of the previous sub-step) -- or, between sub-steps, on its current centre"; the explicit A.p.ic is
// brought up to date (carry_materialize) only where somebody reads it: a real sweep, a response, a snapshot, the end of the step.
template <class C> RR_HD void carry_materialize(Arena<C> &A, int &icm, bool frame_begin) {
    if (icm) {
        RR_SYNC();
        if (RR_IS_LANE0) {
            A.p.ic[0] = frame_begin ? A.pfx[C::NB - 1] : A.p.bcx[C::NB - 1];
            A.p.ic[1] = frame_begin ? A.pfy[C::NB - 1] : A.p.bcy[C::NB - 1];
        }
        RR_SYNC();
        icm = 0;
    }
}
// a quiet sweep; `frame_begin`: the balls have not rolled yet (the push's sweep).  One robot: the explicit chain, until it lands on the ball
template <class C> RR_HD void carry_quiet_sweep(Arena<C> &A, int &icm, bool frame_begin) {
    if constexpr (C::NR >= 2) {
        icm = 1;
    } else {
        static_assert(C::NR >= 2 || C::NB == 1, "one robot: one ball");
        if (icm) return; // on the ball already (the roll's exactness is checked in substep_phase2: an inexact one takes the full path)
        RR_FOR_LANES(l) { if (l < C::NB) A.bbm[l] = 0; }
        RR_SYNC();
        const V2<typename C::Real> c = carry_chain(A, C::NB);
        RR_SYNC();
        if (RR_IS_LANE0) { A.p.ic[0] = c.x; A.p.ic[1] = c.y; }
        RR_SYNC();
        icm = (c.x == A.p.bcx[0] && c.y == A.p.bcy[0]) ? 1 : 0;
    }
}
// the balls outside an island `kb` whose position the island's sweeps and responses depend on through the scratch rect
template <class C> RR_HD uint32_t carry_deps(uint32_t kb) { return kb ? (((kb >> 1) | (1u << (C::NB - 1))) & ~kb) : 0u; }
// a response's setter call (apply_force_to_ball / bounce_ball_off_bot): returns the centre its diameters are built from
template <class C> RR_HD V2<typename C::Real> carry_response(Arena<C> &A, V2<typename C::Real> bc) {
    using R = typename C::Real;
    const V2<R> c = { carry1<R>(A.p.ic[0], bc.x), carry1<R>(A.p.ic[1], bc.y) };
    RR_TRACE("E scratch rect (%.17g,%.17g) -> (%.17g,%.17g) for ball at (%.17g,%.17g)\n", (double)A.p.ic[0], (double)A.p.ic[1], (double)c.x, (double)c.y, (double)bc.x, (double)bc.y);
    RR_SYNC();
    if (RR_IS_LANE0) { A.p.ic[0] = c.x; A.p.ic[1] = c.y; }
    RR_SYNC();
    return c;
}
#endif
// ball_robot_collided (RR_TrashyPhysics.py:39-69): task = (ball, robot, diameter); each lane tests two
// corners against the radius and its diameter against the four sides.  Bit (b*NR + r) of the result.
// CACHED = false (the first sweep of a sub-step, usually the only one): a lane computes everything it needs itself --
// the inner square's corner offsets for rot+45 (RR_TrashyPhysics.py:54-55) and the four side slopes -- so a quiet
// sub-step pays no cache-building phases.  CACHED = true (the later passes of the resolve / undo loops, where the
// robots stand still): the robot-only operands come from the LDS caches (irel keyed by rotation, sm/sc by pose), which
// leaves the diameter's own slope and the four intersections.  Same operands, same operations, same results.
template <class C, bool CACHED> RR_HD uint32_t detect_ball_robot(Arena<C> &A, const SimParams<typename C::Real> &sp) {
    using R = typename C::Real;
    // broad phase: one lane per ball sweeps the robots (radius bound, then the robot-frame bound) and publishes its mask
    RR_T0();
    uint64_t anyc = 0;
    RR_FOR_LANES(l) {
        bool c = false;
        if (l < C::NB) {
            int msk = 0;
#if RR_CARRY
            int cmk = 0;
            const V2<R> xb = { A.p.bcx[l], A.p.bcy[l] };
            for (int r = 0; r < C::NR; r++) {
                const bool near = ball_near_robot(A, l, r);
                msk |= near ? (1 << r) : 0;
                if (near) { // the corner tests that make ball_robot_collided return before it touches the scratch rect (:49-51)
                    bool ch = false;
                    for (int k = 0; k < 4; k++) ch = ch | (dist<R>(robot_corner(A, r, k), xb) < (R)7);
                    cmk |= ch ? (1 << r) : 0;
                }
            }
            A.bbm[l] = (uint16_t)cmk;
#else
            for (int r = 0; r < C::NR; r++) msk |= ball_near_robot(A, l, r) ? (1 << r) : 0;
#endif
            A.brc[l] = (uint16_t)msk;
            c = msk != 0;
        }
        RR_VOTE(anyc, l, c);
    }
#if RR_CARRY
    // where this sweep leaves the scratch rect; written once every lane has built its diameters from the centre it found
    auto carry_done = [&]() {
        RR_SYNC();
        const V2<R> cend = carry_chain(A, C::NB);
        RR_SYNC();
        if (RR_IS_LANE0) { A.p.ic[0] = cend.x; A.p.ic[1] = cend.y; }
        RR_SYNC();
    };
    if (!RR_UNLIKELY(anyc)) { carry_done(); return 0; }
#else
    if (!RR_UNLIKELY(anyc)) return 0;
#endif
    if (CACHED) RR_STAMP(26);
    RR_SYNC();
    uint32_t close = 0;
    for (int b = 0; b < C::NB; b++) close |= (uint32_t)A.brc[b] << (b * C::NR);
    if (CACHED) {
        RR_FOR_LANES(l) {
            if (l < C::NR) refresh_inner_lane(A, sp, l); // no-op while the rotation stands
        }
        RR_SYNC();
        ensure_sides(A);
    }
    if (CACHED) RR_STAMP(27);
    uint32_t pairs = 0;
    if constexpr (C::VW >= 8) {
        // Narrow phase, one lane per (close pair, diameter, side): eight lanes per pair, the close pairs taken in mask order,
        // VW / 8 of them per round.  A lane intersects ITS diameter with ITS side (and, for the first diameter, tests the side's
        // corner number against the radius): two dependent divisions instead of the five of a lane that sweeps the four sides,
        // and no round is spent on a ball's pairs that are not close.  The pair's hit is the OR over its eight lanes -- the very
        // predicates of RR_TrashyPhysics.py:39-69 on the very operands, only spread differently.
        constexpr int PPR = C::VW / 8;
#pragma unroll 1
        for (uint32_t todo = close; todo;) {
            int prs[PPR];
            for (int q = 0; q < PPR; q++) { prs[q] = todo ? low_bit(todo) : -1; todo &= todo - 1; }
            uint64_t m = 0;
            RR_FOR_LANES(l) {
                bool hit = false;
                int pr = prs[0];
                for (int q = 1; q < PPR; q++) pr = ((l >> 3) == q) ? prs[q] : pr; // by value: no dynamically indexed local array
                if (pr >= 0) {
                    const int d = (l >> 2) & 1, sd = l & 3, r = pr % C::NR, b = pr / C::NR;
                    int st = 0;
                    V2<R> bc = { A.p.bcx[b], A.p.bcy[b] };
                    if (d == 0) hit = dist<R>(robot_corner(A, r, sd), bc) < (R)7;
                    R ox, oy;
                    if (CACHED) {
                        ox = A.u.irel[r][2 * d]; oy = A.u.irel[r][2 * d + 1];
                    } else {
                        R iq[8];
                        corners_for<R>(norm360<R>(A.p.rrot[r] + (R)45), sp.inner_h, sp.inner_h, sp.inner_cdist, iq);
                        ox = d == 0 ? iq[0] : iq[2]; oy = d == 0 ? iq[1] : iq[3];
                    }
#if RR_CARRY
                    const V2<R> dc = carry_centre(A, b, r);
#else
                    const V2<R> dc = bc;
#endif
                    Seg<R> dia = { { dc.x + ox, dc.y + oy }, { dc.x + -ox, dc.y + -oy } };
                    Seg<R> side = robot_side(A, r, sd);
                    R md, cd, ms, cs;
                    slope_yint<R>(dia.a, dia.b, md, cd, st);
                    if (CACHED) { ms = A.sm[r][sd]; cs = A.sc[r][sd]; }
                    else slope_yint<R>(side.a, side.b, ms, cs, st);
                    V2<R> q = intersect_mb<R>(ms, cs, side.a.x, md, cd, dia.a.x);
                    hit = hit | (within<R>(q, side, (R)0) & within<R>(q, dia, (R)0));
                }
                RR_VOTE(m, l, hit);
            }
            for (int q = 0; q < PPR; q++)
                if (prs[q] >= 0 && ((m >> (8 * q)) & 0xFFull)) pairs |= 1u << prs[q];
        }
    } else {
    constexpr int NT = C::NB * C::NR * 2; // task = (pair, diameter)
    for (int base = 0; base < NT; base += C::VW) {
        constexpr uint32_t ALL = (C::VW >= 64) ? 0xFFFFFFFFu : ((1u << (C::VW / 2)) - 1u);
        if (!((close >> (base >> 1)) & ALL)) continue;
        uint64_t m = 0;
        RR_FOR_LANES(l) {
            bool hit = false;
            int t = base + l;
            if (t < NT && ((close >> (t >> 1)) & 1u)) {
                int d = t & 1, pr = t >> 1, r = pr % C::NR, b = pr / C::NR, st = 0;
                V2<R> bc = { A.p.bcx[b], A.p.bcy[b] };
                hit = (dist<R>(robot_corner(A, r, 2 * d), bc) < (R)7) | (dist<R>(robot_corner(A, r, 2 * d + 1), bc) < (R)7);
                // diameters (TL->BR) for d = 0 and (TR->BL) for d = 1 of the inner square; BR = -TL, BL = -TR
                R ox, oy;
                if (CACHED) {
                    ox = A.u.irel[r][2 * d]; oy = A.u.irel[r][2 * d + 1];
                } else {
                    R iq[8];
                    corners_for<R>(norm360<R>(A.p.rrot[r] + (R)45), sp.inner_h, sp.inner_h, sp.inner_cdist, iq);
                    ox = d == 0 ? iq[0] : iq[2]; oy = d == 0 ? iq[1] : iq[3];
                }
#if RR_CARRY
                const V2<R> dc = carry_centre(A, b, r);
#else
                const V2<R> dc = bc;
#endif
                Seg<R> dia = { { dc.x + ox, dc.y + oy }, { dc.x + -ox, dc.y + -oy } };
                bool need[4], any_need = false;
                for (int sd = 0; sd < 4; sd++) { need[sd] = boxes_meet<R>(robot_side(A, r, sd), dia, (R)0); any_need = any_need | need[sd]; }
                if (any_need) {
                    R md, cd;
                    slope_yint<R>(dia.a, dia.b, md, cd, st);
#pragma unroll
                    for (int sd = 0; sd < 4; sd++) {
                        if (need[sd]) {
                            Seg<R> side = robot_side(A, r, sd);
                            R ms, cs;
                            if (CACHED) { ms = A.sm[r][sd]; cs = A.sc[r][sd]; }
                            else slope_yint<R>(side.a, side.b, ms, cs, st);
                            V2<R> q = intersect_mb<R>(ms, cs, side.a.x, md, cd, dia.a.x);
                            hit = hit | (within<R>(q, side, (R)0) & within<R>(q, dia, (R)0));
                        }
                    }
                }
            }
            RR_VOTE(m, l, hit);
        }
        for (int q = 0; q < C::VW / 2; q++)
            if ((m >> (2 * q)) & 3ull) pairs |= 1u << ((base >> 1) + q);
    }
    }
    if (CACHED) RR_STAMP(28);
#if RR_CARRY
    carry_done();
#endif
    if (pairs && !CACHED) { // rare: the responses read the cached inner-square offsets and side slopes
        RR_FOR_LANES(l) {
            if (l < C::NR) refresh_inner_lane(A, sp, l);
        }
        RR_SYNC();
        ensure_sides(A);
    }
    return pairs;
}
// balls_collided (RR_TrashyPhysics.py:72-73): one lane per ball pair
template <class C> RR_HD uint64_t detect_ball_pairs(Arena<C> &A) {
    using R = typename C::Real;
    if (C::NPB == 0) return 0;
    // one lane per ball i tests the later balls j > i; the reference's sqrt test only when d^2 <= 197 (14^2 + slack)
    uint64_t anyh = 0;
    RR_FOR_LANES(l) {
        bool h = false;
        if (l < C::NB) {
            int msk = 0;
            for (int j = 0; j < C::NB; j++) {
                V2<R> a = { A.p.bcx[l], A.p.bcy[l] }, b = { A.p.bcx[j], A.p.bcy[j] };
                R dx = b.x - a.x, dy = b.y - a.y;
                if (j > l && dx * dx + dy * dy <= (R)197) msk |= (dist<R>(a, b) <= (R)14) ? (1 << j) : 0;
            }
            A.bbm[l] = (uint16_t)msk;
            h = msk != 0;
        }
        RR_VOTE(anyh, l, h);
    }
    if (!RR_UNLIKELY(anyh)) return 0;
    RR_SYNC();
    uint64_t mask = 0; // bit p of the nested-loop pair order (i < j)
    int p = 0;
    for (int i = 0; i < C::NB; i++) {
        const int mi = A.bbm[i];
        for (int j = i + 1; j < C::NB; j++, p++)
            if ((mi >> j) & 1) mask |= 1ull << p;
    }
    return mask;
}
// A ball whose centre sits at x <= -900 is OUT OF PLAY: it touches nothing, walls included, and earns no ChasePosBall reward.
// The reference itself parks sprites "off map" at -1000 while it re-places them (RR_EnvBase.py:183-184); the opt-in goal-scoring
// mode (rr_extras.hpp: goal_step) parks a ball that a goal has consumed -- the reference's `sprBall.kill()` -- at
// (-1000 - 40 b, -1000) with zero velocity.  Far from everything, so every pair test is negative by itself; only the wall
// test and the reward need to know.
template <typename R> RR_HD R park_x(int b) { return (R)-1000 - (R)40 * (R)b; }
template <typename R> RR_HD R park_y() { return (R)-1000; }
template <class C> RR_HD bool ball_in_play(const Arena<C> &A, int b) { return A.p.bcx[b] > (typename C::Real)-900; }
// collided_wall on the int-truncated rect (RR_TrashyPhysics.py:76-85, RR_Ball.py:8-15)
template <class C> RR_HD bool ball_collided_wall(const Arena<C> &A, const SimParams<typename C::Real> &sp, int b) {
    // pygame.Rect holds C ints: truncation toward zero, which is what a float -> int32 conversion does (one instruction; a
    // 64-bit conversion is a dozen).  Ball coordinates stay within a few arena widths of the arena, far inside int32.
    const int L = (int)A.p.bl[b], T = (int)A.p.bt[b];
    const int Wd = (int)(A.p.brt[b] - A.p.bl[b]), Ht = (int)(A.p.bb[b] - A.p.bt[b]);
    return L < 0 || L + Wd > (int)sp.W || T < 0 || T + Ht > (int)sp.H;
}
template <class C> RR_HD uint32_t detect_ball_wall(const Arena<C> &A, const SimParams<typename C::Real> &sp) {
    uint64_t m = 0;
    RR_FOR_LANES(l) {
        bool hit = (l < C::NB) && ball_in_play(A, l < C::NB ? l : 0) && ball_collided_wall(A, sp, l);
        RR_VOTE(m, l, hit);
    }
    return (uint32_t)m;
}

// excursion bookkeeping for the island freeze (see substep): called after every position write of the contact paths
template <class C> RR_HD void ball_exc_update(Arena<C> &A, int b) {
    using R = typename C::Real;
    const R e = m_abs(A.p.bcx[b] - A.pfx[b]) + m_abs(A.p.bcy[b] - A.pfy[b]);
    if (e > A.exc[b]) A.exc[b] = e;
}
// ball-ball bound of the fused roll phase: the other ball may be seen before or after its own roll, and a roll moves it by
// at most |v| + |force| per axis (RR_Ball.py:78-105); +1 % and 0.02 px of slack.  Written at the frame hooks (force = 0) and
// again by whoever changes a ball's velocity or force before the roll (the push).
template <class C> RR_HD typename C::Real ball_reach(const Arena<C> &A, int b) {
    using R = typename C::Real;
    return (R)14.04 + ((m_abs(A.p.bvx[b]) + m_abs(A.bfx[b]) + m_abs(A.p.bvy[b]) + m_abs(A.bfy[b])) * (R)1.01 + (R)0.02);
}
struct Hit { uint32_t r, b; }; // robots / balls that took part in a hit (response, undo) during the sub-step

// ------------------------------------------------------------------------------------------------ contact responses (wave-uniform, list order)
// rectDblPriorFrame (RR_Robot.py:43-58): pose at the start of the robot's last move that was not undone.  Only the
// (rare) contact responses need its corners, so they are rebuilt here on demand exactly like the reference does
// (`rectPrior.rotation = rot` re-runs the rotation setter on (rot+720)%360).
template <typename R> struct PrevPose { R x, y, tlx, tly, trx, try_; }; // prior centre + TL / TR corner offsets (BR = -TL, BL = -TR)
template <class C>
RR_HD PrevPose<typename C::Real> robot_prev_frame(const Arena<C> &A, const SimParams<typename C::Real> &sp, int r, uint32_t bots_moved) {
    using R = typename C::Real;
    PrevPose<R> q;
    // rectPrior = rectDbl.copy(): centre c' = 10 + (cx - 10); then `rectPrior.center = (x, y)` moves it by (x - c')
    const R ccx = (R)10 + (A.p.rcx[r] - (R)10), ccy = (R)20 + (A.p.rcy[r] - (R)20);
    R rot;
    if (bots_moved & (1u << r)) { q.x = ccx + (A.ax[r] - ccx); q.y = ccy + (A.ay[r] - ccy); rot = A.arot[r]; }
    else if (!is_nan(A.p.px[r])) { q.x = ccx + (A.p.px[r] - ccx); q.y = ccy + (A.p.py[r] - ccy); rot = A.p.prot[r]; }
    else { q.x = ccx; q.y = ccy; rot = A.p.rrot[r]; }
    const R nr = norm360<R>(rot);
    if (nr == A.p.rrot[r]) {
        // same rotation value as the live rect (a robot driving straight, or one that has not turned since): the setter
        // returns early and the copy keeps the live corners -- which ARE corners_for(rrot): every writer of A.rel builds
        // it with that very expression -- so no trigonometry here.  The common case of a robot pushing a ball.
        q.tlx = RR_REL(A)[r][0]; q.tly = RR_REL(A)[r][1]; q.trx = RR_REL(A)[r][2]; q.try_ = RR_REL(A)[r][3];
    } else {
        R rel[8];
        corners_for<R>(nr, (R)10, (R)20, sp.rob_cdist, rel);
        q.tlx = rel[0]; q.tly = rel[1]; q.trx = rel[2]; q.try_ = rel[3];
    }
    return q;
}
template <typename R> RR_HD V2<R> prev_corner(const PrevPose<R> &q, int c) { // TL, TR, BL = -TR, BR = -TL
    const R ox = (c == TL) ? q.tlx : (c == TR) ? q.trx : (c == BL) ? -q.trx : -q.tlx;
    const R oy = (c == TL) ? q.tly : (c == TR) ? q.try_ : (c == BL) ? -q.try_ : -q.tly;
    V2<R> v = { q.x + ox, q.y + oy };
    return v;
}
template <class C> RR_HD void force_diameters(const Arena<C> &A, int r, V2<typename C::Real> bc, Seg<typename C::Real> dia[2]) {
    // (BL->TR) and (BR->TL), RR_TrashyPhysics.py:95-104
    const typename C::Real *q = A.u.irel[r];
    dia[0].a = { bc.x + q[2 * BL], bc.y + q[2 * BL + 1] }; dia[0].b = { bc.x + q[2 * TR], bc.y + q[2 * TR + 1] };
    dia[1].a = { bc.x + q[2 * BR], bc.y + q[2 * BR + 1] }; dia[1].b = { bc.x + q[2 * TL], bc.y + q[2 * TL + 1] };
}
template <typename R> RR_HD Seg<R> pick_dia(const Seg<R> dia[2], int d) { // by value: no dynamically indexed local array
    Seg<R> g = { { d ? dia[1].a.x : dia[0].a.x, d ? dia[1].a.y : dia[0].a.y }, { d ? dia[1].b.x : dia[0].b.x, d ? dia[1].b.y : dia[0].b.y } };
    return g;
}
// First (side, diameter) candidate -- sides in RIGHT,TOP,LEFT,BOTTOM order, diameters (BL->TR) then (BR->TL) -- whose
// intersection lies within the side and within the diameter grown by `buf`: the surface-contact search shared by
// apply_force_to_ball / bounce_ball_off_bot (RR_TrashyPhysics.py:110-115, :180-186).  The eight candidates are
// independent, so they are tested one per lane and the ballot's lowest set bit is the reference's first hit.
template <class C>
RR_HD int first_surface_hit(Arena<C> &A, int r, const Seg<typename C::Real> dia[2], typename C::Real buf) {
    using R = typename C::Real;
    ensure_sides(A);
    uint32_t hits = 0;
    for (int base = 0; base < 8; base += C::VW) {
        uint64_t m = 0;
        RR_FOR_LANES(l) {
            bool hit = false;
            const int t = base + l;
            if (t < 8) {
                const int sd = t >> 1, d = t & 1;
                int st = 0;
                Seg<R> side = robot_side(A, r, sd);
                const Seg<R> di = pick_dia<R>(dia, d);
                if (C::VW >= 8 || boxes_meet<R>(side, di, buf)) { // (the pre-test pays where a lane sweeps several candidates)
                    R md, cd;
                    slope_yint<R>(di.a, di.b, md, cd, st);
                    V2<R> I = intersect_mb<R>(A.sm[r][sd], A.sc[r][sd], side.a.x, md, cd, di.a.x); // cached side slope
                    hit = within<R>(I, side, (R)0) & within<R>(I, di, buf);
                }
            }
            RR_VOTE(m, l, hit);
        }
        hits |= (uint32_t)(m << base);
    }
    return hits ? low_bit(hits) : -1;
}
// The push runs apply_force_to_ball and bounce_ball_off_bot back to back on the same pair, and the first only writes the ball's
// force and mass: both searches look at the same eight intersections and differ in the diameter's buffer alone.  One sweep, two
// ballots: ka = first candidate under bufa, kb under bufb.
template <class C>
RR_HD void first_surface_hit2(Arena<C> &A, int r, const Seg<typename C::Real> dia[2], typename C::Real bufa, typename C::Real bufb,
                              int &ka, int &kb) {
    using R = typename C::Real;
    ensure_sides(A);
    uint32_t ha = 0, hb = 0;
    for (int base = 0; base < 8; base += C::VW) {
        uint64_t ma = 0, mb = 0;
        RR_FOR_LANES(l) {
            bool hita = false, hitb = false;
            const int t = base + l;
            if (t < 8) {
                const int sd = t >> 1, d = t & 1;
                int st = 0;
                Seg<R> side = robot_side(A, r, sd);
                const Seg<R> di = pick_dia<R>(dia, d);
                if (C::VW >= 8 || boxes_meet<R>(side, di, py_max<R>(bufa, bufb))) { // (the larger buffer's box contains the smaller one's)
                    R md, cd;
                    slope_yint<R>(di.a, di.b, md, cd, st);
                    V2<R> I = intersect_mb<R>(A.sm[r][sd], A.sc[r][sd], side.a.x, md, cd, di.a.x);
                    const bool ws = within<R>(I, side, (R)0);
                    hita = ws & within<R>(I, di, bufa);
                    hitb = ws & within<R>(I, di, bufb);
                }
            }
            RR_VOTE(ma, l, hita);
            RR_VOTE(mb, l, hitb);
        }
        ha |= (uint32_t)(ma << base); hb |= (uint32_t)(mb << base);
    }
    ka = ha ? low_bit(ha) : -1;
    kb = hb ? low_bit(hb) : -1;
}
// same idea for the corner-contact search: first corner (TL,TR,BL,BR) closer to the ball centre than `rad`
template <class C>
RR_HD int first_corner_hit(Arena<C> &A, int r, V2<typename C::Real> bc, typename C::Real rad) {
    using R = typename C::Real;
    uint32_t hits = 0;
    for (int base = 0; base < 4; base += C::VW) {
        uint64_t m = 0;
        RR_FOR_LANES(l) {
            const int t = base + l;
            bool hit = (t < 4) && (dist<R>(robot_corner(A, r, t < 4 ? t : 0), bc) < rad);
            RR_VOTE(m, l, hit);
        }
        hits |= (uint32_t)(m << base);
    }
    return hits ? low_bit(hits) : -1;
}
// apply_force_to_ball (RR_TrashyPhysics.py:88-152)
// (kpre: the surface search's result when the caller already has it, -2 = search here)
template <class C> RR_HDN void apply_force_to_ball(Arena<C> &A, const SimParams<typename C::Real> &sp, int r, int b, uint32_t bots_moved, int &st, int kpre = -2) {
    using R = typename C::Real;
    const R cbuf = (R).5;
    V2<R> bc = { A.p.bcx[b], A.p.bcy[b] }, rc = { A.p.rcx[r], A.p.rcy[r] };
    Seg<R> dia[2];
#if RR_CARRY
    force_diameters(A, r, carry_response(A, bc), dia); // `_rectBallInner.center = spr_ball.rectDbl.center` (:94), a relative move
#else
    force_diameters(A, r, bc, dia);
#endif
    R fx = A.bfx[b], fy = A.bfy[b];
    bool done = false;
    const int k = kpre != -2 ? kpre : first_surface_hit(A, r, dia, cbuf);
    if (k >= 0) {
        const int sd = k >> 1, d = k & 1;
        RR_TRACE("E force surface s=%d d=%d b=%d r=%d\n", sd, d, b, r);
        Seg<R> side = robot_side(A, r, sd);
        const Seg<R> di = pick_dia<R>(dia, d);
        R md, cd;
        slope_yint<R>(di.a, di.b, md, cd, st);
        V2<R> I = intersect_mb<R>(A.sm[r][sd], A.sc[r][sd], side.a.x, md, cd, di.a.x);
        R da = dist<R>(di.a, rc), db = dist<R>(di.b, rc);
        V2<R> cp = (da < db) ? di.a : di.b, opp = (da >= db) ? di.a : di.b;
        fx += (I.x - cp.x) + (opp.x - cp.x) * cbuf / (R)14;
        fy += (I.y - cp.y) + (opp.y - cp.y) * cbuf / (R)14;
        done = true;
    } else {
        const int c = first_corner_hit(A, r, bc, (R)7 + cbuf);
        if (c >= 0) {
            RR_TRACE("E force corner c=%d b=%d r=%d\n", c, b, r);
            const PrevPose<R> pv = robot_prev_frame(A, sp, r, bots_moved);
            V2<R> bcn = robot_corner(A, r, c);
            R dc = dist<R>(bcn, bc);
            V2<R> pc = prev_corner<R>(pv, c);
            V2<R> con = { bc.x - (bcn.x * (R)3 + pc.x) / (R)4, bc.y - (bcn.y * (R)3 + pc.y) / (R)4 };
            R cd = m_sqrt(con.x * con.x + con.y * con.y);
            R ex = ((R)7 - dc) * (R)1.2;
            fx += (con.x * ex / cd) + con.x * cbuf / cd;
            fy += (con.y * ex / cd) + con.y * cbuf / cd;
            done = true;
        }
    }
    if (done && RR_IS_LANE0) {
        A.bfx[b] = fx; A.bfy[b] = fy;
        if (A.bmass[b] < 2) A.bmass[b] = 2; // max(MASS_ROBOT, ball mass)
    }
    RR_SYNC();
}
template <typename R> RR_HD void bounce_reflect(V2<R> con, R &vx, R &vy, R &d2) { // RR_TrashyPhysics.py:192-204
    d2 = con.x * con.x + con.y * con.y;
    R term = ((con.x * vx) + (con.y * vy)) / d2;
    R prx = term * con.x, pry = term * con.y;
    if ((prx < (R)0 && con.x > (R)0) || (prx > (R)0 && con.x < (R)0)) vx = -prx * (R).8 * (R).8;
    if ((pry < (R)0 && con.y > (R)0) || (pry > (R)0 && con.y < (R)0)) vy = -pry * (R).8 * (R).8;
}
// bounce_ball_off_bot (RR_TrashyPhysics.py:155-245)
template <class C> RR_HDN void bounce_ball_off_bot(Arena<C> &A, const SimParams<typename C::Real> &sp, int r, int b, uint32_t bots_moved, int &st, int kpre = -2) {
    using R = typename C::Real;
    R vx = A.p.bvx[b], vy = A.p.bvy[b];
    if (vx == (R)0 && vy == (R)0) return;
    const R cbuf = (R).5;
    V2<R> bc = { A.p.bcx[b], A.p.bcy[b] };
    Seg<R> dia[2];
#if RR_CARRY
    force_diameters(A, r, carry_response(A, bc), dia); // (:165)
#else
    force_diameters(A, r, bc, dia);
#endif
    R mvx = 0, mvy = 0;
    bool done = false;
    RR_T0();
    const int k = kpre != -2 ? kpre : first_surface_hit(A, r, dia, (R)0);
    const int c = (k >= 0) ? -1 : first_corner_hit(A, r, bc, (R)7);
    RR_STAMP(23);
    if (k >= 0 || c >= 0) {
        const PrevPose<R> pv = robot_prev_frame(A, sp, r, bots_moved);
        RR_STAMP(24);
        if (k >= 0) {
            const int sd = k >> 1, d = k & 1;
            RR_TRACE("E bounce surface s=%d d=%d b=%d r=%d v=(%.17g,%.17g)\n", sd, d, b, r, (double)vx, (double)vy);
            Seg<R> side = robot_side(A, r, sd);
            const int ca = side_a(sd), cb = side_b(sd);
            Seg<R> sprev = { prev_corner<R>(pv, ca), prev_corner<R>(pv, cb) };
            const Seg<R> di = pick_dia<R>(dia, d);
            R md, cd, mp, cpv;
            slope_yint<R>(di.a, di.b, md, cd, st);
            slope_yint<R>(sprev.a, sprev.b, mp, cpv, st);
            V2<R> I = intersect_mb<R>(A.sm[r][sd], A.sc[r][sd], side.a.x, md, cd, di.a.x);
            V2<R> Ip = intersect_mb<R>(mp, cpv, sprev.a.x, md, cd, di.a.x);
            R da = dist<R>(di.a, Ip), db = dist<R>(di.b, Ip);
            RR_TRACE("E   I=(%.17g,%.17g) Ip=(%.17g,%.17g) da=%.17g db=%.17g prev=(%.17g,%.17g)\n", (double)I.x, (double)I.y, (double)Ip.x, (double)Ip.y, (double)da, (double)db, (double)pv.x, (double)pv.y);
            V2<R> cp = (da < db) ? di.a : di.b, opp = (da >= db) ? di.a : di.b;
            V2<R> con = { opp.x - cp.x, opp.y - cp.y };
            R d2;
            bounce_reflect<R>(con, vx, vy, d2);
            R sq = m_sqrt(d2);
            mvx = (I.x - cp.x) + con.x * cbuf / sq;
            mvy = (I.y - cp.y) + con.y * cbuf / sq;
        } else {
            RR_TRACE("E bounce corner c=%d b=%d r=%d v=(%.17g,%.17g)\n", c, b, r, (double)vx, (double)vy);
            V2<R> bcn = robot_corner(A, r, c);
            V2<R> pc = prev_corner<R>(pv, c);
            V2<R> con = { bc.x - (bcn.x * (R)3 + pc.x) / (R)4, bc.y - (bcn.y * (R)3 + pc.y) / (R)4 };
            R d2;
            bounce_reflect<R>(con, vx, vy, d2);
            R ex = dist<R>(pc, bc), cd = m_sqrt(d2);
            mvx = con.x * ex / cd;
            mvy = con.y * ex / cd;
        }
        done = true;
    }
    if (done && RR_IS_LANE0) {
        A.p.bvx[b] = vx; A.p.bvy[b] = vy;
        // `centerx += v` goes through the setter: the applied delta is (cx + v) - cx
        R nx = bc.x + mvx;
        ball_shift(A, b, nx - bc.x, (R)0);
        R ny = bc.y + mvy;
        ball_shift(A, b, (R)0, ny - bc.y);
        ball_exc_update(A, b);
    }
    RR_SYNC();
    RR_STAMP(25);
}
// ------------------------------------------------------------------------------------------------ lane pairs
// A contact response is a chain of a few hundred dependent fp64 instructions, and one wavefront alone on its SIMD issues an fp64
// instruction every 8 cycles however many of its lanes are active: a stuck arena's step is that instruction count (profiles/r03/
// stuck_arena_phases_*.txt).  Most of a response is the same arithmetic on an x and on a y component, and where it is not it is two
// independent chains (the slope of the diameter | of the prior-frame side, the intersection with the side | with the prior-frame
// side, the distance of the intersection from one end point | from the other).  So a response runs on a PAIR of lanes -- the even
// lane takes the x component / the first chain, the odd lane the y component / the second -- and the two swap values through DPP
// (RR_XOR1's instruction): the same operations on the same operands in the same order as RR_TrashyPhysics.py:155-245, about
// half the instructions.  PV<R> is "one value per lane of the pair": ONE register on the GPU; the host emulation carries both lanes'
// values and runs a pair phase once per pair (RR_FOR_PAIRS), so the very same source is checked against the oracle on the CPU.
#if RR_GPU
#define RR_PV_N 1
#define RR_FOR_PAIRS(l) RR_FOR_LANES(l)
#define RR_PV_ROLE(l, i) ((l) & 1)
#define RR_PVLOAD(name, l) (rr::PV<R>{ { name } })
#define RR_PVSTORE(name, l, val) (name) = (val).v[0]
#else
#define RR_PV_N 2
#define RR_FOR_PAIRS(l) for (int l = 0; l < C::VW; l += 2)
#define RR_PV_ROLE(l, i) (i)
#define RR_PVLOAD(name, l) (rr::PV<R>{ { name[l], name[(l) + 1] } })
#define RR_PVSTORE(name, l, val) do { name[l] = (val).v[0]; name[(l) + 1] = (val).v[1]; } while (0)
#endif
#define RR_PV_EACH(i) for (int i = 0; i < RR_PV_N; i++)
template <typename R> struct PV { R v[RR_PV_N]; };
// the partner lane's value
template <typename R> RR_HD PV<R> pv_swap(PV<R> a) {
#if RR_GPU
    PV<R> r = { { lane_xor1(a.v[0]) } };
#else
    PV<R> r = { { a.v[1], a.v[0] } };
#endif
    return r;
}
// the even (role 0) / odd (role 1) lane's value on both lanes
template <typename R> RR_HD PV<R> pv_from0(PV<R> a, int l) {
#if RR_GPU
    const R o = lane_xor1(a.v[0]);
    PV<R> r = { { (l & 1) ? o : a.v[0] } };
#else
    (void)l;
    PV<R> r = { { a.v[0], a.v[0] } };
#endif
    return r;
}
template <typename R> RR_HD PV<R> pv_from1(PV<R> a, int l) {
#if RR_GPU
    const R o = lane_xor1(a.v[0]);
    PV<R> r = { { (l & 1) ? a.v[0] : o } };
#else
    (void)l;
    PV<R> r = { { a.v[1], a.v[1] } };
#endif
    return r;
}
// element p of a pair of arrays that sit `stride` apart (x fields and y fields of the record are declared back to back): the lane of role p
template <typename R> RR_HD PV<R> pv_ld2(const R *base, int stride, int idx, int l) {
#if RR_GPU
    PV<R> r = { { base[(l & 1) * stride + idx] } };
#else
    (void)l;
    PV<R> r = { { base[idx], base[stride + idx] } };
#endif
    return r;
}
template <typename R> RR_HD void pv_st2(R *base, int stride, int idx, int l, PV<R> val) {
#if RR_GPU
    base[(l & 1) * stride + idx] = val.v[0];
#else
    (void)l;
    base[idx] = val.v[0]; base[stride + idx] = val.v[1];
#endif
}

// ---- concurrent ball-robot bounces of one resolve pass (RR_EnvBase.py:380-384: `for ball, bot in collisions: bounce_ball_off_bot`)
// The hit list is ball-major and a bounce reads the robot (which stands still inside the resolve loop) and reads / writes ITS
// ball only: bounces of DIFFERENT balls are independent, bounces of one ball keep their list order.  So every ball with a hit gets
// a SLOT of G = 8, 4 or 2 lanes of its arena (the more balls, the narrower) and walks its own robots in order, all slots side
// by side; a slot keeps its ball in registers across its bounces (component p of centre / edges / velocity in the lanes of role p)
// and writes it back once.  Inside a slot the surface search spreads its eight (side, diameter) candidates over the G lanes -- the
// slot's share of the arena's ballot, lowest candidate first = the reference's first hit -- and the response runs on lane pairs.
// Parity build: the scratch rect makes every response depend on the previous one (carry_response), so the caller hands over one
// pair at a time; the response arithmetic is shared.
template <class C> struct PairFields {
    using R = typename C::Real;
    using P = typename ArenaBody<C>::P;
    static constexpr int NR = C::NR, NB = C::NB;
    static_assert(offsetof(P, rcy) == offsetof(P, rcx) + NR * sizeof(R), "rcx, rcy back to back");
    static_assert(offsetof(P, py) == offsetof(P, px) + NR * sizeof(R), "px, py back to back");
    static_assert(offsetof(P, bcy) == offsetof(P, bcx) + NB * sizeof(R), "bcx, bcy back to back");
    static_assert(offsetof(P, bt) == offsetof(P, bl) + 2 * NB * sizeof(R) && offsetof(P, bb) == offsetof(P, brt) + 2 * NB * sizeof(R), "bl brt bt bb");
    static_assert(offsetof(P, bvy) == offsetof(P, bvx) + NB * sizeof(R), "bvx, bvy back to back");
    static_assert(offsetof(ArenaBody<C>, ay) == offsetof(ArenaBody<C>, ax) + NR * sizeof(R), "ax, ay back to back");
    static_assert(offsetof(ArenaBody<C>, pfy) == offsetof(ArenaBody<C>, pfx) + NB * sizeof(R), "pfx, pfy back to back");
};
// RR_TrashyPhysics.py:192-204 on a lane pair: con / v hold component p; returns d2 on both lanes
template <typename R> RR_HD PV<R> pv_bounce_reflect(PV<R> con, PV<R> &v) {
    PV<R> sq, pr, d2, num;
    RR_PV_EACH(i) { sq.v[i] = con.v[i] * con.v[i]; pr.v[i] = con.v[i] * v.v[i]; }
    const PV<R> sqo = pv_swap(sq), pro = pv_swap(pr);
    RR_PV_EACH(i) {
        d2.v[i] = sq.v[i] + sqo.v[i];   // con.x^2 + con.y^2 (IEEE addition commutes: both lanes hold the same bits)
        num.v[i] = pr.v[i] + pro.v[i];  // con.x vx + con.y vy
        const R term = num.v[i] / d2.v[i];
        const R prj = term * con.v[i];
        if ((prj < (R)0 && con.v[i] > (R)0) || (prj > (R)0 && con.v[i] < (R)0)) v.v[i] = -prj * (R).8 * (R).8;
    }
    return d2;
}
// rectDblPriorFrame of robot r on a lane pair (see robot_prev_frame): centre component p, and component p of the TL / TR corner offsets
template <class C>
RR_HD void pv_robot_prev_frame(const Arena<C> &A, const SimParams<typename C::Real> &sp, int r, uint32_t bots_moved, int l,
                               PV<typename C::Real> &q, PV<typename C::Real> &tl, PV<typename C::Real> &tr) {
    using R = typename C::Real;
    const PV<R> rc = pv_ld2<R>(&A.p.rcx[0], C::NR, r, l);
    PV<R> cc;
    RR_PV_EACH(i) { const R K = RR_PV_ROLE(l, i) ? (R)20 : (R)10; cc.v[i] = K + (rc.v[i] - K); }
    R rot;
    if (bots_moved & (1u << r)) {
        const PV<R> a = pv_ld2<R>(&A.ax[0], C::NR, r, l);
        RR_PV_EACH(i) q.v[i] = cc.v[i] + (a.v[i] - cc.v[i]);
        rot = A.arot[r];
    } else if (!is_nan(A.p.px[r])) {
        const PV<R> a = pv_ld2<R>(&A.p.px[0], C::NR, r, l);
        RR_PV_EACH(i) q.v[i] = cc.v[i] + (a.v[i] - cc.v[i]);
        rot = A.p.prot[r];
    } else {
        q = cc;
        rot = A.p.rrot[r];
    }
    const R nr = norm360<R>(rot);
    if (nr == A.p.rrot[r]) {
        tl = pv_ld2<R>(&RR_REL(A)[r][0], 1, 0, l); tr = pv_ld2<R>(&RR_REL(A)[r][2], 1, 0, l);
    } else {
        R rel[8];
        corners_for<R>(nr, (R)10, (R)20, sp.rob_cdist, rel);
        RR_PV_EACH(i) { const int p = RR_PV_ROLE(l, i); tl.v[i] = p ? rel[1] : rel[0]; tr.v[i] = p ? rel[3] : rel[2]; }
    }
}
template <typename R> RR_HD R pv_prev_off(int c, R tl, R tr) { return (c == TL) ? tl : (c == TR) ? tr : (c == BL) ? -tr : -tl; } // BR = -TL, BL = -TR
#ifndef RR_NO_CBR
#define RR_CBR 1
#else
#define RR_CBR 0 // A/B builds only: the resolve pass applies its ball-robot bounces one after the other (bounce_ball_off_bot)
#endif
template <class C>
RR_HDN void bounce_pass(Arena<C> &A, const SimParams<typename C::Real> &sp, uint32_t br, uint32_t bots_moved, int &st) {
    using R = typename C::Real;
    constexpr int NR = C::NR, NB = C::NB, VW = C::VW;
    static_assert(VW >= 2 && (VW & 1) == 0, "lane pairs");
    (void)sizeof(PairFields<C>);
    constexpr uint32_t RM = (1u << NR) - 1u;
    constexpr int MAXS = NB < VW / 2 ? NB : VW / 2; // slots of one round
    uint32_t hb = 0;
    for (int b = 0; b < NB; b++) hb |= ((br >> (b * NR)) & RM) ? (1u << b) : 0u;
    RR_LANE_VAR(int, s_b);        // the slot's ball (-1: the lane has no slot)
    RR_LANE_VAR(uint32_t, s_rm);  // its robots still to bounce off, list order = ascending
    RR_LANE_VAR(R, s_c); RR_LANE_VAR(R, s_lo); RR_LANE_VAR(R, s_hi); RR_LANE_VAR(R, s_v); // component p of centre / low edge / high edge / velocity
    RR_LANE_VAR(R, s_exc);
    RR_LANE_VAR(R, s_dc);         // component p of the centre the diameters are built from (parity build: the carried scratch rect)
    RR_LANE_VAR(int, s_go); RR_LANE_VAR(int, s_k); RR_LANE_VAR(int, s_cn); RR_LANE_VAR(int, s_st);
    RR_LANE_VAR(R, s_md); RR_LANE_VAR(R, s_cd); // slope / intercept of this lane's diameter (its candidates share it: the lane's d is fixed)
    RR_T0();
#pragma unroll 1
    while (hb) { // one round, unless more balls were hit than the arena has lane pairs
        const int nhb = __builtin_popcount(hb);
        const int lgG = (VW >= 8 && nhb * 8 <= VW) ? 3 : (VW >= 4 && nhb * 4 <= VW) ? 2 : 1;
        const int G = 1 << lgG;
        const int nsl = nhb < (VW >> lgG) ? nhb : (VW >> lgG);
        uint32_t rest = hb;
        for (int s = 0; s < MAXS; s++) rest = (s < nsl) ? (rest & (rest - 1)) : rest;
        RR_FOR_LANES(l) {
            const int slot = l >> lgG, p = l & 1;
            uint32_t m = hb;
            for (int s = 0; s < MAXS; s++) m = (s < slot) ? (m & (m - 1)) : m;
            const bool on = slot < nsl;
            const int b = on ? low_bit(m | (1u << 31)) : 0;
            RR_LV(s_b, l) = on ? b : -1;
            RR_LV(s_rm, l) = on ? ((br >> (b * NR)) & RM) : 0u;
            RR_LV(s_c, l) = (&A.p.bcx[0])[p * NB + b];
            RR_LV(s_lo, l) = (&A.p.bl[0])[p * 2 * NB + b];
            RR_LV(s_hi, l) = (&A.p.brt[0])[p * 2 * NB + b];
            RR_LV(s_v, l) = (&A.p.bvx[0])[p * NB + b];
            RR_LV(s_exc, l) = A.exc[b];
            RR_LV(s_st, l) = 0;
        }
        hb = rest;
#pragma unroll 1
        for (;;) {
            uint64_t any = 0;
            RR_FOR_LANES(l) { RR_VOTE(any, l, RR_LV(s_rm, l) != 0u); }
            if (!any) break;
            // ---- this bounce: a no-op for a ball at rest (:156); else where do the diameters sit
            RR_FOR_LANES(l) {
                const R vo = RR_XOR1(s_v, l);
                RR_LV(s_go, l) = (RR_LV(s_rm, l) != 0u && !(RR_LV(s_v, l) == (R)0 && vo == (R)0)) ? 1 : 0;
                RR_LV(s_k, l) = -1; RR_LV(s_cn, l) = -1;
#if !RR_CARRY
                RR_LV(s_dc, l) = RR_LV(s_c, l);
#endif
            }
#if RR_CARRY
            // `_rectBallInner.center = ball.center` (:165) is a relative move of the module-global rect (see "scratch-rect carry")
            RR_FOR_LANES(l) {
                if (RR_LV(s_go, l)) RR_LV(s_dc, l) = carry1<R>(A.p.ic[l & 1], RR_LV(s_c, l));
            }
            RR_SYNC();
            RR_FOR_LANES(l) {
                if (RR_LV(s_go, l) && (l & (G - 1)) < 2) A.p.ic[l & 1] = RR_LV(s_dc, l);
            }
            RR_SYNC();
#endif
            // ---- surface search (:180-186): candidate t = (side t >> 1, diameter t & 1), t = sub + G c in round c
            RR_FOR_LANES(l) {
                const R dco = RR_XOR1(s_dc, l);
                if (RR_LV(s_go, l)) {
                    const int r = low_bit(RR_LV(s_rm, l)), d = l & 1;
                    const R dcx = (l & 1) ? dco : RR_LV(s_dc, l), dcy = (l & 1) ? RR_LV(s_dc, l) : dco;
                    const R *q = A.u.irel[r];
                    // (BL -> TR) for d = 0, (BR -> TL) for d = 1 (RR_TrashyPhysics.py:95-104)
                    const V2<R> da = { dcx + q[2 * (d ? BR : BL)], dcy + q[2 * (d ? BR : BL) + 1] }, db = { dcx + q[2 * (d ? TL : TR)], dcy + q[2 * (d ? TL : TR) + 1] };
                    int lst = 0;
                    slope_yint<R>(da, db, RR_LV(s_md, l), RR_LV(s_cd, l), lst);
                    RR_LV(s_st, l) |= lst;
                }
            }
#pragma unroll 1
            for (int c = 0; c < (8 >> lgG); c++) {
                uint64_t m = 0;
                RR_FOR_LANES(l) {
                    bool hit = false;
                    const R dco = RR_XOR1(s_dc, l);
                    if (RR_LV(s_go, l) && RR_LV(s_k, l) < 0) {
                        const int r = low_bit(RR_LV(s_rm, l)), t = (l & (G - 1)) + (c << lgG), sd = t >> 1, d = t & 1;
                        const R dcx = (l & 1) ? dco : RR_LV(s_dc, l), dcy = (l & 1) ? RR_LV(s_dc, l) : dco;
                        const R *q = A.u.irel[r];
                        const Seg<R> di = { { dcx + q[2 * (d ? BR : BL)], dcy + q[2 * (d ? BR : BL) + 1] }, { dcx + q[2 * (d ? TL : TR)], dcy + q[2 * (d ? TL : TR) + 1] } };
                        const Seg<R> side = robot_side(A, r, sd);
                        if (lgG == 3 || boxes_meet<R>(side, di, (R)0)) { // (the pre-test pays where a lane sweeps several candidates)
                            const V2<R> I = intersect_mb<R>(A.sm[r][sd], A.sc[r][sd], side.a.x, RR_LV(s_md, l), RR_LV(s_cd, l), di.a.x);
                            hit = within<R>(I, side, (R)0) & within<R>(I, di, (R)0);
                        }
                    }
                    RR_VOTE(m, l, hit);
                }
                RR_FOR_LANES(l) {
                    const uint32_t bits = (uint32_t)(m >> (l & ~(G - 1))) & ((1u << G) - 1u); // this slot's share of the ballot
                    if (RR_LV(s_go, l) && RR_LV(s_k, l) < 0 && bits) RR_LV(s_k, l) = (c << lgG) + low_bit(bits);
                }
            }
            // ---- corner search (:219): first corner (TL, TR, BL, BR) closer to the ball centre than 7
            uint64_t needc = 0;
            RR_FOR_LANES(l) { RR_VOTE(needc, l, RR_LV(s_go, l) && RR_LV(s_k, l) < 0); }
            if (RR_UNLIKELY(needc)) {
                for (int c = 0; c < (G >= 4 ? 1 : 2); c++) {
                    uint64_t m = 0;
                    RR_FOR_LANES(l) {
                        bool hit = false;
                        const R co = RR_XOR1(s_c, l);
                        const int t = (l & (G - 1)) + (c << lgG);
                        if (RR_LV(s_go, l) && RR_LV(s_k, l) < 0 && RR_LV(s_cn, l) < 0 && t < 4) {
                            const int r = low_bit(RR_LV(s_rm, l));
                            const V2<R> bc = { (l & 1) ? co : RR_LV(s_c, l), (l & 1) ? RR_LV(s_c, l) : co };
                            hit = dist<R>(robot_corner(A, r, t), bc) < (R)7;
                        }
                        RR_VOTE(m, l, hit);
                    }
                    RR_FOR_LANES(l) {
                        const uint32_t bits = (uint32_t)(m >> (l & ~(G - 1))) & ((1u << G) - 1u);
                        if (RR_LV(s_go, l) && RR_LV(s_k, l) < 0 && RR_LV(s_cn, l) < 0 && bits) RR_LV(s_cn, l) = (c << lgG) + low_bit(bits);
                    }
                }
            }
            RR_STAMP(23);
            // ---- the response, on lane pairs
            RR_FOR_PAIRS(l) {
                const int k = RR_LV(s_k, l), cn = RR_LV(s_cn, l);
                PV<R> c0 = RR_PVLOAD(s_c, l), v = RR_PVLOAD(s_v, l);
                const PV<R> dc = RR_PVLOAD(s_dc, l);
                const PV<R> dco = pv_swap(dc);
                if (RR_LV(s_go, l) && (k >= 0 || cn >= 0)) {
                    const int r = low_bit(RR_LV(s_rm, l)), b = RR_LV(s_b, l);
                    PV<R> pvq, ptl, ptr;
                    pv_robot_prev_frame(A, sp, r, bots_moved, l, pvq, ptl, ptr);
                    const PV<R> pvqo = pv_swap(pvq), ptlo = pv_swap(ptl), ptro = pv_swap(ptr);
                    PV<R> mv;
                    if (k >= 0) {
                        const int sd = k >> 1, d = k & 1, ca = side_a(sd), cb = side_b(sd);
                        if ((l & (G - 1)) == 0) RR_TRACE("E bounce surface s=%d d=%d b=%d r=%d\n", sd, d, b, r);
                        // role 0: the diameter (di.a, di.b); role 1: the prior-frame side (sprev.a, sprev.b) -- each lane its segment's slope
                        PV<R> sax, say, sbx, sby, m, cc;
                        RR_PV_EACH(i) {
                            const int p = RR_PV_ROLE(l, i);
                            const R dcx = p ? dco.v[i] : dc.v[i], dcy = p ? dc.v[i] : dco.v[i];
                            const R qx = p ? pvqo.v[i] : pvq.v[i], qy = p ? pvq.v[i] : pvqo.v[i];
                            const R tlx = p ? ptlo.v[i] : ptl.v[i], tly = p ? ptl.v[i] : ptlo.v[i], trx = p ? ptro.v[i] : ptr.v[i], try_ = p ? ptr.v[i] : ptro.v[i];
                            const R *q = A.u.irel[r];
                            const int ea = d ? BR : BL, eb = d ? TL : TR;
                            sax.v[i] = p ? qx + pv_prev_off<R>(ca, tlx, trx) : dcx + q[2 * ea];
                            say.v[i] = p ? qy + pv_prev_off<R>(ca, tly, try_) : dcy + q[2 * ea + 1];
                            sbx.v[i] = p ? qx + pv_prev_off<R>(cb, tlx, trx) : dcx + q[2 * eb];
                            sby.v[i] = p ? qy + pv_prev_off<R>(cb, tly, try_) : dcy + q[2 * eb + 1];
                            int lst = 0;
                            const V2<R> a = { sax.v[i], say.v[i] }, bb_ = { sbx.v[i], sby.v[i] };
                            slope_yint<R>(a, bb_, m.v[i], cc.v[i], lst);
                            if (lst) RR_LV(s_st, l) |= lst;
                        }
                        const PV<R> md = pv_from0(m, l), cd = pv_from0(cc, l), dax = pv_from0(sax, l), day = pv_from0(say, l),
                                    dbx = pv_from0(sbx, l), dby = pv_from0(sby, l);
                        // role 0: I = side x diameter; role 1: Ip = prior-frame side x diameter
                        PV<R> ix, iy;
                        const Seg<R> side = robot_side(A, r, sd);
                        RR_PV_EACH(i) {
                            const int p = RR_PV_ROLE(l, i);
                            const R m1 = p ? m.v[i] : A.sm[r][sd], b1 = p ? cc.v[i] : A.sc[r][sd], x1 = p ? sax.v[i] : side.a.x;
                            const V2<R> I = intersect_mb<R>(m1, b1, x1, md.v[i], cd.v[i], dax.v[i]);
                            ix.v[i] = I.x; iy.v[i] = I.y;
                        }
                        const PV<R> ipx = pv_from1(ix, l), ipy = pv_from1(iy, l), Ix = pv_from0(ix, l), Iy = pv_from0(iy, l);
                        // role 0: da = dist(di.a, Ip); role 1: db = dist(di.b, Ip)
                        PV<R> dd;
                        RR_PV_EACH(i) {
                            const int p = RR_PV_ROLE(l, i);
                            const V2<R> e = { p ? dbx.v[i] : dax.v[i], p ? dby.v[i] : day.v[i] }, ip = { ipx.v[i], ipy.v[i] };
                            dd.v[i] = dist<R>(e, ip);
                        }
                        const PV<R> da = pv_from0(dd, l), db = pv_from1(dd, l);
                        PV<R> cp, con;
                        RR_PV_EACH(i) {
                            const int p = RR_PV_ROLE(l, i);
                            const R ea_ = p ? day.v[i] : dax.v[i], eb_ = p ? dby.v[i] : dbx.v[i]; // component p of di.a / di.b
                            cp.v[i] = (da.v[i] < db.v[i]) ? ea_ : eb_;
                            const R opp = (da.v[i] >= db.v[i]) ? ea_ : eb_;
                            con.v[i] = opp - cp.v[i];
                        }
                        const PV<R> d2 = pv_bounce_reflect<R>(con, v);
                        RR_PV_EACH(i) {
                            const int p = RR_PV_ROLE(l, i);
                            const R sq = m_sqrt(d2.v[i]);
                            mv.v[i] = ((p ? Iy.v[i] : Ix.v[i]) - cp.v[i]) + con.v[i] * (R).5 / sq;
                        }
                    } else {
                        if ((l & (G - 1)) == 0) RR_TRACE("E bounce corner c=%d b=%d r=%d\n", cn, b, r);
                        const PV<R> rc = pv_ld2<R>(&A.p.rcx[0], NR, r, l), ro = pv_ld2<R>(&RR_REL(A)[r][2 * cn], 1, 0, l);
                        PV<R> con, pc, sq;
                        RR_PV_EACH(i) {
                            const R bcn = rc.v[i] + ro.v[i];
                            pc.v[i] = pvq.v[i] + pv_prev_off<R>(cn, ptl.v[i], ptr.v[i]);
                            con.v[i] = c0.v[i] - (bcn * (R)3 + pc.v[i]) / (R)4;
                        }
                        const PV<R> d2 = pv_bounce_reflect<R>(con, v);
                        RR_PV_EACH(i) { const R df = c0.v[i] - pc.v[i]; sq.v[i] = df * df; }
                        const PV<R> sqo = pv_swap(sq);
                        RR_PV_EACH(i) {
                            const R ex = m_sqrt(sq.v[i] + sqo.v[i]), cd = m_sqrt(d2.v[i]);
                            mv.v[i] = con.v[i] * ex / cd;
                        }
                    }
                    // `centerx += v` goes through the setter: the applied delta is (c + v) - c; the other axis' setter call adds 0 to this
                    // axis' fields -- after this axis' own delta for x, before it for y (RR_TrashyPhysics.py:242-243, MyUtils.py:141-148)
                    PV<R> lo = RR_PVLOAD(s_lo, l), hi = RR_PVLOAD(s_hi, l), exc = RR_PVLOAD(s_exc, l), ab;
                    const PV<R> pf = pv_ld2<R>(&A.pfx[0], NB, b, l);
                    RR_PV_EACH(i) {
                        const int p = RR_PV_ROLE(l, i);
                        const R n = c0.v[i] + mv.v[i], dl = n - c0.v[i];
                        const R d1 = p ? (R)0 : dl, d2_ = p ? dl : (R)0;
                        c0.v[i] = (c0.v[i] + d1) + d2_; lo.v[i] = (lo.v[i] + d1) + d2_; hi.v[i] = (hi.v[i] + d1) + d2_;
                        ab.v[i] = m_abs(c0.v[i] - pf.v[i]);
                    }
                    const PV<R> abo = pv_swap(ab);
                    RR_PV_EACH(i) { const R e = ab.v[i] + abo.v[i]; if (e > exc.v[i]) exc.v[i] = e; }
                    RR_PVSTORE(s_c, l, c0); RR_PVSTORE(s_lo, l, lo); RR_PVSTORE(s_hi, l, hi); RR_PVSTORE(s_v, l, v); RR_PVSTORE(s_exc, l, exc);
                }
            }
            RR_FOR_LANES(l) { RR_LV(s_rm, l) &= RR_LV(s_rm, l) - 1u; }
            RR_STAMP(25);
        }
        // ---- the slots' balls go back to the arena: the first pair of a slot writes, role p its component's fields
        uint64_t anyst = 0;
        RR_FOR_LANES(l) {
            const int b = RR_LV(s_b, l), p = l & 1;
            if (b >= 0 && (l & (G - 1)) < 2) {
                (&A.p.bcx[0])[p * NB + b] = RR_LV(s_c, l);
                (&A.p.bl[0])[p * 2 * NB + b] = RR_LV(s_lo, l);
                (&A.p.brt[0])[p * 2 * NB + b] = RR_LV(s_hi, l);
                (&A.p.bvx[0])[p * NB + b] = RR_LV(s_v, l);
                if (p == 0) A.exc[b] = RR_LV(s_exc, l);
            }
            RR_VOTE(anyst, l, (RR_LV(s_st, l) & ST_DIV0) != 0);
        }
        if (anyst) st |= ST_DIV0;
        RR_SYNC();
    }
}
// bounce_balls (RR_TrashyPhysics.py:248-316)
template <class C> RR_HDN void bounce_balls(Arena<C> &A, int i, int j, int &st) {
    using R = typename C::Real;
    R x1 = A.p.bcx[i], y1 = A.p.bcy[i], x2 = A.p.bcx[j], y2 = A.p.bcy[j];
    if (x1 == x2 && y1 == y2) { st |= ST_SAME_SPOT; return; }
    R vxx = x2 - x1, vyy = y2 - y1;
    R d12 = m_sqrt(vxx * vxx + vyy * vyy);
    R rx = vxx * (R)7 / d12, ry = vyy * (R)7 / d12;
    R p1x = x1 + rx, p1y = y1 + ry, p2x = x2 - rx, p2y = y2 - ry;
    const R buffer = (R)1.1;
    R hx = (p2x - p1x) / (R)2, hy = (p2y - p1y) / (R)2;
    int m1 = A.bmass[i], m2 = A.bmass[j];
    R n1x = x1, n1y = y1, n2x = x2, n2y = y2;
    if (m1 == m2) {
        n1x = x1 + hx * buffer; n1y = y1 + hy * buffer;
        n2x = x2 - hx * buffer; n2y = y2 - hy * buffer;
    } else if (m1 > m2) {
        n2x = x2 + (p1x - p2x) * buffer; n2y = y2 + (p1y - p2y) * buffer;
        m2 = m1;
    } else {
        n1x = x1 + (p2x - p1x) * buffer; n1y = y1 + (p2y - p1y) * buffer;
        m1 = m2;
    }
    // centre setters are incremental: new centre = c + (n - c)
    R c1x = x1 + (n1x - x1), c1y = y1 + (n1y - y1), c2x = x2 + (n2x - x2), c2y = y2 + (n2y - y2);
    R ax = c2x - c1x, ay = c2y - c1y;          // tplVect1to2
    R bx = ax * (R)-1, by = ay * (R)-1;          // tplVect2to1
    R d2 = ax * ax + ay * ay;
    R v1x = A.p.bvx[i], v1y = A.p.bvy[i], v2x = A.p.bvx[j], v2y = A.p.bvy[j];
    R t1 = div0<R>(ax * v1x + ay * v1y, d2, st);
    R t2 = div0<R>(bx * v2x + by * v2y, d2, st);
    R dfx = t1 * ax - t2 * bx, dfy = t1 * ay - t2 * by;
    v1x -= dfx * (R).995; v1y -= dfy * (R).995;
    v2x += dfx * (R).995; v2y += dfy * (R).995;
    R f1x = A.bfx[i], f1y = A.bfy[i], f2x = A.bfx[j], f2y = A.bfy[j];
    if (f1x > (R)0) v1x = py_max<R>(v1x, f1x); else if (f1x < (R)0) v1x = py_min<R>(v1x, f1x);
    if (f1y > (R)0) v1y = py_max<R>(v1y, f1y); else if (f1y < (R)0) v1y = py_min<R>(v1y, f1y);
    if (f2x > (R)0) v2x = py_max<R>(v2x, f2x); else if (f2x < (R)0) v2x = py_min<R>(v2x, f2x);
    if (f2y > (R)0) v2y = py_max<R>(v2y, f2y); else if (f2y < (R)0) v2y = py_min<R>(v2y, f2y);
    if (RR_IS_LANE0) {
        // an unmoved ball gets a zero delta, which leaves its rect bit-identical
        ball_shift(A, i, n1x - x1, (R)0);
        ball_shift(A, i, (R)0, n1y - y1);
        ball_shift(A, j, n2x - x2, (R)0);
        ball_shift(A, j, (R)0, n2y - y2);
        A.bmass[i] = m1; A.bmass[j] = m2;
        A.p.bvx[i] = v1x; A.p.bvy[i] = v1y; A.p.bvx[j] = v2x; A.p.bvy[j] = v2y;
        ball_exc_update(A, i); ball_exc_update(A, j);
    }
    RR_SYNC();
}
// The ball-ball bounces of one resolve pass (RR_EnvBase.py:373-378: `for ball1, ball2 in collisions: bounce_balls`), on lane pairs.
// A bounce reads and writes its two balls only, so consecutive bounces of the (frozen, ordered) hit list that share no ball are
// independent: the list is cut into maximal runs of mutually disjoint pairs, a run's bounces go side by side -- one lane pair
// each -- and the runs follow each other in list order.  Inside a bounce the even lane holds the x components, the odd lane the y
// components (bounce_balls above, operation for operation; the two projections t1 | t2 are one division on either lane).
template <class C> RR_HDN void bounce_balls_pass(Arena<C> &A, uint64_t bbm, int &st) {
    using R = typename C::Real;
    constexpr int NB = C::NB, VW = C::VW;
    (void)sizeof(PairFields<C>);
    RR_LANE_VAR(int, s_i); RR_LANE_VAR(int, s_j); RR_LANE_VAR(int, s_st);
#pragma unroll 1
    while (bbm) {
        // the next run: pairs in list order until one shares a ball with the run (or the lane pairs are used up)
        uint32_t used = 0;
        int nrun = 0;
        uint64_t todo = bbm;
        int ri[VW / 2], rj[VW / 2];
        for (int q = 0; q < VW / 2; q++) { ri[q] = -1; rj[q] = -1; }
#pragma unroll
        for (int q = 0; q < VW / 2; q++) {
            if (todo && nrun == q) {
                int i, j;
                pair_of<C>(low_bit(todo), NB, i, j);
                const uint32_t two = (1u << i) | (1u << j);
                if (!(used & two)) { used |= two; ri[q] = i; rj[q] = j; nrun = q + 1; todo &= todo - 1; }
            }
        }
        bbm = todo;
        RR_FOR_LANES(l) {
            int i = -1, j = -1;
#pragma unroll
            for (int q = 0; q < VW / 2; q++) { i = ((l >> 1) == q) ? ri[q] : i; j = ((l >> 1) == q) ? rj[q] : j; } // by value: no dynamically indexed local array
            RR_LV(s_i, l) = i; RR_LV(s_j, l) = j; RR_LV(s_st, l) = 0;
        }
        RR_FOR_PAIRS(l) {
            const int i = RR_LV(s_i, l), j = RR_LV(s_j, l);
            const int bi = i < 0 ? 0 : i, bj = j < 0 ? 0 : j;
            const PV<R> c1 = pv_ld2<R>(&A.p.bcx[0], NB, bi, l), c2 = pv_ld2<R>(&A.p.bcx[0], NB, bj, l);
            PV<R> eq;
            RR_PV_EACH(k) eq.v[k] = (c1.v[k] == c2.v[k]) ? (R)1 : (R)0;
            const PV<R> eqo = pv_swap(eq);
            if (i >= 0) {
                RR_TRACE("E bb %d %d (pair)\n", i, j);
                if (eq.v[0] != (R)0 && eqo.v[0] != (R)0) { // "balls are in the EXACT same spot" (:250)
                    RR_LV(s_st, l) |= ST_SAME_SPOT;
                } else {
                    PV<R> u, sq;
                    RR_PV_EACH(k) { u.v[k] = c2.v[k] - c1.v[k]; sq.v[k] = u.v[k] * u.v[k]; }
                    const PV<R> sqo = pv_swap(sq);
                    int m1 = A.bmass[bi], m2 = A.bmass[bj];
                    const int m1_0 = m1, m2_0 = m2;
                    PV<R> n1 = c1, n2 = c2, a, a2, pr1, pr2;
                    RR_PV_EACH(k) {
                        const R d12 = m_sqrt(sq.v[k] + sqo.v[k]);
                        const R rr_ = u.v[k] * (R)7 / d12;
                        const R p1 = c1.v[k] + rr_, p2 = c2.v[k] - rr_;
                        const R buffer = (R)1.1, h = (p2 - p1) / (R)2;
                        if (m1_0 == m2_0) { n1.v[k] = c1.v[k] + h * buffer; n2.v[k] = c2.v[k] - h * buffer; }
                        else if (m1_0 > m2_0) { n2.v[k] = c2.v[k] + (p1 - p2) * buffer; }
                        else { n1.v[k] = c1.v[k] + (p2 - p1) * buffer; }
                        // centre setters are incremental: new centre = c + (n - c)
                        const R cc1 = c1.v[k] + (n1.v[k] - c1.v[k]), cc2 = c2.v[k] + (n2.v[k] - c2.v[k]);
                        a.v[k] = cc2 - cc1;          // tplVect1to2
                        a2.v[k] = a.v[k] * a.v[k];
                    }
                    if (m1_0 > m2_0) m2 = m1_0; else if (m1_0 < m2_0) m1 = m2_0;
                    PV<R> v1 = pv_ld2<R>(&A.p.bvx[0], NB, bi, l), v2 = pv_ld2<R>(&A.p.bvx[0], NB, bj, l);
                    RR_PV_EACH(k) { const R bneg = a.v[k] * (R)-1; pr1.v[k] = a.v[k] * v1.v[k]; pr2.v[k] = bneg * v2.v[k]; } // tplVect2to1 = -a
                    const PV<R> a2o = pv_swap(a2), pr1o = pv_swap(pr1), pr2o = pv_swap(pr2);
                    // the two projections: t1 on the even lane, t2 on the odd one -- one division instruction for both
                    PV<R> tt;
                    RR_PV_EACH(k) {
                        const int p = RR_PV_ROLE(l, k);
                        const R d2 = a2.v[k] + a2o.v[k];
                        const R num = p ? (pr2.v[k] + pr2o.v[k]) : (pr1.v[k] + pr1o.v[k]);
                        int lst = 0;
                        tt.v[k] = div0<R>(num, d2, lst);
                        if (lst) RR_LV(s_st, l) |= lst;
                    }
                    const PV<R> t1 = pv_from0(tt, l), t2 = pv_from1(tt, l);
                    const PV<R> f1 = pv_ld2<R>(&A.bfx[0], NB, bi, l), f2 = pv_ld2<R>(&A.bfx[0], NB, bj, l);
                    PV<R> c1n = c1, c2n = c2, lo1 = pv_ld2<R>(&A.p.bl[0], 2 * NB, bi, l), hi1 = pv_ld2<R>(&A.p.brt[0], 2 * NB, bi, l),
                          lo2 = pv_ld2<R>(&A.p.bl[0], 2 * NB, bj, l), hi2 = pv_ld2<R>(&A.p.brt[0], 2 * NB, bj, l), ab1, ab2;
                    const PV<R> pf1 = pv_ld2<R>(&A.pfx[0], NB, bi, l), pf2 = pv_ld2<R>(&A.pfx[0], NB, bj, l);
                    RR_PV_EACH(k) {
                        const int p = RR_PV_ROLE(l, k);
                        const R bneg = a.v[k] * (R)-1;
                        const R df = t1.v[k] * a.v[k] - t2.v[k] * bneg;
                        R w1 = v1.v[k], w2 = v2.v[k];
                        w1 -= df * (R).995; w2 += df * (R).995;
                        const R g1 = f1.v[k], g2 = f2.v[k];
                        if (g1 > (R)0) w1 = py_max<R>(w1, g1); else if (g1 < (R)0) w1 = py_min<R>(w1, g1);
                        if (g2 > (R)0) w2 = py_max<R>(w2, g2); else if (g2 < (R)0) w2 = py_min<R>(w2, g2);
                        v1.v[k] = w1; v2.v[k] = w2;
                        // ball_shift(i, n1x - x1, 0); ball_shift(i, 0, n1y - y1): this axis' delta, and the other call's + 0 (after it for x, before for y)
                        const R e1 = n1.v[k] - c1.v[k], e2 = n2.v[k] - c2.v[k];
                        const R e1a = p ? (R)0 : e1, e1b = p ? e1 : (R)0, e2a = p ? (R)0 : e2, e2b = p ? e2 : (R)0;
                        c1n.v[k] = (c1.v[k] + e1a) + e1b; lo1.v[k] = (lo1.v[k] + e1a) + e1b; hi1.v[k] = (hi1.v[k] + e1a) + e1b;
                        c2n.v[k] = (c2.v[k] + e2a) + e2b; lo2.v[k] = (lo2.v[k] + e2a) + e2b; hi2.v[k] = (hi2.v[k] + e2a) + e2b;
                        ab1.v[k] = m_abs(c1n.v[k] - pf1.v[k]); ab2.v[k] = m_abs(c2n.v[k] - pf2.v[k]);
                    }
                    const PV<R> ab1o = pv_swap(ab1), ab2o = pv_swap(ab2);
                    pv_st2<R>(&A.p.bcx[0], NB, bi, l, c1n); pv_st2<R>(&A.p.bl[0], 2 * NB, bi, l, lo1); pv_st2<R>(&A.p.brt[0], 2 * NB, bi, l, hi1);
                    pv_st2<R>(&A.p.bcx[0], NB, bj, l, c2n); pv_st2<R>(&A.p.bl[0], 2 * NB, bj, l, lo2); pv_st2<R>(&A.p.brt[0], 2 * NB, bj, l, hi2);
                    pv_st2<R>(&A.p.bvx[0], NB, bi, l, v1); pv_st2<R>(&A.p.bvx[0], NB, bj, l, v2);
                    RR_PV_EACH(k) {
                        if (RR_PV_ROLE(l, k) == 0) {
                            A.bmass[bi] = m1; A.bmass[bj] = m2;
                            const R x1 = ab1.v[k] + ab1o.v[k], x2 = ab2.v[k] + ab2o.v[k];
                            if (x1 > A.exc[bi]) A.exc[bi] = x1;
                            if (x2 > A.exc[bj]) A.exc[bj] = x2;
                        }
                    }
                }
            }
        }
        uint64_t m_same = 0, m_div0 = 0;
        RR_FOR_LANES(l) {
            RR_VOTE(m_same, l, (RR_LV(s_st, l) & ST_SAME_SPOT) != 0);
            RR_VOTE(m_div0, l, (RR_LV(s_st, l) & ST_DIV0) != 0);
        }
        if (m_same) st |= ST_SAME_SPOT;
        if (m_div0) st |= ST_DIV0;
        RR_SYNC();
    }
}
// bounce_ball_off_wall (RR_TrashyPhysics.py:320-338) -- independent per ball, one lane each
template <class C> RR_HD void bounce_ball_off_wall_lane(Arena<C> &A, const SimParams<typename C::Real> &sp, int b) {
    using R = typename C::Real;
    if (A.p.bl[b] < (R)0) { R v = A.p.bl[b] * (R)-1.1; ball_shift(A, b, v - A.p.bl[b], (R)0); A.p.bvx[b] *= (R)-1 * (R).8; A.bmass[b] = 3; }
    if (A.p.brt[b] > sp.W) { R v = sp.W - (A.p.brt[b] - sp.W) * (R)1.1; ball_shift(A, b, v - A.p.brt[b], (R)0); A.p.bvx[b] *= (R)-1 * (R).8; A.bmass[b] = 3; }
    if (A.p.bt[b] <= (R)0) { R v = A.p.bt[b] * (R)-1.1; ball_shift(A, b, (R)0, v - A.p.bt[b]); A.p.bvy[b] *= (R)-1 * (R).8; A.bmass[b] = 3; }
    if (A.p.bb[b] >= sp.H) { R v = sp.H - (A.p.bb[b] - sp.W) * (R)1.1; ball_shift(A, b, (R)0, v - A.p.bb[b]); A.p.bvy[b] *= (R)-1 * (R).8; A.bmass[b] = 3; }
    ball_exc_update(A, b);
}
// Ball.move (RR_Ball.py:78-105)
template <class C> RR_HD void ball_move_lane(Arena<C> &A, int b) {
    using R = typename C::Real;
    R vx = A.p.bvx[b], vy = A.p.bvy[b], fx = A.bfx[b], fy = A.bfy[b];
    if (vx >= (R)0 && fx >= (R)0) vx = py_max<R>(vx, fx);
    else if (vx <= (R)0 && fx <= (R)0) vx = py_min<R>(vx, fx);
    else vx += fx;
    if (vy >= (R)0 && fy >= (R)0) vy = py_max<R>(vy, fy);
    else if (vy <= (R)0 && fy <= (R)0) vy = py_min<R>(vy, fy);
    else vy += fy;
    R nl = A.p.bl[b] + vx;
    ball_shift(A, b, nl - A.p.bl[b], (R)0);
    R nt = A.p.bt[b] + vy;
    ball_shift(A, b, (R)0, nt - A.p.bt[b]);
    vx *= (R).995; vy *= (R).995;
    if (m_abs(vx) < (R)0.005) vx = (R)0;
    if (m_abs(vy) < (R)0.005) vy = (R)0;
    A.p.bvx[b] = vx; A.p.bvy[b] = vy;
}
// FloatRect.copy() of a ball rect (MyUtils.py:150-154): centre re-derived through the setters
template <typename R> RR_HD R ball_copy_c(R c) { return (R)7 + (c - (R)7); }
// Ball.undo_move (RR_Ball.py:107-113): rect = copy of the (copied) prior-frame rect
template <class C> RR_HD void ball_undo_lane(Arena<C> &A, int b) {
    using R = typename C::Real;
    R dx = A.pfx[b] - (R)7, dy = A.pfy[b] - (R)7;
    A.p.bcx[b] = (R)7 + dx; A.p.bl[b] = (R)0 + dx; A.p.brt[b] = (R)14 + dx;
    A.p.bcy[b] = (R)7 + dy; A.p.bt[b] = (R)0 + dy; A.p.bb[b] = (R)14 + dy;
}

// budgeted step (see step_arena): the clock an arena is measured against, and what a sub-step that parks BETWEEN TWO PASSES of its
// resolve loop hands back
struct ParkCtx {
    uint32_t *buf = nullptr;            // this arena's PARK_WORDS slot
    uint32_t budget = 0;                // shader clocks (s_memtime ticks) a wavefront may run before its expensive arenas park
    unsigned long long t_begin = 0;     // s_memtime at the wavefront's start
    uint32_t *host_rng = nullptr;       // host emulation only: park at pseudo-random sub-step boundaries, quiet ones included
    uint32_t host_mod = 0;              //   (1 in host_mod; <= 1: at every boundary)
    RR_HD bool over(int work) const {
#if RR_GPU
        return work > 0 && (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin) > (unsigned long long)budget;
#else
        (void)work;
        if (!host_rng) return false;
        *host_rng = *host_rng * 1664525u + 1013904223u;
        return host_mod <= 1 || ((*host_rng >> 16) % host_mod) == 0;
#endif
    }
};
struct MidState {
    int phase = 0;      // 0: at a sub-step boundary; 1: inside _resolve_ball_collisions, before pass `count + 1`
    int count = 0;
    uint32_t bots_moved = 0, balls_moved = 0, n_sub = 0;
    int st_sub = 0, work = 0;
    Hit hit = { 0, 0 };
};

// ------------------------------------------------------------------------------------------------ sub-step pieces (RR_EnvBase.py:303-454)
// returns whether any pair collided (i.e. whether any robot may have been put back)
template <class C> RR_HDN bool resolve_bot_collisions(Arena<C> &A, const SimParams<typename C::Real> &sp, uint32_t &bots_moved, uint32_t &naughty, int &st, int &work, Hit &hit) {
    if (C::NPR == 0) return false;
    uint32_t pairs = detect_robot_pairs(A);
    if (!pairs) return false;
    int attempts = 0;
    while (pairs) {
        attempts++;
        work += 4;
        if (attempts > C::NR) { st |= ST_BOT_RESOLVE_FAIL; return true; }
#pragma unroll 1
        for (uint32_t todo = pairs; todo; todo &= todo - 1) {
            const int p = low_bit(todo);
            int i, j;
            pair_of<C>(p, C::NR, i, j);
            hit.r |= (1u << i) | (1u << j);
            // on_robot_collision -> NaughtyBots (RR_ScoreKeepers.py:123-128)
            if (A.i.thl[i] != 0 || A.i.thr[i] != 0) naughty |= 1u << i;
            if (A.i.thl[j] != 0 || A.i.thr[j] != 0) naughty |= 1u << j;
            uint32_t undo = bots_moved & ((1u << i) | (1u << j));
            if (!undo) { st |= ST_BOT_STUCK; return true; }
            bots_moved &= ~undo;
            RR_FOR_LANES(l) {
                if (l < C::NR && (undo & (1u << l))) robot_undo_lane(A, sp, l);
            }
            RR_SYNC();
        }
        pairs = detect_robot_pairs(A);
    }
    return true;
}
// returns 1 (resolved), 0 (gave up after 10 passes) or, budgeted step only, -1: over the budget between two passes -- `count` passes are
// done, the caller parks the arena and a later call re-enters here with count0 = count (at least one pass runs per entry)
template <class C, bool BUDGET = false>
RR_HDN int resolve_ball_collisions(Arena<C> &A, const SimParams<typename C::Real> &sp, uint32_t bots_moved, int &st, int &work, Hit &hit,
                                   int count0 = 0, const ParkCtx *pk = nullptr, int *count_out = nullptr) {
    bool naughty = true;
    int count = count0;
    RR_T0();
    if (!BUDGET || count0 == 0) {
        RR_FOR_LANES(l) { if (l < C::NB) ball_exc_update(A, l); } // where the push and the roll have left each ball
    }
    while (naughty) {
        if constexpr (BUDGET) {
            if (count > count0 && count < 10 && RR_UNLIKELY(pk->over(1))) { *count_out = count; return -1; }
        }
        count++;
        work++;
        if (count > 10) { RR_TRACE("E resolve gave up\n"); return 0; }
        naughty = false;
        uint64_t bb = detect_ball_pairs(A);
#if RR_CBR
        if (bb) {
            naughty = true;
#pragma unroll 1
            for (uint64_t todo = bb; todo; todo &= todo - 1) {
                int i, j;
                pair_of<C>(low_bit(todo), C::NB, i, j);
                RR_TRACE("E pass %d bb %d %d\n", count, i, j);
                hit.b |= (1u << i) | (1u << j);
            }
            bounce_balls_pass(A, bb, st);
        }
#else
#pragma unroll 1
        for (uint64_t todo = bb; todo; todo &= todo - 1) {
            int i, j;
            pair_of<C>(low_bit(todo), C::NB, i, j);
            RR_TRACE("E pass %d bb %d %d\n", count, i, j);
            hit.b |= (1u << i) | (1u << j);
            naughty = true;
            bounce_balls(A, i, j, st);
        }
#endif
        RR_STAMP(14);
        // caches already built in this sub-step (a hit in the push or in an earlier pass)?  then the cheap variant
        uint32_t br = A.sides_ok ? detect_ball_robot<C, true>(A, sp) : detect_ball_robot<C, false>(A, sp);
        RR_STAMP(15);
#if RR_CBR
        if (br) {
            naughty = true;
            for (int b = 0; b < C::NB; b++) {
                const uint32_t rm = (br >> (b * C::NR)) & ((1u << C::NR) - 1u);
                hit.r |= rm; hit.b |= rm ? (1u << b) : 0u;
            }
#if RR_CARRY
#pragma unroll 1
            for (uint32_t todo = br; todo; todo &= todo - 1) bounce_pass(A, sp, todo & (0u - todo), bots_moved, st); // one at a time: the scratch rect
#else
            bounce_pass(A, sp, br, bots_moved, st); // the bounces of different balls side by side
#endif
        }
#else
#pragma unroll 1
        for (uint32_t todo = br; todo; todo &= todo - 1) {
            const int p = low_bit(todo);
            naughty = true;
            hit.b |= 1u << (p / C::NR); hit.r |= 1u << (p % C::NR);
            bounce_ball_off_bot(A, sp, p % C::NR, p / C::NR, bots_moved, st);
        }
#endif
        RR_STAMP(16);
        uint32_t bw = detect_ball_wall(A, sp);
        if (bw) {
            naughty = true;
            hit.b |= bw;
            RR_FOR_LANES(l) {
                if (l < C::NB && (bw & (1u << l))) bounce_ball_off_wall_lane(A, sp, l);
            }
            RR_SYNC();
        }
        RR_STAMP(17);
#if !RR_GPU && defined(RR_EMU_TRACE)
        if (RR_EMU_TRACE > 1) { // host emulation, trace level 2: the balls after this pass, bit for bit (pass-periodicity studies, profiles/r04/pass_replay_experiment)
            fprintf(stderr, "E pass %d state", count);
            for (int b = 0; b < C::NB; b++) fprintf(stderr, " %a %a %a %a %a %a %a %a", (double)A.p.bcx[b], (double)A.p.bcy[b], (double)A.p.bl[b], (double)A.p.brt[b], (double)A.p.bt[b], (double)A.p.bb[b], (double)A.p.bvx[b], (double)A.p.bvy[b]);
            fprintf(stderr, "\n");
        }
#endif
    }
    RR_TRACE("E resolve done in %d passes\n", count);
    return 1;
}
template <class C>
RR_HDN void undo_naughty_movement(Arena<C> &A, const SimParams<typename C::Real> &sp, uint32_t &balls_moved,
                                 uint32_t &bots_moved, int &st, Hit &hit) {
    bool naughty = true;
    int count = 0;
    const int limit = C::NB + C::NR;
    RR_T0();
    while (naughty) {
        count++;
        if (count > limit) {
            if (sp.game_mode) st |= ST_UNDO_WARN;
            else { st |= ST_UNDO_FAIL; return; }
        }
        uint32_t nbots = 0, nballs = 0;
        uint64_t bb = detect_ball_pairs(A);
#pragma unroll 1
        for (uint64_t todo = bb; todo; todo &= todo - 1) {
            int i, j;
            pair_of<C>(low_bit(todo), C::NB, i, j);
            nballs |= (1u << i) | (1u << j);
        }
        uint32_t br = A.sides_ok ? detect_ball_robot<C, true>(A, sp) : detect_ball_robot<C, false>(A, sp);
#pragma unroll 1
        for (uint32_t todo = br; todo; todo &= todo - 1) {
            const int p = low_bit(todo);
            nballs |= 1u << (p / C::NR);
            nbots |= 1u << (p % C::NR);
        }
        nballs |= detect_ball_wall(A, sp);
        RR_STAMP(18);
        naughty = (nbots | nballs) != 0;
        hit.r |= nbots; hit.b |= nballs;
        uint32_t ubots = bots_moved & nbots, uballs = balls_moved & nballs;
        RR_TRACE("E undo iteration %d: contacts r %x b %x, undone r %x b %x\n", count, nbots, nballs, ubots, uballs);
        bots_moved &= ~ubots;
        balls_moved &= ~uballs;
        if (ubots | uballs) {
            RR_FOR_LANES(l) {
                if (l < C::NR && (ubots & (1u << l))) robot_undo_lane(A, sp, l);
                if (l < C::NB && (uballs & (1u << l))) ball_undo_lane(A, l);
            }
            RR_SYNC();
        }
        RR_STAMP(19);
        if (naughty && !(ubots | uballs)) {
            // nothing left to undo: every further iteration would find the same contacts and change nothing, so the
            // reference ends the same way -- the raise after NB+NR iterations (GAME_MODE=False) or the warning
            // followed by an endless loop (GAME_MODE=True).  Exit now with exactly those status bits.
            st |= ST_UNDO_FAIL | (sp.game_mode ? ST_UNDO_WARN : 0);
            return;
        }
    }
}

// the two fast phases of a sub-step (see substep); FZ: an island is frozen
template <class C, bool FZ>
RR_HD void substep_phase1(Arena<C> &A, const SimParams<typename C::Real> &sp, const Hit &fz, uint32_t prev_moved, uint64_t &m_rr,
                          uint64_t &m_br, uint64_t &m_wm, uint64_t &m_brk) {
    using R = typename C::Real;
    // A robot's move is spread over a PAIR of lanes (2r, 2r+1) when the virtual wave has them: both lanes run the same
    // instructions on different data -- two of the three sin/cos evaluations at once, then one renormalised corner each --
    // and swap results through DPP.  Same operations on the same operands as robot_move_lane; ~30 % fewer instructions.
    constexpr bool PAIRED = C::VW >= 2 * C::NR;
    RR_LANE_VAR(R, sv); RR_LANE_VAR(R, cv); // sin / cos of this lane's angle: a1 on the even lane, a2 on the odd one
    RR_LANE_VAR(R, qx); RR_LANE_VAR(R, qy); // this lane's corner: TL on the even lane, TR on the odd one
    RR_LANE_VAR(MovePlan<R>, mp);           // the robot's move plan (both lanes of the pair hold a copy)
    RR_LANE_VAR(R, s3v); RR_LANE_VAR(R, c3v); // sin / cos of the third angle (the re-centring after a pivot), issued with the corners
    RR_FOR_LANES(l) {
        bool c_rr = false, c_br = false, c_brk = false; // (c_brk, frozen variant: an outside ball within the bound of an ISLAND robot -- see "witness" in substep)
        const int r = PAIRED ? (l >> 1) : l, part = PAIRED ? (l & 1) : 0;
        // a frozen robot still makes its move here (and is put back at the end of the sub-step): how the move meets the walls
        // depends on the robot's incrementally kept edges, so it is re-evaluated, not assumed; only its pair tests are skipped
        const bool robot_lane = r < C::NR;
        const bool own = robot_lane && !(FZ && ((fz.r >> (r < C::NR ? r : 0)) & 1u));
        if (robot_lane && part == 0) {
            if (prev_moved & (1u << r)) { A.p.px[r] = A.ax[r]; A.p.py[r] = A.ay[r]; A.p.prot[r] = A.arot[r]; }
            // on_frame_begin (RR_Robot.py:119-120): the ring entry written this frame
            const R ox = A.p.rcx[r], oy = A.p.rcy[r];
            A.ax[r] = ox; A.ay[r] = oy; A.arot[r] = A.p.rrot[r];
            for (int j = 0; j < C::NR; j++) { // robot-robot: needs centres within 2 x 22.36 (+ 2 x 3 px of motion)
                R dx = A.p.rcx[j] - ox, dy = A.p.rcy[j] - oy;
                // each pair once (j > r); a frozen robot tests nothing itself, so its partner outside the island tests the pair
                bool cl = own & ((j > r) | (FZ && ((fz.r >> j) & 1u) != 0)) & (j != r) & (dx * dx + dy * dy <= (R)(51.5 * 51.5));
                // (frozen variant only -- it costs the common path nothing: a spurious "close" thaws the island, so the pair also
                // has to pass the separating-axis test, grown by the 2 x 3 px the two robots can still move; every robot is
                // still on its frame-begin pose here)
                if ((FZ || RR_BROAD_TIGHT) && PAIRED && cl) cl = !robots_separated(A, r, j, dx, dy, (R)6.05);
                c_rr = c_rr | cl;
            }
            if (!PAIRED) { // _move_bots
                const int wm_before = FZ ? A.wm[r] : 0;
                robot_move_lane(A, sp, r);
                if (FZ && !own && A.wm[r] != wm_before) c_rr = true; // a frozen robot met the walls differently: thaw
            }
        }
        if (PAIRED) {
            R s_ = (R)0, c_ = (R)1;
            if (robot_lane) {
                const MovePlan<R> m = robot_move_plan(A, r);
                sincos_deg<R>(part == 0 ? m.a1 : m.a2, s_, c_);
                RR_LV(mp, l) = m;
            }
            RR_LV(sv, l) = s_; RR_LV(cv, l) = c_;
        }
        if (l < C::NB) {
            if (!(FZ && ((fz.b >> l) & 1u))) { // on_frame_begin (RR_Ball.py:63-68)
                A.bmass[l] = 1; A.bfx[l] = (R)0; A.bfy[l] = (R)0; A.exc[l] = (R)0;
                A.pfx[l] = ball_copy_c<R>(A.p.bcx[l]); A.pfy[l] = ball_copy_c<R>(A.p.bcy[l]);
                A.reach[l] = (R)14.04 + ((m_abs(A.p.bvx[l]) + m_abs(A.p.bvy[l])) * (R)1.01 + (R)0.02);
#if RR_CARRY
                // the frozen island's sweeps found the scratch rect where the ball before each of its balls (and the last ball, via
                // A.p.ic) had left it: such a ball outside the island has to stand still, or the island is thawed
                if (FZ && ((carry_deps<C>(fz.b) >> l) & 1u)) c_br = c_br | (A.p.bvx[l] != (R)0) | (A.p.bvy[l] != (R)0);
#endif
                for (int r2 = 0; r2 < C::NR; r2++) { // ball-robot: 22.36 + 9.9 (+ 3 px of robot motion)
                    R dx = A.p.bcx[l] - A.p.rcx[r2], dy = A.p.bcy[l] - A.p.rcy[r2];
                    bool cl = dx * dx + dy * dy <= (R)(36.0 * 36.0);
                    if ((FZ || RR_BROAD_TIGHT) && PAIRED && cl) { // the robot-frame bound of ball_near_robot, grown by the same 3 px
                        const R *q = RR_REL(A)[r2];
                        const R ux = (q[2] - q[0]) * (R)0.05, uy = (q[3] - q[1]) * (R)0.05, vx = (q[4] - q[0]) * (R)0.025, vy = (q[5] - q[1]) * (R)0.025;
                        cl = (m_abs(dx * ux + dy * uy) <= (R)20.05) & (m_abs(dx * vx + dy * vy) <= (R)30.05);
                    }
                    if (FZ && ((fz.r >> r2) & 1u)) c_brk = c_brk | cl;
                    else c_br = c_br | cl;
                }
            } else { // frozen ball: anywhere within its recorded excursion, against the robots outside the island
                const R reach = (R)36.05 + A.exc[l];
                for (int r2 = 0; r2 < C::NR; r2++) {
                    R dx = A.p.bcx[l] - A.p.rcx[r2], dy = A.p.bcy[l] - A.p.rcy[r2];
                    c_br = c_br | ((((fz.r >> r2) & 1u) == 0) & (dx * dx + dy * dy <= reach * reach));
                }
            }
        }
        RR_VOTE(m_rr, l, c_rr);
        RR_VOTE(m_br, l, c_br);
        if (FZ) RR_VOTE(m_brk, l, c_brk);
    }
    if (PAIRED) {
        // (the broad phase above read robot centres that no lane has moved yet: its bounds hold a fortiori)
        RR_FOR_LANES(l) { // the rotation setter's sin/cos sits in the odd lane: both lanes need it for their corner
            const int r = l >> 1, part = l & 1;
            const R ps = RR_XOR1(sv, l), pc = RR_XOR1(cv, l);
            const R s2 = part ? RR_LV(sv, l) : ps, c2 = part ? RR_LV(cv, l) : pc;
            R ax = (R)0, ay = (R)0;
            if (r < C::NR) {
                corner_from_sc<R>(RR_LV(mp, l).nrot, s2, c2, part ? (R)10 : (R)-10, (R)-20, sp.rob_cdist, ax, ay);
            }
            RR_LV(qx, l) = ax; RR_LV(qy, l) = ay;
            // the third angle's sin/cos does not depend on the corners: issued here, next to them (same instructions for the wave,
            // only the even lane's values are used), the two chains overlap instead of running one after the other (G +1.5 %)
            R s3_ = (R)0, c3_ = (R)1;
            if (r < C::NR) sincos_deg<R>(RR_LV(mp, l).a3, s3_, c3_);
            RR_LV(s3v, l) = s3_; RR_LV(c3v, l) = c3_;
        }
        RR_FOR_LANES(l) {
            const int r = l >> 1, part = l & 1;
            const R trx = RR_XOR1(qx, l), try_ = RR_XOR1(qy, l); // the odd lane's corner (TR), seen from the even lane
            bool wm_changed = false;
            if (part == 0 && r < C::NR) {
                const MovePlan<R> m = RR_LV(mp, l);
                const int wm_before = FZ ? A.wm[r] : 0;
                A.i.mc[r] += 1;
                if (!m.idle) {
                    const R s3 = RR_LV(s3v, l), c3 = RR_LV(c3v, l);
                    robot_move_finish(A, sp, r, m, RR_LV(sv, l), RR_LV(cv, l), s3, c3, RR_LV(qx, l), RR_LV(qy, l), trx, try_);
                } else {
                    A.wm[r] = 0;
                }
                if (FZ) wm_changed = ((fz.r >> r) & 1u) && A.wm[r] != wm_before; // the frozen robot met the walls differently
            }
            if (FZ) RR_VOTE(m_wm, l, wm_changed);
        }
    }
}
template <class C, bool FZ>
RR_HD void substep_phase2(Arena<C> &A, const SimParams<typename C::Real> &sp, const Hit &fz, uint64_t &m_any, int icm, uint64_t &m_anyk) {
    using R = typename C::Real;
    RR_FOR_LANES(l) {
        bool c = false, ck = false; // (ck, frozen variant: a rolled ball within the radius bound of an ISLAND robot)
        if (l < C::NB && !(FZ && ((fz.b >> l) & 1u))) {
            ball_move_lane(A, l);
            const R mx = A.p.bcx[l], my = A.p.bcy[l];
#if RR_CARRY
            // one robot, the scratch rect riding on the ball (carry_quiet_sweep): a roll that is not an exact difference would leave it
            // a last bit off the ball -- take the full path, which carries it explicitly
            if (C::NR < 2 && icm) c = c | (carry1<R>(A.pfx[l], mx) != mx) | (carry1<R>(A.pfy[l], my) != my);
#endif
            for (int j = 0; j < C::NB; j++) {
                R dx = A.p.bcx[j] - mx, dy = A.p.bcy[j] - my;
                R reach = A.reach[j];
                if (FZ) reach = ((fz.b >> j) & 1u) ? (R)14.06 + A.exc[j] : reach; // a frozen ball: wherever its island carried it
                c = c | ((j != l) & (dx * dx + dy * dy <= reach * reach));
            }
            for (int r = 0; r < C::NR; r++) {
                if (FZ && ((fz.r >> r) & 1u)) { // a frozen robot: its frame-begin or its (undone) moved pose -- the radius bound + 3 px
                    R dx = mx - A.p.rcx[r], dy = my - A.p.rcy[r];
                    ck = ck | (dx * dx + dy * dy <= (R)(36.0 * 36.0));
                } else {
                    c = c | ball_near_robot(A, l, r);
                }
            }
            c = c | (ball_in_play(A, l) & ball_collided_wall(A, sp, l));
        }
        RR_VOTE(m_any, l, c);
        if (FZ) RR_VOTE(m_anyk, l, ck);
    }
}
// _push_balls over a frozen hit list (RR_EnvBase.py:335-339), ball-major order
template <class C>
RR_HD void push_balls(Arena<C> &A, const SimParams<typename C::Real> &sp, uint32_t br, uint32_t bots_moved, int &st, Hit &hit) {
    RR_T0();
#pragma unroll 1
    for (uint32_t todo = br; todo; todo &= todo - 1) {
        const int p = low_bit(todo);
        hit.b |= 1u << (p / C::NR); hit.r |= 1u << (p % C::NR);
        int kf, kb;
#if RR_CARRY
        kf = kb = -2; // the two responses may see the scratch rect at centres one ulp apart: each runs its own search
#else
        {
            using R = typename C::Real;
            const int r = p % C::NR, b = p / C::NR;
            V2<R> bc = { A.p.bcx[b], A.p.bcy[b] };
            Seg<R> dia[2];
            force_diameters(A, r, bc, dia);
            first_surface_hit2(A, r, dia, (R).5, (R)0, kf, kb);
        }
#endif
        apply_force_to_ball(A, sp, p % C::NR, p / C::NR, bots_moved, st, kf);
        RR_STAMP(20);
        bounce_ball_off_bot(A, sp, p % C::NR, p / C::NR, bots_moved, st, kb);
        RR_STAMP(21);
    }
    if (br) { // the pushed balls' force and velocity changed: their roll bound with them
        RR_FOR_LANES(l) { if (l < C::NB) A.reach[l] = ball_reach(A, l); }
        RR_SYNC();
    }
}
// the frozen island rejoins the sub-step: its robots have already made their move in phase 1, its balls run the frame
// hooks they skipped
template <class C> RR_HD void thaw_island(Arena<C> &A, const SimParams<typename C::Real> &sp, Hit &fz, uint32_t &bots_moved) {
    using R = typename C::Real;
    RR_FOR_LANES(l) {
        if (l < C::NB && ((fz.b >> l) & 1u)) {
            A.bmass[l] = 1; A.bfx[l] = (R)0; A.bfy[l] = (R)0; A.exc[l] = (R)0;
            A.pfx[l] = ball_copy_c<R>(A.p.bcx[l]); A.pfy[l] = ball_copy_c<R>(A.p.bcy[l]);
            A.reach[l] = (R)14.04 + ((m_abs(A.p.bvx[l]) + m_abs(A.p.bvy[l])) * (R)1.01 + (R)0.02);
        }
    }
    RR_SYNC();
    bots_moved |= fz.r;
    fz.r = 0; fz.b = 0;
}

// one of the 12 physics sub-steps ("frame", RR_EnvBase.py:275-287).
// Wall time per sub-step is set by the number of dependent phases (LDS round trip + ballot each), not by their
// arithmetic, so the common no-contact path is folded into two phases: [frame hooks + robot moves + robot-robot and
// ball-robot broad phase] and [roll + ball-ball / ball-robot broad phase + wall test], with motion-aware bounds.  Only
// when a ballot reports something close do the reference-shaped loops below (unchanged, exact) run.
// `prev_moved`: robots whose move of the previous sub-step survived (their ring entry moveCount-1 is that frame's).
// `work`: += the contact-resolution passes this sub-step ran (0 on the common path); `hit`: who took part in them.
//
// Island freeze (`fz`).  A stuck island -- a ball squeezed between two robots, robots locked against each other --
// costs ~150 quiet sub-steps per sub-step (push, ten resolve passes, undo loop) and ends every sub-step exactly where
// it began, while the rest of the arena moves on; one such arena used to set the duration of a 65,536-arena launch.
// The sub-step is deterministic, and entities only influence each other through pair tests that HIT.  So once the
// entities that took part in the hits of a sub-step (the island K = fz.r | fz.b) are found bit-identical to what
// they were a sub-step earlier (step_arena checks that), the next sub-step maps K onto itself again PROVIDED nothing
// outside K hits K -- and then K need not be recomputed at all.  A frozen sub-step therefore runs the two fast phases
// for everything outside K and keeps K as an obstacle: its robots with the usual motion slack, its balls with the
// excursion `exc` their responses reached while the island was last computed.  If every broad test involving an
// entity outside K stays negative, nothing outside K touched anything (K included) and the sub-step is complete: exact.
// If any fires, the island is thawed on the spot: its balls run the frame hooks they skipped (and, if the thaw comes
// after the roll phase, the push restricted to K and their roll), and the reference-shaped path runs for the whole
// arena as if it had never been frozen.  The island's robots are never held back: they make their real move in phase 1
// like every robot (how it meets the walls depends on their drifting edges) and are put back at the end of the sub-step.
// Budgeted step: returns true when the arena parked between two passes of the resolve loop (`mid` holds where); a call with
// mid->phase == 1 skips everything before that loop and re-enters it.
template <class C, bool BUDGET = false>
RR_HD bool substep(Arena<C> &A, const SimParams<typename C::Real> &sp, uint32_t &naughty, int &st, uint32_t &prev_moved,
                   int &work, Hit &fz, Hit &hit, int &icm, MidState *mid = nullptr, const ParkCtx *pk = nullptr) {
    using R = typename C::Real;
    uint32_t bots_moved = ((1u << C::NR) - 1) & ~fz.r, balls_moved = (1u << C::NB) - 1;
    RR_TRACE("E substep\n");
    RR_T0();
    uint64_t m_any = 0;
    int count0 = 0;
    bool reentry = false, put_back = false;
    if constexpr (BUDGET) reentry = mid->phase == 1;
    if (RR_UNLIKELY(reentry)) {
        bots_moved = mid->bots_moved; balls_moved = mid->balls_moved; naughty = mid->n_sub; st = mid->st_sub; work = mid->work; hit = mid->hit;
        count0 = mid->count;
        mid->phase = 0;
        m_any = 1;
    } else {
    // phase 1: frame hooks, robot moves AND the exact broad phase.  The broad phase runs in the same phase as the moves:
    // it may see a robot centre from before or after this sub-step's move, so its bounds carry the largest centre
    // displacement a move can cause (1 px drive / 0.17 px pivot / <= 1.5 px wall clamp: 3 px per robot is generous).
    uint64_t m_rr = 0, m_br = 0, m_wm = 0, m_brk = 0;
    const bool frozen = (fz.r | fz.b) != 0; // the frozen variants are separate instantiations: the common path pays nothing
    if (RR_UNLIKELY(frozen)) substep_phase1<C, true>(A, sp, fz, prev_moved, m_rr, m_br, m_wm, m_brk);
    else substep_phase1<C, false>(A, sp, fz, prev_moved, m_rr, m_br, m_wm, m_brk);
    RR_SYNC();
    RR_STAMP(1);
    // Witness (default build).  A ball OUTSIDE the island within the bounds of one of its robots -- drifting past its flank, half a
    // pixel away -- used to thaw the island in every sub-step only to be found untouched: twelve expensive sub-steps per step while it
    // lasts.  The island K reproduces itself as long as nothing outside touches it, and the only tests the reference runs on such a
    // pair (ball x, island robot r) in a sub-step are ball_robot_collided at: (x before its roll, r moved) in the push sweep; (x rolled,
    // r moved) in every resolve pass and the first undo iteration; (x rolled, r put back) in the later undo iterations.  If those three
    // narrow tests are negative the pair did not interact and K stays frozen: exact.  They run here at exactly those poses -- the frozen
    // robots have made their real move in phase 1, are put back below with the reference's own undo arithmetic, and are restored bit for
    // bit if the third test fires (then the sub-step takes the thaw-after-the-roll path as before).
    uint32_t kx = 0; // pairs (ball outside the island, robot inside)
    if (RR_WITNESS && RR_UNLIKELY(frozen))
        for (int b = 0; b < C::NB; b++) if (!((fz.b >> b) & 1u)) kx |= fz.r << (b * C::NR);
    if (!RR_WITNESS) m_br |= m_brk;
    if (RR_UNLIKELY((fz.r | fz.b) && (m_rr | m_br | m_wm | m_brk))) { // thaw before anything depended on the island
        bool thaw = true;
        if (RR_WITNESS && !(m_rr | m_br | m_wm)) { // only (outside ball, island robot) bounds fired: the push sweep's test decides
            thaw = (detect_ball_robot<C, false>(A, sp) & kx) != 0u;
            RR_TRACE("E witness 1 (before the roll, robots moved): %s\n", thaw ? "hit" : "clear");
        }
        if (thaw) {
            RR_TRACE("E thaw in phase 1: rr %llx br %llx wm %llx\n", (unsigned long long)m_rr, (unsigned long long)(m_br | m_brk), (unsigned long long)m_wm);
            thaw_island(A, sp, fz, bots_moved);
            m_rr = 1; m_br = 1;
        }
    }
    if (RR_UNLIKELY(m_rr)) {
        // an undone robot changes the ball-robot picture: let the full detection decide.  (No pair really touching -- most
        // of the time two robots merely pass within 51.5 px of each other -- leaves every robot where phase 1 put it, and
        // phase 1's own ball-robot bound stands.)
        if (resolve_bot_collisions(A, sp, bots_moved, naughty, st, work, hit)) m_br = 1;
    }
    RR_STAMP(2);
    if (RR_UNLIKELY(m_br)) { // _push_balls (RR_EnvBase.py:335-339): frozen hit list, ball-major order
#if RR_CARRY
        carry_materialize(A, icm, true);
#endif
        // (with fewer than eight lanes per arena a lane of the uncached variant sweeps all four sides and builds every robot-only
        // operand itself -- nine dependent divisions; going through the caches is shorter there: T 650 -> 690 M env-steps/s)
        uint32_t br = (C::VW < 8) ? detect_ball_robot<C, true>(A, sp) : detect_ball_robot<C, false>(A, sp);
        RR_STAMP(22);
        RR_TRACE("E push mask %08x\n", br);
        push_balls(A, sp, br, bots_moved, st, hit);
    }
#if RR_CARRY
    else carry_quiet_sweep(A, icm, true); // the push sweep the fast path skips still walks the scratch rect over every ball
#endif
    RR_STAMP(3);
    // phase 2: _roll_balls AND the fused first pass of _resolve_ball_collisions: anything possibly touching?  Ball-ball
    // runs in the same phase as the roll, so the other ball may be seen before or after its own roll: the bound (A.reach,
    // written at the frame hooks and refreshed by the push) adds the most it can travel in its roll.  Ball-robot uses the settled robot centres; the wall test is the exact int-rect test.
    RR_TRACE("E phase1 rr %d br %d\n", (int)(m_rr != 0), (int)(m_br != 0));
    uint64_t m_anyk = 0;
    if (RR_UNLIKELY(frozen && (fz.r | fz.b))) substep_phase2<C, true>(A, sp, fz, m_any, icm, m_anyk);
    else substep_phase2<C, false>(A, sp, fz, m_any, icm, m_anyk);
    RR_SYNC();
    RR_STAMP(4);
    if (!RR_WITNESS) m_any |= m_anyk;
    if (RR_WITNESS && RR_UNLIKELY((fz.r | fz.b) && !m_any && m_anyk)) { // witnesses 2 and 3 (see above)
        bool hitw = (detect_ball_robot<C, false>(A, sp) & kx) != 0u;
        RR_TRACE("E witness 2 (after the roll, robots moved): %s\n", hitw ? "hit" : "clear");
        if (!hitw) {
            // the island's robots are put back as the undo loop of their (reproduced) sub-step does -- a lane keeps what it needs to
            // restore its robot bit for bit: centre, edges, rotation (the corner offsets are a function of the rotation)
            RR_LANE_VAR(R, w0); RR_LANE_VAR(R, w1); RR_LANE_VAR(R, w2); RR_LANE_VAR(R, w3); RR_LANE_VAR(R, w4); RR_LANE_VAR(R, w5); RR_LANE_VAR(R, w6);
            RR_FOR_LANES(l) {
                const int r = l < C::NR ? l : 0;
                RR_LV(w0, l) = A.p.rcx[r]; RR_LV(w1, l) = A.p.rcy[r]; RR_LV(w2, l) = A.p.rl[r]; RR_LV(w3, l) = A.p.rrt[r];
                RR_LV(w4, l) = A.p.rt[r]; RR_LV(w5, l) = A.p.rb[r]; RR_LV(w6, l) = A.p.rrot[r];
            }
            RR_FOR_LANES(l) { if (l < C::NR && ((fz.r >> l) & 1u)) robot_undo_lane(A, sp, l); }
            RR_SYNC();
            hitw = (detect_ball_robot<C, false>(A, sp) & kx) != 0u;
            RR_TRACE("E witness 3 (after the roll, robots put back): %s\n", hitw ? "hit" : "clear");
            if (hitw) { // the reference's undo loop would have found this contact: back to the moved poses, then the full path
                RR_FOR_LANES(l) {
                    if (l < C::NR && ((fz.r >> l) & 1u)) {
                        if (A.p.rrot[l] != RR_LV(w6, l)) corners_for<R>(RR_LV(w6, l), (R)10, (R)20, sp.rob_cdist, RR_REL(A)[l]);
                        A.p.rcx[l] = RR_LV(w0, l); A.p.rcy[l] = RR_LV(w1, l); A.p.rl[l] = RR_LV(w2, l); A.p.rrt[l] = RR_LV(w3, l);
                        A.p.rt[l] = RR_LV(w4, l); A.p.rb[l] = RR_LV(w5, l); A.p.rrot[l] = RR_LV(w6, l);
                        A.i.mc[l] += 1;
                        A.sides_ok = 0;
                    }
                }
                RR_SYNC();
            } else {
                put_back = true; // (the end of the sub-step has nothing left to do for the island)
            }
        }
        if (hitw) m_any = 1;
    }
    if (RR_UNLIKELY((fz.r | fz.b) && m_any)) { // thaw after the roll phase: the island catches up (move, push, roll), then the full path
        RR_TRACE("E thaw in phase 2\n");
        const Hit k = fz;
        thaw_island(A, sp, fz, bots_moved);
        resolve_bot_collisions(A, sp, bots_moved, naughty, st, work, hit);
        // the push saw the other balls before their roll and found none of them near a robot (phase 1): only K's pairs count
        uint32_t kmask = 0;
        for (int b = 0; b < C::NB; b++) if ((k.b >> b) & 1u) kmask |= ((1u << C::NR) - 1u) << (b * C::NR);
        if (k.b) { // (an island of robots only has no pair to push)
#if RR_CARRY
            // the push's sweep, late: the balls it depends on through the scratch rect stand still or are the island's, not yet rolled
            carry_materialize(A, icm, true);
#endif
            uint32_t br = detect_ball_robot<C, false>(A, sp) & kmask;
            push_balls(A, sp, br, bots_moved, st, hit);
            RR_FOR_LANES(l) { if (l < C::NB && ((k.b >> l) & 1u)) ball_move_lane(A, l); }
            RR_SYNC();
        }
    }
    RR_TRACE("E phase2 any %d\n", (int)(m_any != 0));
#if RR_CARRY
    if (!m_any) carry_quiet_sweep(A, icm, false); // ... and so does the one pass of the resolve loop
    else carry_materialize(A, icm, true);         // (the push's sweep was quiet: the rect is where the last ball was before the roll)
#endif
    } // !reentry
    if (RR_UNLIKELY(m_any)) { // the reference's loop, from its first pass (nothing has changed since the broad phase above)
        int count_now = 0;
        const int rr_ok_ = resolve_ball_collisions<C, BUDGET>(A, sp, bots_moved, st, work, hit, count0, pk, &count_now);
        if constexpr (BUDGET) {
            if (RR_UNLIKELY(rr_ok_ < 0)) { // over the budget between two passes: hand the loop's state back, the arena parks
                mid->phase = 1; mid->count = count_now; mid->bots_moved = bots_moved; mid->balls_moved = balls_moved;
                mid->n_sub = naughty; mid->st_sub = st; mid->work = work; mid->hit = hit;
                return true;
            }
        }
        RR_STAMP(5);
        if (rr_ok_ == 0) { work += 8; undo_naughty_movement(A, sp, balls_moved, bots_moved, st, hit); }
    }
    RR_STAMP(6);
    if (RR_UNLIKELY(fz.r) && !put_back) { // still frozen: the island's robots made their move in phase 1; the undo they would have met puts them back
        RR_FOR_LANES(l) {
            if (l < C::NR && ((fz.r >> l) & 1u)) robot_undo_lane(A, sp, l);
        }
        RR_SYNC();
    }
    prev_moved = bots_moved;
    return false;
}
// Balls OUTSIDE a would-be island K that belong to the picture K reproduces: at rest (v = 0), not touched in the compared sub-steps
// (not in the hit set), bit for bit what they were a sub-step earlier, and within the broad-phase bound of one of K's robots.  Such
// a ball -- typically one the robot pushed aside earlier, resting half a pixel off its flank -- fires the frozen variant's ball-robot
// bound in EVERY sub-step and thaws the island on the spot, only to be found untouched again: freeze, thaw, twelve expensive sub-steps
// per step for as long as it lies there (the slowest arena of a random-policy launch at the steady state: ~1 ms against a median
// wavefront of 65 us, profiles/r04/verify/).  It joins the island instead.  Exact for the same reason the island is: K + the ball was
// identical at two consecutive frame begins and nothing outside hit either, so the sub-step maps K + the ball onto itself again as
// long as nothing outside comes near -- which the frozen variant's bounds keep checking, now around the ball too.
template <class C> RR_HD uint32_t resting_neighbours(const Arena<C> &A, const Hit &k, uint32_t chg_b) {
    using R = typename C::Real;
    uint64_t m = 0;
    RR_FOR_LANES(l) {
        bool near = false;
        const int b = l < C::NB ? l : 0;
        if (l < C::NB && !((k.b >> b) & 1u) && !((chg_b >> b) & 1u) && A.p.bvx[b] == (R)0 && A.p.bvy[b] == (R)0 && ball_in_play(A, b)) {
            for (int r = 0; r < C::NR; r++) {
                const R dx = A.p.bcx[b] - A.p.rcx[r], dy = A.p.bcy[b] - A.p.rcy[r];
                near = near | ((((k.r >> r) & 1u) != 0) & (dx * dx + dy * dy <= (R)(36.0 * 36.0)));
            }
        }
        RR_VOTE(m, l, near);
    }
    return (uint32_t)m & ((1u << C::NB) - 1u);
}
// after the last sub-step: the pose-ring bookkeeping the next sub-step would have done
template <class C> RR_HD void substeps_end(Arena<C> &A, uint32_t prev_moved) {
    RR_FOR_LANES(l) {
        if (l < C::NR && (prev_moved & (1u << l))) { A.p.px[l] = A.ax[l]; A.p.py[l] = A.ay[l]; A.p.prot[l] = A.arot[l]; }
    }
    RR_SYNC();
}

// ------------------------------------------------------------------------------------------------ observation (RR_Observers.py:301-406)
// lidar: one lane per (ray, rect, side); candidates go to LDS, the wave-uniform tail takes the minima
template <class C, typename O>
RR_HDN bool observe(Arena<C> &A, const SimParams<typename C::Real> &sp, int team, int ridx, int bidx, O *out, int &st) {
    using R = typename C::Real;
    if (ridx < 0) {
        if (team == 1 && C::NRH == 0) return false;
        if (team == -1 && C::NRG == 0) return false;
        ridx = (team == 1) ? 0 : C::NRH;
    }
    if (bidx < 0) bidx = 0;
    RR_FOR_LANES(l) { if (l < C::NR) A.irot[l] = (R)NAN; } // the candidates below overwrite the aliased inner-square offsets
    RR_SYNC();
    ensure_sides(A);
    // one lane per (ray, rect): minimum over the rect's four sides in registers, then over the rects below
    constexpr int NT = 3 * C::NR;
    for (int base = 0; base < NT; base += C::VW) {
        RR_FOR_LANES(l) {
            const int t = base + l;
            if (t < NT) {
                const int k = t / C::NR, j = t % C::NR;
                int lst = 0;
                V2<R> a, b; // ray start, end
                if (k == 0) { // back-mid -> front-mid ; front = RIGHT side, back = LEFT side
                    Seg<R> fr = robot_side(A, ridx, 0), bk = robot_side(A, ridx, 2);
                    b = { (fr.a.x + fr.b.x) / (R)2, (fr.a.y + fr.b.y) / (R)2 };
                    a = { (bk.a.x + bk.b.x) / (R)2, (bk.a.y + bk.b.y) / (R)2 };
                } else if (k == 1) { a = robot_corner(A, ridx, BL); b = robot_corner(A, ridx, TR); }
                else { a = robot_corner(A, ridx, TL); b = robot_corner(A, ridx, BR); }
                R mr, cr;
                slope_yint<R>(a, b, mr, cr, lst);
                R bf = inf_<R>(), bb = inf_<R>();
                const R hx = sp.W / (R)2, hy = sp.H / (R)2;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    Seg<R> side;
                    R ms, cs;
                    if (j < C::NR - 1) {
                        const int ro = j < ridx ? j : j + 1;
                        side = robot_side(A, ro, s);
                        ms = A.sm[ro][s]; cs = A.sc[ro][s];
                    } else { // rect_walls = FloatRect(0, W, 0, H) (RR_EnvBase.py:74): corner = centre + (+-W/2, +-H/2)
                        const int ca = side_a(s), cb = side_b(s);
                        side = { { hx + ((ca & 1) ? hx : -hx), hy + ((ca & 2) ? hy : -hy) },
                                 { hx + ((cb & 1) ? hx : -hx), hy + ((cb & 2) ? hy : -hy) } };
                        // get_slope_yint of the wall sides, constant: RIGHT (W,0)->(W,H) +inf/-inf; TOP (0,0)->(W,0) 0/0;
                        // LEFT (0,H)->(0,0) -inf/+inf; BOTTOM (W,H)->(0,H) -0.0 / H
                        ms = s == 0 ? inf_<R>() : s == 2 ? -inf_<R>() : s == 1 ? (R)0 : (R)-0.0;
                        cs = s == 0 ? -inf_<R>() : s == 2 ? inf_<R>() : s == 1 ? (R)0 : sp.H;
                    }
                    V2<R> I = intersect_mb<R>(ms, cs, side.a.x, mr, cr, a.x);
                    R de = dist<R>(I, b), ds = dist<R>(I, a);
                    if (de <= ds && de < bf) bf = de;
                    if (ds <= de && ds < bb) bb = ds;
                }
                A.u.lidar[0][t] = bf;
                A.u.lidar[1][t] = bb;
            }
        }
    }
    RR_SYNC();
    for (int base = 0; base < 6; base += C::VW) {
        RR_FOR_LANES(l) { // min over the rects: one lane per (ray, direction)
            const int t = base + l;
            if (t < 6) {
                const R *src = (t & 1) ? A.u.lidar[1] : A.u.lidar[0];
                const int k = t >> 1;
                R best = inf_<R>();
                for (int q = 0; q < C::NR; q++) {
                    R v = src[k * C::NR + q];
                    if (v < best) best = v;
                }
                A.lid[t] = py_min<R>(best, (R)150);
            }
        }
    }
    RR_SYNC();
    R lid[6]; // front, back per ray
    for (int k = 0; k < 6; k++) lid[k] = A.lid[k];
    V2<R> rc = { A.p.rcx[ridx], A.p.rcy[ridx] }, bc = { A.p.bcx[bidx], A.p.bcy[bidx] };
    V2<R> good = { sp.W, sp.H }, bad = { (R)0, (R)0 };
    R ball_angle = angle_degrees<R>(rc, bc, st);
    R ball_dist = py_min<R>(dist<R>(rc, bc), (R)150);
    R goal_angle = angle_degrees<R>(rc, good, st);
    R bot_angle = A.p.rrot[ridx];
    R dbad = dist<R>(rc, bad), dgood = dist<R>(rc, good);
    R goal_dist = (dgood <= dbad) ? py_min<R>(dgood, (R)390) : (R)-1 * py_min<R>(dbad, (R)390);
    bool ball_neg = bidx >= C::NBP;
    if ((team == 1 && ball_neg) || (team == -1 && !ball_neg)) {
        goal_dist *= (R)-1;
        ball_angle = py_mod<R>(ball_angle + (R)180, (R)360);
        goal_angle = py_mod<R>(goal_angle + (R)180, (R)360);
        bot_angle = py_mod<R>(bot_angle + (R)180, (R)360);
    }
    // The row leaves as ONE lane-strided run per arena (consecutive arenas of a wavefront are consecutive rows: 2 store
    // instructions per wavefront for eight 44-B rows instead of 11 single-lane ones): lane 0 parks the values in A.lid -- every
    // lane holds its copy of the six minima by now -- and the arena's lanes write them out side by side.
    constexpr bool STAGED = RR_GPU && RR_OBS_STAGE && sizeof(A.lid) >= 11 * sizeof(O);
    if constexpr (STAGED) RR_SYNC();
    O *row = STAGED ? reinterpret_cast<O *>(&A.lid[0]) : out;
    if (RR_IS_LANE0) {
        row[0] = (O)bot_angle; row[1] = (O)ball_angle; row[2] = (O)ball_dist; row[3] = (O)goal_angle; row[4] = (O)goal_dist;
        // lidar_front, front_l, front_r, back, back_l, back_r ; ray1 = (front_l, back_r), ray2 = (front_r, back_l)
        row[5] = (O)lid[0]; row[6] = (O)lid[2]; row[7] = (O)lid[4]; row[8] = (O)lid[1]; row[9] = (O)lid[5]; row[10] = (O)lid[3];
    }
    RR_SYNC();
    if constexpr (STAGED) {
        for (int base = 0; base < 11; base += C::VW) { RR_FOR_LANES(l) { if (base + l < 11) out[base + l] = row[base + l]; } }
        RR_SYNC();
    }
    return true;
}

// Both teams' default observations (own robot 0, positive ball 0) in ONE pass: the 2 x 3 x NR (ray, rect) lidar tasks share
// their rounds, and the two scalar tails (angles, distances) run side by side in lanes 0 and 1 instead of one after the
// other.  Same per-task arithmetic as observe().  The second team's candidates live in per-sub-step ball scratch that is
// dead by now (bfx..pfy, exc).
template <class C, typename O>
RR_HDN void observe_both(Arena<C> &A, const SimParams<typename C::Real> &sp, O *out_h, O *out_g, int &st) {
    using R = typename C::Real;
    static_assert(C::NRH > 0 && C::NRG > 0, "two teams");
    static_assert(6 * C::NR <= 4 * C::NB && 6 <= C::NB, "the second team's lidar scratch aliases bfx..pfy / exc");
    static_assert(offsetof(ArenaBody<C>, pfy) == offsetof(ArenaBody<C>, bfx) + 3 * C::NB * sizeof(R), "bfx, bfy, pfx, pfy are contiguous");
    constexpr int NT1 = 3 * C::NR;
    // [team][front|back][ray, rect] candidates and the two teams' capped minima, as SELECTED addresses: a runtime-indexed array of
    // pointers lands in scratch memory -- 48 B per lane written by every wavefront (that was the kernel's whole "scratch" and
    // 256 B of HBM write traffic per arena) and turns the accesses into flat loads / stores
    auto cand = [&](int tm, int fb) -> R * { return tm == 0 ? (fb == 0 ? &A.u.lidar[0][0] : &A.u.lidar[1][0]) : (fb == 0 ? &A.bfx[0] : &A.bfx[0] + NT1); };
    auto lids = [&](int tm) -> R * { return tm == 0 ? &A.lid[0] : &A.exc[0]; };
    RR_FOR_LANES(l) { if (l < C::NR) A.irot[l] = (R)NAN; } // the candidates below overwrite the aliased inner-square offsets
    RR_SYNC();
    ensure_sides(A);
    for (int base = 0; base < 2 * NT1; base += C::VW) {
        RR_FOR_LANES(l) {
            const int t2 = base + l;
            if (t2 < 2 * NT1) {
                const int tm = t2 / NT1, t = t2 % NT1, ridx = tm == 0 ? 0 : C::NRH;
                const int k = t / C::NR, j = t % C::NR;
                int lst = 0;
                V2<R> a, b; // ray start, end
                if (k == 0) { // back-mid -> front-mid ; front = RIGHT side, back = LEFT side
                    Seg<R> fr = robot_side(A, ridx, 0), bk = robot_side(A, ridx, 2);
                    b = { (fr.a.x + fr.b.x) / (R)2, (fr.a.y + fr.b.y) / (R)2 };
                    a = { (bk.a.x + bk.b.x) / (R)2, (bk.a.y + bk.b.y) / (R)2 };
                } else if (k == 1) { a = robot_corner(A, ridx, BL); b = robot_corner(A, ridx, TR); }
                else { a = robot_corner(A, ridx, TL); b = robot_corner(A, ridx, BR); }
                R mr, cr;
                slope_yint<R>(a, b, mr, cr, lst);
                R bf = inf_<R>(), bb = inf_<R>();
                const R hx = sp.W / (R)2, hy = sp.H / (R)2;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    Seg<R> side;
                    R ms, cs;
                    if (j < C::NR - 1) {
                        const int ro = j < ridx ? j : j + 1;
                        side = robot_side(A, ro, s);
                        ms = A.sm[ro][s]; cs = A.sc[ro][s];
                    } else { // rect_walls = FloatRect(0, W, 0, H) (RR_EnvBase.py:74), see observe()
                        const int ca = side_a(s), cb = side_b(s);
                        side = { { hx + ((ca & 1) ? hx : -hx), hy + ((ca & 2) ? hy : -hy) },
                                 { hx + ((cb & 1) ? hx : -hx), hy + ((cb & 2) ? hy : -hy) } };
                        ms = s == 0 ? inf_<R>() : s == 2 ? -inf_<R>() : s == 1 ? (R)0 : (R)-0.0;
                        cs = s == 0 ? -inf_<R>() : s == 2 ? inf_<R>() : s == 1 ? (R)0 : sp.H;
                    }
                    V2<R> I = intersect_mb<R>(ms, cs, side.a.x, mr, cr, a.x);
                    R de = dist<R>(I, b), ds = dist<R>(I, a);
                    if (de <= ds && de < bf) bf = de;
                    if (ds <= de && ds < bb) bb = ds;
                }
                cand(tm, 0)[t] = bf;
                cand(tm, 1)[t] = bb;
            }
        }
    }
    RR_SYNC();
    for (int base = 0; base < 12; base += C::VW) {
        RR_FOR_LANES(l) { // min over the rects: one lane per (team, ray, direction)
            const int t2 = base + l;
            if (t2 < 12) {
                const int tm = t2 / 6, t = t2 % 6;
                const R *src = cand(tm, t & 1);
                const int k = t >> 1;
                R best = inf_<R>();
                for (int q = 0; q < C::NR; q++) {
                    R v = src[k * C::NR + q];
                    if (v < best) best = v;
                }
                lids(tm)[t] = py_min<R>(best, (R)150);
            }
        }
    }
    RR_SYNC();
    uint64_t any_div0 = 0;
    constexpr bool STAGED = RR_GPU && RR_OBS_STAGE && sizeof(A.lid) >= 11 * sizeof(O) && sizeof(A.exc) >= 11 * sizeof(O);
    RR_FOR_LANES(l) { // the scalar tails of the two teams, side by side
        int lst = 0;
        if (l < 2) {
            const int ridx = l == 0 ? 0 : C::NRH;
            O *out = l == 0 ? out_h : out_g;
            R lid[6]; // (by value: the team's staged output row reuses this very array below)
            for (int k = 0; k < 6; k++) lid[k] = lids(l)[k];
            V2<R> rc = { A.p.rcx[ridx], A.p.rcy[ridx] }, bc = { A.p.bcx[0], A.p.bcy[0] };
            V2<R> good = { sp.W, sp.H }, bad = { (R)0, (R)0 };
            R ball_angle = angle_degrees<R>(rc, bc, lst);
            R ball_dist = py_min<R>(dist<R>(rc, bc), (R)150);
            R goal_angle = angle_degrees<R>(rc, good, lst);
            R bot_angle = A.p.rrot[ridx];
            R dbad = dist<R>(rc, bad), dgood = dist<R>(rc, good);
            R goal_dist = (dgood <= dbad) ? py_min<R>(dgood, (R)390) : (R)-1 * py_min<R>(dbad, (R)390);
            if (l == 1) { // GRUMPY looking at a positive ball (RR_Observers.py:386-392)
                goal_dist *= (R)-1;
                ball_angle = py_mod<R>(ball_angle + (R)180, (R)360);
                goal_angle = py_mod<R>(goal_angle + (R)180, (R)360);
                bot_angle = py_mod<R>(bot_angle + (R)180, (R)360);
            }
            // staged (see observe()): each team's row is parked in the array its minima came from -- A.lid / A.exc, read into
            // registers above by this very lane -- and leaves as lane-strided runs below
            O *row = STAGED ? reinterpret_cast<O *>(lids(l)) : out;
            row[0] = (O)bot_angle; row[1] = (O)ball_angle; row[2] = (O)ball_dist; row[3] = (O)goal_angle; row[4] = (O)goal_dist;
            row[5] = (O)lid[0]; row[6] = (O)lid[2]; row[7] = (O)lid[4]; row[8] = (O)lid[1]; row[9] = (O)lid[5]; row[10] = (O)lid[3];
        }
        RR_VOTE(any_div0, l, (lst & ST_DIV0) != 0);
    }
    if (any_div0) st |= ST_DIV0;
    RR_SYNC();
    if constexpr (STAGED) {
        for (int base = 0; base < 22; base += C::VW) {
            RR_FOR_LANES(l) {
                const int k = base + l;
                if (k < 11) out_h[k] = reinterpret_cast<const O *>(lids(0))[k];
                else if (k < 22) out_g[k - 11] = reinterpret_cast<const O *>(lids(1))[k - 11];
            }
        }
        RR_SYNC();
    }
}

// ------------------------------------------------------------------------------------------------ derived data, clean poses
// scratch that is a pure function of the persistent record: corner offsets of every pose in it
template <class C> RR_HD void derive(Arena<C> &A, const SimParams<typename C::Real> &sp) {
    using R = typename C::Real;
    RR_FOR_LANES(l) {
        if (l < C::NR) {
#ifndef RR_REL_IN_RECORD
            corners_for<R>(A.p.rrot[l], (R)10, (R)20, sp.rob_cdist, RR_REL(A)[l]);
#endif
            A.irot[l] = (R)NAN; // inner-square offsets are built lazily by the first narrow phase that needs them
        }
    }
    if (RR_IS_LANE0) A.sides_ok = 0;
    RR_SYNC();
}
// "clean" robot pose: centre exactly (x,y), edges re-derived like the rotation setter, no history
template <class C> RR_HD void robot_set_clean_lane(Arena<C> &A, const SimParams<typename C::Real> &sp, int r,
                                                   typename C::Real x, typename C::Real y, typename C::Real rot) {
    using R = typename C::Real;
    FR<R> f;
    f.cx = x; f.cy = y; f.rot = norm360<R>(rot);
    corners_for<R>(f.rot, (R)10, (R)20, sp.rob_cdist, f.rel);
    fr_edges_from_rel<R>(f);
    store_robot(A, r, f);
    A.p.px[r] = (R)NAN; A.p.py[r] = (R)NAN; A.p.prot[r] = (R)NAN;
    A.i.mc[r] = 0; A.i.thl[r] = 0; A.i.thr[r] = 0;
    A.irot[r] = (R)NAN;
}
template <class C> RR_HD void ball_set_clean_lane(Arena<C> &A, int b, typename C::Real x, typename C::Real y,
                                                  typename C::Real vx, typename C::Real vy) {
    using R = typename C::Real;
    A.p.bcx[b] = x; A.p.bcy[b] = y;
    A.p.bl[b] = x - (R)7; A.p.brt[b] = x + (R)7; A.p.bt[b] = y - (R)7; A.p.bb[b] = y + (R)7;
    A.p.bvx[b] = vx; A.p.bvy[b] = vy;
}

// ------------------------------------------------------------------------------------------------ reset (RR_EnvBase.py:155-216)
// Counter-based RNG: Philox4x32-10, key = seed, counter = (global arena id, episode, draw/4).
RR_HD void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
struct Rng { uint64_t seed, arena, episode; uint32_t draw; };
RR_HD uint32_t rng_u32(Rng &g) {
    uint32_t c[4] = { (uint32_t)g.arena, (uint32_t)(g.arena >> 32), (uint32_t)g.episode, g.draw >> 2 };
    philox4x32(c, (uint32_t)g.seed, (uint32_t)(g.seed >> 32));
    uint32_t v = c[g.draw & 3];
    g.draw++;
    return v;
}
RR_HD long rng_randint(Rng &g, long a, long b) {
    uint64_t n = (uint64_t)(b - a + 1);
    return a + (long)(((uint64_t)rng_u32(g) * n) >> 32);
}
struct IRect { long l, t, w, h; };
template <typename R> RR_HD IRect irect(R l, R t, R r, R b) { // pygame.Rect(l, t, r-l, b-t): C-int truncation
    IRect q = { (long)l, (long)t, (long)(r - l), (long)(b - t) };
    return q;
}
RR_HD bool icollide(IRect a, IRect b) { return a.l < b.l + b.w && a.t < b.t + b.h && a.l + a.w > b.l && a.t + a.h > b.t; }
template <class C> RR_HD IRect robot_irect(const Arena<C> &A, int r) { return irect(A.p.rl[r], A.p.rt[r], A.p.rrt[r], A.p.rb[r]); }
template <class C> RR_HD IRect ball_irect(const Arena<C> &A, int b) { return irect(A.p.bl[b], A.p.bt[b], A.p.brt[b], A.p.bb[b]); }

template <class C>
RR_HDN void reset_arena(Arena<C> &A, const SimParams<typename C::Real> &sp, uint64_t arena_gid, uint64_t episode, int &st) {
    using R = typename C::Real;
    Rng g = { sp.seed, arena_gid, episode, 0 };
    const int MAX_TRY = 4096;
    // sprite.on_reset: robots re-init in place at rot 90 / 270, balls stop (RR_Robot.py:87-88, RR_Ball.py:70-76)
    RR_FOR_LANES(l) {
        if (l < C::NR) robot_set_clean_lane(A, sp, l, A.p.rcx[l], A.p.rcy[l], l < C::NRH ? (R)90 : (R)-90);
        if (l < C::NB) { A.p.bvx[l] = (R)0; A.p.bvy[l] = (R)0; }
        if (l == 0) { A.i.step = 0; A.i.episode = (int32_t)episode; A.i.ep_len = 0; A.i.fault = 0; A.i.fzp = 0; A.p.acc[0] = (R)0; A.p.acc[1] = (R)0; }
    }
    RR_SYNC();
    // robots not yet placed still block at their old pose (RR_EnvBase.py:157-158 writes sprite attrs only)
    for (int i = 0; i < C::NR; i++) {
        for (int t = 0;; t++) {
            long x = rng_randint(g, 40 * 2, (long)sp.W - 40 * 2);
            long y = rng_randint(g, 20 * 2, (long)sp.H - 20 * 2);
            long rot = rng_randint(g, 0, 360);
            RR_FOR_LANES(l) {
                if (l == 0) robot_set_clean_lane(A, sp, i, (R)x, (R)y, (R)rot);
            }
            RR_SYNC();
            IRect me = robot_irect(A, i);
            int hits = 0;
            for (int j = 0; j < C::NR; j++) hits += icollide(me, robot_irect(A, j)) ? 1 : 0;
            if (hits <= 1) break;
            if (t >= MAX_TRY) { st |= ST_RESET_GAVE_UP; break; }
        }
    }
    RR_FOR_LANES(l) {
        if (l < C::NB) ball_set_clean_lane(A, l, (R)-1000, (R)-1000, (R)0, (R)0); // RR_EnvBase.py:183-184
    }
    RR_SYNC();
    IRect goal_h = { (long)(sp.W - (R)240), (long)(sp.H - (R)240), 240, 240 }, goal_g = { 0, 0, 240, 240 }; // RR_Goal.py:30-35
    for (int i = 0; i < C::NB; i++) {
        for (int t = 0;; t++) {
            long x = rng_randint(g, 40, (long)sp.W - 40);
            long y = rng_randint(g, 40, (long)sp.H - 40);
            RR_FOR_LANES(l) {
                if (l == 0) ball_set_clean_lane(A, i, (R)x, (R)y, (R)0, (R)0);
            }
            RR_SYNC();
            IRect me = ball_irect(A, i);
            int hits = (icollide(me, goal_h) ? 1 : 0) + (icollide(me, goal_g) ? 1 : 0);
            for (int j = 0; j < C::NR; j++) hits += icollide(me, robot_irect(A, j)) ? 1 : 0;
            for (int j = 0; j < C::NB; j++) hits += icollide(me, ball_irect(A, j)) ? 1 : 0;
            if (hits <= 1) break;
            if (t >= MAX_TRY) { st |= ST_RESET_GAVE_UP; break; }
        }
    }
    derive(A, sp);
}

// ------------------------------------------------------------------------------------------------ whole step (RR_EnvBase.py:260-297)
template <typename R> RR_HD bool is_done(int step, const SimParams<R> &sp) {
    return sp.time_limit ? (step >= sp.game_len) : (step > sp.game_len); // TimeLimit wrapper vs raw :555-559
}
// Output slots of one arena: uniform base pointers + the arena index.  The per-arena addresses are formed at the point of
// use -- carried as ready-made pointers they cost ~16 VGPRs across the whole sub-step loop and pushed the kernel into spills.
template <typename O> struct StepOut {
    O *obs_base, *obs_g_base, *reward_base, *reward_g_base;
    uint8_t *done_base;
    int32_t *status_base;
    uint32_t *snap_base = nullptr; // per-arena scratch in HBM for the fixed-point check (null: shortcuts off)
    int32_t *isnap_base = nullptr;
    int arena = 0;
    int snap_stride = 0, isnap_stride = 0; // words per arena
    RR_HD O *obs() const { return obs_base + (size_t)arena * 11; }
    RR_HD O *obs_g() const { return obs_g_base ? obs_g_base + (size_t)arena * 11 : nullptr; }
    RR_HD O *reward() const { return reward_base + arena; }
    RR_HD O *reward_g() const { return reward_g_base ? reward_g_base + arena : nullptr; }
    RR_HD uint8_t *done() const { return done_base + arena; }
    RR_HD int32_t *status() const { return status_base ? status_base + arena : nullptr; }
    RR_HD uint32_t *snap() const { return snap_base ? snap_base + (size_t)arena * snap_stride : nullptr; }
    RR_HD int32_t *isnap() const { return isnap_base + (size_t)arena * isnap_stride; }
};

// Fixed point of the sub-step map.  A stuck arena -- a ball squeezed between two robots, a robot pushing a ball into
// a wall -- runs the push, all 10 resolve passes and the undo loop in every sub-step, and the undo puts everything
// back: the sub-step maps the state onto itself, 12 times per step, at ~150x the cost of a quiet sub-step (one such
// arena used to set the duration of a whole 65,536-arena launch).  A sub-step is a deterministic function of the
// persistent state (the P record and the move counters), the thrust (fixed during a step), `prev_moved` and the
// frame-begin poses ax/ay/arot of the robots in it (the pose-ring update at the top of the next sub-step reads them);
// so if all of that after sub-step k is bit-identical to what it was after sub-step k-1, sub-step k+1 gets the very
// input sub-step k got, and by induction every later sub-step of this step reproduces the same state (and ORs the
// same bits into `naughty` / `st`): the loop can stop.  Exact, not approximate.
// The comparison is bitwise (NaN-safe), lane-strided, against a snapshot kept in a per-arena scratch buffer in HBM; it is
// only made after sub-steps that ran the expensive contact paths, so quiet arenas never touch that buffer.
// did the last move of any robot in `mask` end in the 0.5-px wall clamp (RR_Robot.py:195-203)?
template <class C> RR_HD bool robots_clamped(const Arena<C> &A, uint32_t mask) {
    uint64_t any = 0;
    RR_FOR_LANES(l) {
        const bool c = (l < C::NR) && ((mask >> l) & 1u) && (A.wm[l] & 2);
        RR_VOTE(any, l, c);
    }
    return any != 0;
}
// cheap necessary condition, from LDS only: every robot ends the sub-step on its frame-begin pose (undone, blocked or idle)
template <class C> RR_HD bool robots_unmoved(const Arena<C> &A) {
    uint64_t any_moved = 0;
    RR_FOR_LANES(l) {
        const bool mv = (l < C::NR) && !(A.p.rcx[l] == A.ax[l] && A.p.rcy[l] == A.ay[l] && A.p.rrot[l] == A.arot[l]);
        RR_VOTE(any_moved, l, mv);
    }
    return !any_moved;
}
// Compares the arena with the snapshot of one sub-step earlier, word by word, and refreshes the snapshot.  Out: the
// robots whose pose / pose history / move counter changed (chg_r), the robots whose AABB edges changed (chg_re: the edges
// are kept incrementally, MyUtils.py:141-148, and a move + undo does not always give them back to the last bit), the balls
// that changed, and whether any frame-begin pose ax/ay/arot did.  Without a valid snapshot everything counts as changed.
template <class C>
RR_HD void snapshot_compare_update(const Arena<C> &A, uint32_t *snap, int32_t *isnap, bool have, uint32_t &chg_r, uint32_t &chg_re,
                                   uint32_t &chg_b, bool &ax_diff) {
    using R = typename C::Real;
    constexpr int WR = (int)(sizeof(R) / 4);
    constexpr int NW = (int)(sizeof(typename Arena<C>::P) / 4);   // the persistent reals ...
    constexpr int NA = 3 * C::NR * WR;                            // ... + ax, ay, arot: the frame-begin poses the next sub-step's ring update reads
    static_assert(Arena<C>::SNAP_WORDS == NW + NA, "snapshot buffer stride");
    static_assert(offsetof(ArenaBody<C>, ay) == offsetof(ArenaBody<C>, ax) + C::NR * sizeof(R) &&
                  offsetof(ArenaBody<C>, arot) == offsetof(ArenaBody<C>, ax) + 2 * C::NR * sizeof(R), "ax, ay, arot are contiguous");
    const uint32_t *p = reinterpret_cast<const uint32_t *>(&A.p);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(&A.ax[0]);
    uint32_t cr = 0, cre = 0, cb = 0;
    uint64_t any_ax = 0, m = 0;
    constexpr int NE = 2 * C::NR + C::NB; // bits: robots (core), robots (edges), balls
    RR_LANE_VAR(uint32_t, mine_of);
    // each lane collects the entities its words belong to; the masks are then OR-ed over the lanes bit by bit (ballots)
    RR_FOR_LANES(l) {
        uint32_t mine = 0; // bits 0..NR-1 robots (core fields), NR..2NR-1 robots (edge fields 2..5), 2NR.. balls
        bool axd = false;
        for (int k = l; k < NW + NA; k += C::VW) {
            const uint32_t v = k < NW ? p[k] : q[k - NW];
            const bool d = !have || snap[k] != v;
            snap[k] = v;
            if (k < NW) {
                const int idx = k / WR; // index of the real inside P: 10 robot fields x NR, 8 ball fields x NB, acc[4]
                if (idx < 10 * C::NR) {
                    const int fld = idx / C::NR; // rcx rcy | rl rrt rt rb | rrot px py prot
                    mine |= d ? (1u << (idx % C::NR + ((fld >= 2 && fld <= 5) ? C::NR : 0))) : 0u;
                } else if (idx < 10 * C::NR + 8 * C::NB) mine |= d ? (1u << (2 * C::NR + (idx - 10 * C::NR) % C::NB)) : 0u;
#if RR_CARRY
                else if (idx >= 10 * C::NR + 8 * C::NB + 4) mine |= d ? (1u << (2 * C::NR + C::NB - 1)) : 0u; // the scratch rect sits on the last ball
#endif
            } else {
                axd = axd | d;
            }
        }
        for (int k = l; k < 2 * C::NR; k += C::VW) { // move counters, then how each robot's last move met the walls
            const int32_t v = k < C::NR ? A.i.mc[k] : A.wm[k - C::NR];
            mine |= (!have || isnap[k] != v) ? (1u << (k % C::NR)) : 0u;
            isnap[k] = v;
        }
        RR_VOTE(any_ax, l, axd);
        RR_LV(mine_of, l) = mine;
    }
    for (int e = 0; e < NE; e++) {
        m = 0;
        RR_FOR_LANES(l) { RR_VOTE(m, l, (RR_LV(mine_of, l) >> e) & 1u); }
        if (m) { if (e < C::NR) cr |= 1u << e; else if (e < 2 * C::NR) cre |= 1u << (e - C::NR); else cb |= 1u << (e - 2 * C::NR); }
    }
    chg_r = cr; chg_re = cre; chg_b = cb; ax_diff = any_ax != 0;
}

// Island freeze ACROSS steps.  A stuck island usually stays stuck for many steps (a robot driving a ball into a wall keeps
// driving), and within a step it costs two expensive sub-steps to find it again (snapshot, compare).  Nothing between two
// steps touches an island except its robots' new thrust -- the step-begin hooks only take copies for the rewards -- so an
// island that is still frozen when a step ends, whose robots are given the same thrust again, is the same fixed point of the
// same map: the next step starts with it frozen.  What the frozen sub-steps need travels in two ints of the record
// (Arena::I::fzp, fexc): the island (robots 4 bits, balls 8), how each of its robots' moves met the walls (wm, 2 bits each),
// the NaughtyBots members and status bits one computed sub-step of the island produces (they are per step: a frozen
// sub-step ORs them in), and an upper bound of its balls' excursions.  Everything that rewrites an arena from outside
// (reset, rr_set_state, rr_set_poses, the goal-scoring side kernel) clears the word.  Same results with RR_NO_MEMO=1.
RR_HD uint32_t fz_pack_bits(uint32_t naughty, int st) { return (naughty & 0xFu) | ((uint32_t)(st & 63) << 4) | ((uint32_t)((st >> 8) & 1) << 10); }
RR_HD uint32_t fz_bits_naughty(uint32_t pk) { return pk & 0xFu; }
RR_HD int fz_bits_status(uint32_t pk) { return (int)((pk >> 4) & 63u) | (int)(((pk >> 10) & 1u) << 8); }
RR_HD float bits_float(int32_t v) { float f; __builtin_memcpy(&f, &v, 4); return f; }
RR_HD int32_t float_bits(float f) { int32_t v; __builtin_memcpy(&v, &f, 4); return v; }

// Budgeted step (opt-in, rr_config.step_budget_clocks / rr_set_step_budget; BUDGET = false compiles all of it out).
// A launch ends when its slowest arena does, and under a contact-rich policy one arena in 65,536 needs 10-20x the
// median step (ten resolve passes + undo in every sub-step) while two thirds of the chip idle behind it.  With a budget, an
// arena whose step is over the budget at the end of an expensive sub-step PARKS: the few values a sub-step boundary
// carries besides the persistent record -- sub-step index, surviving-move mask, the step's NaughtyBots / status
// accumulators, the frozen island, the frame-begin poses (ax, ay, arot), the step-begin copies the rewards read (psx, psy,
// dist_sum0), how each robot's last move met the walls (wm), the frozen balls' excursions -- go to its slot of a side buffer,
// bit 31 of the record's `fzp` word marks it, the call reports ST_NOT_READY for it (reward 0, done 0, observation rows NOT
// written) and the other arenas of its wavefront run on undisturbed.  The next call resumes it at that sub-step boundary and
// ignores the action it is given; everything else in LDS is either rewritten at the top of every sub-step (frame hooks)
// or a cache that derive() / the lazy builders reproduce bit for bit.  So each arena's trajectory, as a function of the actions
// it ACCEPTED, is the synchronous mode's bit for bit (tests/test_budgeted_step.py: emulation parking at arbitrary boundaries,
// GPU at budgets from 1 clock up).  Whatever rewrites an arena from outside (reset, rr_set_state, rr_set_poses) clears the
// word, and with it the parked step.
constexpr int32_t FZP_PARKED = (int32_t)0x80000000; // Arena::I::fzp while the arena is parked mid-step

template <class C>
RR_HD void park_save(Arena<C> &A, uint32_t *buf, int f_next, uint32_t prev_moved, uint32_t naughty, int st, const Hit &fz,
                     uint32_t fz_bits, bool snap_valid, uint32_t snap_moved, typename C::Real dist_sum0, const MidState &mid) {
    using R = typename C::Real;
    constexpr int NR = C::NR, NB = C::NB;
    int32_t *pi = reinterpret_cast<int32_t *>(buf);
    R *pr = reinterpret_cast<R *>(buf + Arena<C>::PARK_INTS);
    RR_FOR_LANES(l) {
        if (l < NR) { pr[l] = A.ax[l]; pr[NR + l] = A.ay[l]; pr[2 * NR + l] = A.arot[l]; pr[3 * NR + l] = A.psx[l]; pr[4 * NR + l] = A.psy[l]; }
        if (l < NB) {
            pr[5 * NR + 1 + l] = A.exc[l];
            if (mid.phase) { // mid-sub-step: the frame-begin picture of the balls and what the push left in them
                R *q = pr + 5 * NR + 1 + NB;
                q[l] = A.bfx[l]; q[NB + l] = A.bfy[l]; q[2 * NB + l] = A.pfx[l]; q[3 * NB + l] = A.pfy[l];
                pi[20 + l] = A.bmass[l];
            }
        }
        if (l == 0) {
            pi[10] = mid.phase; pi[11] = mid.count; pi[12] = (int32_t)mid.bots_moved; pi[13] = (int32_t)mid.balls_moved;
            pi[14] = (int32_t)mid.n_sub; pi[15] = mid.st_sub; pi[16] = mid.work; pi[17] = (int32_t)mid.hit.r; pi[18] = (int32_t)mid.hit.b;
            uint32_t wmb = 0;
            for (int r = 0; r < NR; r++) wmb |= (uint32_t)(A.wm[r] & 3) << (2 * r);
            pr[5 * NR] = dist_sum0;
            pi[0] = f_next; pi[1] = (int32_t)prev_moved; pi[2] = (int32_t)naughty; pi[3] = st; pi[4] = (int32_t)fz.r; pi[5] = (int32_t)fz.b;
            pi[6] = (int32_t)fz_bits; pi[7] = snap_valid ? 1 : 0; pi[8] = (int32_t)snap_moved; pi[9] = (int32_t)wmb;
            A.i.fzp = FZP_PARKED;
        }
    }
    RR_SYNC();
}
template <class C>
RR_HD void park_load(Arena<C> &A, const uint32_t *buf, int &f_next, uint32_t &prev_moved, uint32_t &naughty, int &st, Hit &fz,
                     uint32_t &fz_bits, int &snap_at, uint32_t &snap_moved, typename C::Real &dist_sum0, MidState &mid) {
    using R = typename C::Real;
    constexpr int NR = C::NR, NB = C::NB;
    static_assert(NR <= 16, "wm bytes travel as 2 bits per robot in one word");
    const int32_t *pi = reinterpret_cast<const int32_t *>(buf);
    const R *pr = reinterpret_cast<const R *>(buf + Arena<C>::PARK_INTS);
    const uint32_t wmb = (uint32_t)pi[9];
    RR_FOR_LANES(l) {
        if (l < NR) {
            A.ax[l] = pr[l]; A.ay[l] = pr[NR + l]; A.arot[l] = pr[2 * NR + l]; A.psx[l] = pr[3 * NR + l]; A.psy[l] = pr[4 * NR + l];
            A.wm[l] = (uint8_t)((wmb >> (2 * l)) & 3u);
        }
        if (l < NB) {
            A.exc[l] = pr[5 * NR + 1 + l];
            if (pi[10]) {
                const R *q = pr + 5 * NR + 1 + NB;
                A.bfx[l] = q[l]; A.bfy[l] = q[NB + l]; A.pfx[l] = q[2 * NB + l]; A.pfy[l] = q[3 * NB + l];
                A.bmass[l] = pi[20 + l];
            }
        }
    }
    mid.phase = pi[10]; mid.count = pi[11]; mid.bots_moved = (uint32_t)pi[12]; mid.balls_moved = (uint32_t)pi[13];
    mid.n_sub = (uint32_t)pi[14]; mid.st_sub = pi[15]; mid.work = pi[16]; mid.hit.r = (uint32_t)pi[17]; mid.hit.b = (uint32_t)pi[18];
    dist_sum0 = pr[5 * NR];
    f_next = pi[0]; prev_moved = (uint32_t)pi[1]; naughty = (uint32_t)pi[2]; st = pi[3]; fz.r = (uint32_t)pi[4]; fz.b = (uint32_t)pi[5];
    fz_bits = (uint32_t)pi[6]; snap_at = pi[7] ? f_next - 1 : -2; snap_moved = (uint32_t)pi[8];
    RR_SYNC();
}

// actions: this arena's na discrete actions (thrust == nullptr) or 2*na thrust floats
template <class C, typename O, bool BUDGET = false>
RR_HD void step_arena(Arena<C> &A, const SimParams<typename C::Real> &sp, uint64_t arena_gid, const int32_t *actions,
                      const float *thrust, int na, const StepOut<O> &o, const ParkCtx &pk = ParkCtx()) {
    using R = typename C::Real;
    int st = 0;
    R dist_sum0 = (R)0;
    uint32_t naughty = 0;
    uint32_t prev_moved = 0, snap_moved = 0;
    int snap_at = -2; // sub-step whose end state the snapshot holds
    Hit fz = { 0, 0 }; // the frozen island (see substep)
    uint32_t fz_bits = 0; // NaughtyBots members + status bits of one computed sub-step of the frozen island (fz_pack_bits)
    int f0 = 0; // first sub-step to run: 0, or where a parked step goes on
    MidState mid; // (budgeted step: a sub-step parked between two passes of its resolve loop re-enters there)
    int icm = 0;  // parity build: 1 while the scratch rect is implicitly on the last ball (carry_quiet_sweep); 0: A.p.ic holds it
    // (what the packed word holds; and not under the F32State policy: the record is rounded to fp32 between two steps, so the island a
    // step ended with is not bit for bit the one the next step would find)
#ifdef RR_NO_FZP // A/B builds only
    constexpr bool FZP = false;
#else
    constexpr bool FZP = C::NR <= 4 && C::NB <= 8 && !C::MIXED;
#endif
    bool resumed = false;
    RR_T0();
    if constexpr (BUDGET) {
        resumed = A.i.fzp < 0; // uniform per arena (read after load_record's sync)
        if (RR_UNLIKELY(resumed)) {
            park_load(A, pk.buf, f0, prev_moved, naughty, st, fz, fz_bits, snap_at, snap_moved, dist_sum0, mid);
            RR_TRACE("E resumed at sub-step %d\n", f0);
        }
    }
    if (!resumed) {
    // raw mode raises when stepping a finished game (:261-262); with auto_reset the call resets instead
    if (RR_UNLIKELY((sp.time_limit ? (A.i.step >= sp.game_len) : (A.i.step > sp.game_len)) || A.i.fault)) {
        if (sp.auto_reset) {
            reset_arena(A, sp, arena_gid, (uint64_t)(uint32_t)(A.i.episode + 1), st);
            st |= ST_WAS_RESET;
            observe<C, O>(A, sp, 1, -1, -1, o.obs(), st);
            if (o.obs_g()) {
                if (!observe<C, O>(A, sp, -1, -1, -1, o.obs_g(), st)) {
                    for (int base = 0; base < 11; base += C::VW) { RR_FOR_LANES(l) { if (base + l < 11) o.obs_g()[base + l] = (O)NAN; } }
                }
            }
            if (RR_IS_LANE0) {
                *o.reward() = (O)0; *o.done() = 0;
                if (o.reward_g()) *o.reward_g() = (O)0;
                if (o.status()) *o.status() = st;
            }
            return;
        }
        st |= ST_STEP_AFTER_DONE;
        observe<C, O>(A, sp, 1, -1, -1, o.obs(), st);
        if (o.obs_g()) {
            if (!observe<C, O>(A, sp, -1, -1, -1, o.obs_g(), st)) {
                for (int base = 0; base < 11; base += C::VW) { RR_FOR_LANES(l) { if (base + l < 11) o.obs_g()[base + l] = (O)NAN; } }
            }
        }
        if (RR_IS_LANE0) {
            *o.reward() = (O)0; *o.done() = 1;
            if (o.reward_g()) *o.reward_g() = (O)0;
            if (o.status()) *o.status() = st;
        }
        return;
    }
    RR_TR();
    // ---- on_step_begin (:264-265; sprites, then the score keepers RR_ScoreKeepers.py:30-33,119-121,145-147)
    static_assert(C::NBP + C::NR * C::NBP <= 2 * 3 * C::NR, "reward scratch reuses the lidar slots");
    // _calc_ball_dist_sum (RR_ScoreKeepers.py:155-157): one lane per positive ball, summed in list order below
    for (int base = 0; base < C::NBP; base += C::VW) {
        RR_FOR_LANES(l) {
            const int b = base + l;
            if (b < C::NBP) { V2<R> o0 = { (R)0, (R)0 }, c = { A.p.bcx[b], A.p.bcy[b] }; A.u.lidar[1][b] = dist<R>(o0, c); }
        }
    }
    RR_SYNC();
    for (int b = 0; b < C::NBP; b++) dist_sum0 = dist_sum0 + A.u.lidar[1][b];
    RR_SYNC();
    uint64_t m_thr = 0; // robots whose thrust this step changes
    RR_FOR_LANES(l) {
        bool thr_chg = false;
        if (l < C::NR) {
            // rectDblPriorStep = rectDbl.copy(): a fresh 20x40 rect centred at (10,20) moved by the centre setters (MyUtils.py:150-154)
            A.psx[l] = (R)10 + (A.p.rcx[l] - (R)10); A.psy[l] = (R)20 + (A.p.rcy[l] - (R)20);
            if (l < na) { // set_thrust (RR_Robot.py:100-102)
                const int oL = A.i.thl[l], oR = A.i.thr[l];
                if (thrust) {
                    A.i.thl[l] = (int)m_rint(thrust[2 * l]); A.i.thr[l] = (int)m_rint(thrust[2 * l + 1]);
                } else {
                    int a = actions[l];
                    if (a >= 0 && a <= 7) { // Direction table, RR_EnvBase.py:593-602
                        A.i.thl[l] = (a == 0 || a == 3 || a == 5) ? 1 : (a == 4 || a == 7) ? 0 : -1;
                        A.i.thr[l] = (a == 0 || a == 2 || a == 4) ? 1 : (a == 5 || a == 6) ? 0 : -1;
                    }
                }
                thr_chg = (A.i.thl[l] != oL) | (A.i.thr[l] != oR);
            }
        }
        if (l == 0) A.i.step += 1;
        if (FZP) RR_VOTE(m_thr, l, thr_chg);
    }
    if (!thrust) for (int q = 0; q < na && q < C::NR; q++) { int a = actions[q]; if (a < 0 || a > 7) st |= ST_BAD_ACTION; }
    RR_SYNC();
    RR_STAMP(8);
    if (FZP) { // the island the last step ended with, if its robots keep their thrust (see above)
        const uint32_t fzp = o.snap() ? (uint32_t)A.i.fzp : 0u;
        if (RR_UNLIKELY(fzp != 0u) && !((uint32_t)m_thr & fzp & 0xFu)) {
            fz.r = fzp & 0xFu; fz.b = (fzp >> 4) & 0xFFu;
            fz_bits = fzp >> 20;
            const R ex = (R)bits_float(A.i.fexc);
            RR_FOR_LANES(l) {
                if (l < C::NR && ((fz.r >> l) & 1u)) A.wm[l] = (uint8_t)((fzp >> (12 + 2 * l)) & 3u);
                if (l < C::NB && ((fz.b >> l) & 1u)) A.exc[l] = ex;
            }
            RR_SYNC();
            RR_TRACE("E step begins frozen: robots %x balls %x\n", fz.r, fz.b);
        }
    }
    } // !resumed
#if defined(RR_PROFILE_PHASES)
    int dbg_work_ = 0;
    const int dbg_frozen_ = (fz.r | fz.b) ? 1 : 0;
#endif
#pragma unroll 1
    for (int f = f0; f < RR_NUM_SUBSTEPS; f++) { // MOVES_PER_FRAME
        int work = 0;
        Hit hit = { 0, 0 };
        uint32_t n_sub = 0; // what THIS sub-step adds: a freeze keeps it for the frozen sub-steps of later steps
        int st_sub = 0;
        if constexpr (BUDGET) {
            const bool parked_mid = substep<C, true>(A, sp, n_sub, st_sub, prev_moved, work, fz, hit, icm, &mid, &pk);
            if (RR_UNLIKELY(parked_mid)) {
                RR_TRACE("E parked inside sub-step %d before resolve pass %d\n", f, mid.count + 1);
                park_save(A, pk.buf, f, prev_moved, naughty, st, fz, fz_bits, snap_at == f - 1, snap_moved, dist_sum0, mid);
                if (RR_IS_LANE0) {
                    *o.reward() = (O)0; *o.done() = 0;
                    if (o.reward_g()) *o.reward_g() = (O)0;
                    if (o.status()) *o.status() = ST_NOT_READY;
                }
                return;
            }
        } else {
            substep(A, sp, n_sub, st_sub, prev_moved, work, fz, hit, icm);
        }
        naughty |= n_sub; st |= st_sub;
#if defined(RR_PROFILE_PHASES) // diagnostic builds only: the step's contact work in status bits 20-29, "began frozen" in bit 30
        dbg_work_ += work;
#endif
        if (FZP && RR_UNLIKELY(fz.r | fz.b)) { naughty |= fz_bits_naughty(fz_bits); st |= fz_bits_status(fz_bits); } // a frozen sub-step's share
        // An expensive sub-step: has the arena, or the island that made it expensive, stopped changing?  (The snapshot
        // costs a round trip to the arena's HBM record, so it is only taken when a whole-arena fixed point is possible --
        // no robot moved -- or when the sub-step exhausted the resolve loop.)
        if (RR_UNLIKELY(o.snap() && f + 1 < RR_NUM_SUBSTEPS && work >= 3 && ((C::NR > 1 && work >= 12) || robots_unmoved(A)))) {
            uint32_t chg_r, chg_re, chg_b;
            bool ax_diff;
            const bool have = snap_at == f - 1;
#if RR_CARRY
            carry_materialize(A, icm, false); // (the snapshot holds the explicit centre)
#endif
            snapshot_compare_update(A, o.snap(), o.isnap(), have, chg_r, chg_re, chg_b, ax_diff);
            RR_TRACE("E snapshot at %d: have %d chg_r %x edges %x chg_b %x hit r %x b %x moved %x/%x work %d\n", f, (int)have, chg_r, chg_re, chg_b, hit.r, hit.b, prev_moved, snap_moved, work);
            if (have) {
                bool island_ok = (hit.r | hit.b) && !(chg_r & hit.r) && !(chg_b & hit.b) && !((prev_moved | snap_moved) & hit.r) &&
                                 !robots_clamped(A, hit.r);
#if RR_CARRY
                island_ok = island_ok && !(carry_deps<C>(hit.b) & chg_b); // (see substep_phase1: the balls the island's scratch-rect chain starts from)
#endif
                if (!(chg_r | chg_re | chg_b) && !ax_diff && snap_moved == prev_moved) {
                    RR_TRACE("E fixed point after sub-step %d\n", f);
                    if (FZP && island_ok) { fz = hit; fz_bits = fz_pack_bits(n_sub, st_sub); } // for the NEXT step (this one is done)
                    break; // sub-steps f+1.. would reproduce this state bit for bit
                }
                // the island: everything that took part in a hit.  Unchanged since the previous sub-step, its robots'
                // moves undone in both (so neither the ring update nor ax/ay/arot matter to it) -> freeze it.  A robot's
                // AABB edges may keep drifting in the last bits (move + undo, incremental); nothing but the wall test / clamp of
                // its own move reads them.  So a frozen robot keeps making its move and being put back -- on its real edges --
                // and the island is thawed the moment that move meets the walls differently (wm); a move that was clamped
                // (its centre then depends on an edge value) is never frozen.
                // (a one-robot arena too: robot + ball squeezed against a wall is a whole-arena fixed point EXCEPT for those
                // drifting edge bits, which is most of what a trained or chase policy runs into on preset T)
                if (island_ok) {
                    fz = hit;
#ifndef RR_NO_NEIGHBOURS // (A/B builds only)
                    if (hit.r) {
                        const uint32_t nb_ = resting_neighbours(A, hit, chg_b);
#if RR_CARRY
                        if (nb_ && !(carry_deps<C>(hit.b | nb_) & chg_b)) fz.b |= nb_;
#else
                        fz.b |= nb_;
#endif
                    }
#endif
                    RR_TRACE("E freeze robots %x balls %x (hit %x) after sub-step %d\n", fz.r, fz.b, hit.b, f);
                    fz_bits = fz_pack_bits(n_sub, st_sub);
                } else {
                    RR_TRACE("E no freeze: island changed r %x b %x, moved %x, clamped %d\n", chg_r & hit.r, chg_b & hit.b, (prev_moved | snap_moved) & hit.r, (int)robots_clamped(A, hit.r));
                }
            }
            snap_at = f; snap_moved = prev_moved;
        }
        if constexpr (BUDGET) { // over the budget after an expensive sub-step, and more to come: park at this boundary
            if (f + 1 < RR_NUM_SUBSTEPS && RR_UNLIKELY(pk.over(work))) {
                RR_TRACE("E parked after sub-step %d\n", f);
#if RR_CARRY
                carry_materialize(A, icm, false);
#endif
                park_save(A, pk.buf, f + 1, prev_moved, naughty, st, fz, fz_bits, snap_at == f, snap_moved, dist_sum0, mid);
                if (RR_IS_LANE0) {
                    *o.reward() = (O)0; *o.done() = 0;
                    if (o.reward_g()) *o.reward_g() = (O)0;
                    if (o.status()) *o.status() = ST_NOT_READY;
                }
                return;
            }
        }
    }
    if (FZP) { // still frozen: the next step may start that way
        int32_t fzw = 0, fex = 0;
        if (RR_UNLIKELY(fz.r | fz.b)) {
            R emax = (R)0;
            uint32_t wmb = 0;
            for (int b = 0; b < C::NB; b++) if ((fz.b >> b) & 1u) emax = py_max<R>(emax, A.exc[b]);
            for (int r = 0; r < C::NR; r++) if ((fz.r >> r) & 1u) wmb |= (uint32_t)(A.wm[r] & 3) << (2 * r);
            fzw = (int32_t)(fz.r | (fz.b << 4) | (wmb << 12) | (fz_bits << 20));
            fex = float_bits((float)emax * 1.000001f + 1e-30f); // rounded up: a larger reach only thaws earlier
        }
        if (RR_IS_LANE0) { A.i.fzp = fzw; A.i.fexc = fex; }
    } else if (BUDGET) {
        if (RR_IS_LANE0) A.i.fzp = 0; // (configurations without a carried island: the word only ever holds the parked mark)
    }
#if RR_CARRY
    carry_materialize(A, icm, false); // the record carries the explicit centre from step to step (a reset moves the balls, not the rect)
#endif
    substeps_end(A, prev_moved);
    RR_STAMP(9);
    // ---- on_step_end: NaughtyBots, ChasePosBall, PushPosBallsToGoal (SURVEY 3.1 accumulation order)
    // one lane per (robot, positive ball) ChasePosBall term and per positive ball distance; the sums keep list order
    for (int base = 0; base < C::NR * C::NBP; base += C::VW) {
        RR_FOR_LANES(l) {
            const int t = base + l;
            if (t < C::NR * C::NBP) {
                const int r = t / C::NBP, b = t % C::NBP;
                V2<R> rc = { A.p.rcx[r], A.p.rcy[r] }, pc = { A.psx[r], A.psy[r] }, bc = { A.p.bcx[b], A.p.bcy[b] };
                R now = dist<R>(rc, bc), prior = dist<R>(pc, bc);
                (&A.u.lidar[0][0])[C::NBP + t] = ball_in_play(A, b) ? (prior - now) * sp.mult_robot : (R)0; // (a consumed ball: no term)
            }
            if (t < C::NBP) { V2<R> o0 = { (R)0, (R)0 }, c = { A.p.bcx[t], A.p.bcy[t] }; (&A.u.lidar[0][0])[t] = dist<R>(o0, c); }
        }
    }
    RR_SYNC();
    R rew_h = (R)0, rew_g = (R)0;
    for (int r = 0; r < C::NR; r++) if (naughty & (1u << r)) { if (r < C::NRH) rew_h -= (R).005; else rew_g -= (R).005; }
    for (int r = 0; r < C::NR; r++)
        for (int b = 0; b < C::NBP; b++) {
            R term = (&A.u.lidar[0][0])[C::NBP + r * C::NBP + b];
            if (r < C::NRH) rew_h += term; else rew_g += term;
        }
    {
        R s1 = (R)0;
        for (int b = 0; b < C::NBP; b++) s1 = s1 + (&A.u.lidar[0][0])[b];
        R delta = s1 - dist_sum0;
        rew_h += delta * sp.mult_ball;
        rew_g -= delta * sp.mult_ball;
    }
    RR_SYNC(); // observe() reuses the scratch
    RR_STAMP(10);
    const int step_now = A.i.step;
    constexpr int ST_FATAL = ST_BOT_RESOLVE_FAIL | ST_BOT_STUCK | ST_UNDO_MOVE_FAIL | ST_UNDO_FAIL | ST_SAME_SPOT | ST_DIV0;
    const bool faulted = sp.reset_on_fault && (st & ST_FATAL); // the reference raised (or hangs) inside this step
    const bool done = is_done<R>(step_now, sp) || faulted;
    // (the one-pass two-team observation parks the second team's lidar candidates in per-sub-step ball scratch: it needs the room)
    constexpr bool BOTH = C::NRH > 0 && C::NRG > 0 && 6 * C::NR <= 4 * C::NB && 6 <= C::NB;
    if constexpr (BOTH) {
        if (o.obs_g()) observe_both<C, O>(A, sp, o.obs(), o.obs_g(), st);
        else observe<C, O>(A, sp, 1, -1, -1, o.obs(), st);
    } else {
        observe<C, O>(A, sp, 1, -1, -1, o.obs(), st);
        if (o.obs_g()) {
            if (!observe<C, O>(A, sp, -1, -1, -1, o.obs_g(), st)) {
                for (int base = 0; base < 11; base += C::VW) { RR_FOR_LANES(l) { if (base + l < 11) o.obs_g()[base + l] = (O)NAN; } }
            }
        }
    }
    if (RR_IS_LANE0) {
        *o.reward() = (O)rew_h; *o.done() = done ? 1 : 0;
        if (o.reward_g()) *o.reward_g() = (O)rew_g;
        if (o.status()) *o.status() = st | (int)(naughty << 16); // bits 16..: robots NaughtyBots flagged this step
#if defined(RR_PROFILE_PHASES)
        if (o.status()) *o.status() |= ((dbg_work_ > 1023 ? 1023 : dbg_work_) << 20) | (dbg_frozen_ << 30);
#endif
        // episode bookkeeping for logging (the caller sums `score` the same way, Training_DQN_pytorch.py:345-346)
        A.i.ep_len += 1;
        if (!sp.acc_external) { A.p.acc[0] += rew_h; A.p.acc[1] += rew_g; }
        if (done) {
            if (!sp.acc_external) { A.p.acc[2] = A.p.acc[0]; A.p.acc[3] = A.p.acc[1]; }
            A.i.last_len = A.i.ep_len; A.i.ep_count += 1;
        }
        if (faulted) A.i.fault = 1;
    }
    RR_SYNC();
    RR_STAMP(11);
}

} // namespace rr
