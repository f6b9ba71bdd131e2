// rr_kstep.hpp -- the step kernel template (k_step) and the list of built configurations.
//
// Shared by rr_kernels.hip (the C-ABI + every other kernel) and rr_kstep_inst.hip: the product build compiles the k_step
// instantiations -- 14 configurations x up to five variants, most of the library's compile time -- in parallel translation
// units (rr_kstep_inst.hip with -DRR_PART=k holds the explicit instantiations of its share, rr_kernels.hip declares them
// `extern template` under -DRR_SPLIT_BUILD).  No relocatable device code is involved: every kernel is self-contained, each
// object registers its own code object, the host side only needs the launch stubs at link time.  Tuning builds
// (tools/build_variant.sh, tools/kernel_resources.py) keep the single translation unit with implicit instantiation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rr_sim.hpp"

using namespace rr;

// ------------------------------------------------------------------------------------------------ device side
// (under the F32State policy `rec` holds fp32 values: they are widened on the way into LDS and rounded once on the way back)
template <class C> __device__ __forceinline__ void load_record(Arena<C> &A, const typename C::Store *rec, const int32_t *irec) {
    using R = typename C::Real;
    R *p = reinterpret_cast<R *>(&A.p);
    int32_t *q = reinterpret_cast<int32_t *>(&A.i);
    const int lane = threadIdx.x & (C::VW - 1);
    for (int k = lane; k < Arena<C>::P_REALS; k += C::VW) p[k] = (R)rec[k];
    for (int k = lane; k < Arena<C>::I_INTS; k += C::VW) q[k] = irec[k];
    RR_SYNC();
}
template <class C> __device__ __forceinline__ void store_record(const Arena<C> &A, typename C::Store *rec, int32_t *irec) {
    using R = typename C::Real;
    const R *p = reinterpret_cast<const R *>(&A.p);
    const int32_t *q = reinterpret_cast<const int32_t *>(&A.i);
    const int lane = threadIdx.x & (C::VW - 1);
    RR_SYNC();
    for (int k = lane; k < Arena<C>::P_REALS; k += C::VW) rec[k] = (typename C::Store)p[k];
    // the ints and the record's padding: the whole 64-B tail is written, so no line of the record is left partially
    // dirty (a partial line costs a read-for-merge in L2: 0.3 KB per env-step showed up in FETCH_SIZE)
    constexpr int TAIL = Arena<C>::I_STRIDE - Arena<C>::P_REALS * Arena<C>::WS;
    for (int k = lane; k < TAIL; k += C::VW) irec[k] = k < Arena<C>::I_INTS ? q[k] : 0;
}

#ifndef RR_MIN_WAVES_PER_SIMD
#define RR_MIN_WAVES_PER_SIMD 4 // upper bound of the occupancy asked from the register allocator (<=128 VGPRs)
#endif
#ifndef RR_WAVES_PER_BLOCK
#define RR_WAVES_PER_BLOCK 1 // arenas never cooperate across wavefronts, so a workgroup IS a wavefront (finer dispatch: +9 % measured)
#endif
constexpr int WAVES_PER_BLOCK = RR_WAVES_PER_BLOCK;
#ifdef RR_ARENAS_PER_WAVE // occupancy probe only: fewer arenas per wavefront (idle lanes) so that LDS admits a third wave per SIMD
template <class C> constexpr int arenas_per_block() { return (C::VW == 8 ? RR_ARENAS_PER_WAVE : 64 / C::VW) * WAVES_PER_BLOCK; }
#else
template <class C> constexpr int arenas_per_block() { return 64 * WAVES_PER_BLOCK / C::VW; }
#endif
// Waves per SIMD the 160 KiB of LDS admit for this configuration.  Asking the register allocator for more than that
// (launch bounds) only buys spills: with 17 KB of LDS per wavefront G/VW=8 and T/VW=2 top out at 2 waves/SIMD, and
// capping them at 128 VGPRs put ~30 scratch round trips into every sub-step (measured: 487 VMEM instructions per
// wave-step instead of ~90, and a 0.25 ms latency floor per launch).
template <class C> constexpr int lds_waves_per_simd() {
#ifdef RR_FORCE_WAVES // occupancy experiments only (tools/kernel_resources.py ... -DRR_FORCE_WAVES=3)
    return RR_FORCE_WAVES;
#endif
    constexpr int per_cu = (160 * 1024) / (int)(sizeof(Arena<C>) * arenas_per_block<C>()) * WAVES_PER_BLOCK;
    return per_cu / 4 < 1 ? 1 : (per_cu / 4 > RR_MIN_WAVES_PER_SIMD ? RR_MIN_WAVES_PER_SIMD : per_cu / 4);
}


// MULTI = false: rr_step, one step per launch (nsteps, repeat unused); true: rr_rollout's loop over nsteps.  Separate
// instantiations: the loop around step_arena costs the single-step kernel 12 % (measured) through register allocation alone.
// BUDGET = true: the budgeted step (rr_sim.hpp: ParkCtx) -- a separate instantiation, so the default kernel carries none of it.
template <class C, typename O, bool MULTI, bool BUDGET = false>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, lds_waves_per_simd<C>()) void k_step(SimParams<typename C::Real> sp, typename C::Store *recs,
                                                              int32_t *irecs, int n, const int32_t *actions,
                                                              const float *thrust, int na, O *obs, O *reward,
                                                              uint8_t *done, O *obs_g, O *reward_g, int32_t *status,
                                                              const uint32_t *order, uint32_t *cost, int nsteps, int repeat,
                                                              uint32_t *snap, int32_t *isnap, uint32_t *park = nullptr,
                                                              uint32_t budget = 0) {
    static_assert(!(MULTI && BUDGET), "rr_rollout keeps the record in LDS across steps: no barrier to budget");
#ifdef RR_FAKE_LDS_ARENAS // resource experiments only (never run): what the register allocator does when LDS stops capping the occupancy
    __shared__ Arena<C> lds[RR_FAKE_LDS_ARENAS];
#else
    __shared__ Arena<C> lds[arenas_per_block<C>()];
#endif
    const int wave = threadIdx.x / C::VW; // virtual wave = arena slot in this workgroup
#ifdef RR_ARENAS_PER_WAVE
    if (wave >= arenas_per_block<C>()) return;
#endif
    // slowest-first dispatch: workgroup b steps the group of arenas that was the b-th slowest in the previous step
    const unsigned long long t_begin = (cost || BUDGET) ? __builtin_amdgcn_s_memtime() : 0ull;
    const int group = order ? (int)order[blockIdx.x] : (int)blockIdx.x;
    const int arena = group * arenas_per_block<C>() + wave;
    if (arena >= n) return; // uniform per virtual wave; no workgroup barrier is ever used
    Arena<C> &A = lds[wave];
    typename C::Store *rec = recs + (size_t)arena * Arena<C>::P_STRIDE;
    int32_t *irec = irecs + (size_t)arena * Arena<C>::I_STRIDE;
    RR_T0();
#if defined(RR_PROFILE_PHASES)
    const unsigned long long rr_wave_t0_ = __builtin_amdgcn_s_memrealtime(); // 100 MHz, one base for the whole chip
#endif
    load_record(A, rec, irec);
    derive(A, sp);
    RR_STAMP(12);
    if constexpr (!MULTI) {
        StepOut<O> o = { obs, obs_g, reward, reward_g, done, status, sp.memo ? snap : nullptr, isnap, arena,
                         (int)Arena<C>::SNAP_WORDS, (int)Arena<C>::ISNAP_WORDS };
        if constexpr (BUDGET) {
            ParkCtx pk;
            pk.buf = park + (size_t)arena * Arena<C>::PARK_WORDS; pk.budget = budget; pk.t_begin = t_begin;
            step_arena<C, O, true>(A, sp, sp.arena_offset + (uint64_t)arena, actions ? actions + (size_t)arena * na : nullptr,
                                   thrust ? thrust + (size_t)arena * 2 * na : nullptr, na, o, pk);
        } else {
            step_arena<C, O>(A, sp, sp.arena_offset + (uint64_t)arena, actions ? actions + (size_t)arena * na : nullptr,
                             thrust ? thrust + (size_t)arena * 2 * na : nullptr, na, o);
        }
    } else {
        // nsteps consecutive GameEnv.step calls on the record held in LDS (rr_rollout): step s reads its actions at
        // [s][arena] (or the same ones again when `repeat`) and writes its outputs at [s][arena]
#pragma unroll 1
        for (int s = 0; s < nsteps; s++) {
            const size_t so = (size_t)s * (size_t)n;
            StepOut<O> o = { obs + so * 11, obs_g ? obs_g + so * 11 : nullptr, reward + so, reward_g ? reward_g + so : nullptr, done + so,
                             status ? status + so : nullptr, sp.memo ? snap : nullptr, isnap, arena,
                             (int)Arena<C>::SNAP_WORDS, (int)Arena<C>::ISNAP_WORDS };
            const size_t ao = repeat ? 0 : so;
            step_arena<C, O>(A, sp, sp.arena_offset + (uint64_t)arena, actions ? actions + (ao + (size_t)arena) * na : nullptr,
                             thrust ? thrust + (ao + (size_t)arena) * 2 * na : nullptr, na, o);
        }
    }
    RR_TR();
    {   // the record addresses again, from an arena index the optimiser cannot tie to the first one: otherwise the two
        // 64-bit pointers stay live across the whole step (4 VGPRs of a kernel that sits at the 256-VGPR limit)
        int arena_again = arena;
        asm volatile("" : "+v"(arena_again));
        store_record(A, recs + (size_t)arena_again * Arena<C>::P_STRIDE, irecs + (size_t)arena_again * Arena<C>::I_STRIDE);
    }
    RR_STAMP(13);
    if (cost && threadIdx.x == 0) { // what this group cost, in shader clocks / 256 (saturating): next step's dispatch key
        const unsigned long long dt = ((__builtin_amdgcn_s_memtime() - t_begin) >> 8) / (unsigned)(MULTI ? nsteps : 1); // per step
        cost[group] = dt > 0xFFFFull ? 0xFFFFu : (uint32_t)dt;
    }
#if defined(RR_PROFILE_PHASES)
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 65536) {
        g_rr_wave_t[2 * blockIdx.x] = rr_wave_t0_;
        g_rr_wave_t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// Built configurations: (kind = shape + 3*dtype, entity counts, precision policy, VW) -- dtype 0: fp64, 1: fp32, 2: fp32 state with fp64
// arithmetic (rr_sim.hpp: F32State; default lane widths only, no budgeted / multi-step variant); shapes: 0 = T (1+0 robots, 1+0 balls), 1 = G (2+2, 4+4),
// 2 = D (1+1 robots, 1+1 balls: the two-team duel).  The first VW listed for a kind is the default; the environment variable
// RR_VW selects another built width (kernel tuning / A-B runs).
#if defined(RR_CUSTOM_SHAPE)
// A library for ONE shape outside the built list (roborugby_amd.build.build_shape_library: the reference's entity counts are free
// integers, RR_Constants.py:30-34): -DRR_CUSTOM_SHAPE -DRR_NRH= -DRR_NRG= -DRR_NBP= -DRR_NBN= -DRR_CVW=<lanes per arena> (+ RR_CFG_SUBSET:
// one translation unit, implicit instantiation).  Precisions: fp64 and fp32 state / fp64 arithmetic.
#define RR_FOR_EACH_CFG(X) X(0, RR_NRH, RR_NRG, RR_NBP, RR_NBN, double, RR_CVW) X(6, RR_NRH, RR_NRG, RR_NBP, RR_NBN, F32State, RR_CVW)
#elif defined(RR_CFG_SUBSET) && RR_CFG_SUBSET == 2 // occupancy probe: T at 4 lanes per arena (its LDS admits 4 waves per SIMD)
#define RR_FOR_EACH_CFG(X) X(0, 1, 0, 1, 0, double, 4) X(1, 2, 2, 4, 4, double, 8)
#elif defined(RR_CFG_SUBSET) && RR_CFG_SUBSET == 3 // tuning builds of the duel shape
#define RR_FOR_EACH_CFG(X) X(2, 1, 1, 1, 1, double, 4) X(1, 2, 2, 4, 4, double, 8)
#elif defined(RR_CFG_SUBSET) // tuning builds only (tools/build_variant.sh): the two default configurations, quick to compile
#define RR_FOR_EACH_CFG(X) X(0, 1, 0, 1, 0, double, 2) X(1, 2, 2, 4, 4, double, 8)
#else
#define RR_FOR_EACH_CFG(X)                                                                             \
    X(0, 1, 0, 1, 0, double, 2) X(0, 1, 0, 1, 0, double, 4) X(0, 1, 0, 1, 0, double, 8) X(0, 1, 0, 1, 0, double, 64) \
    X(1, 2, 2, 4, 4, double, 8) X(1, 2, 2, 4, 4, double, 16) X(1, 2, 2, 4, 4, double, 32) X(1, 2, 2, 4, 4, double, 64) \
    X(2, 1, 1, 1, 1, double, 4) X(2, 1, 1, 1, 1, double, 8) X(2, 1, 1, 1, 1, double, 64)                              \
    X(3, 1, 0, 1, 0, float, 2) X(3, 1, 0, 1, 0, float, 4) X(3, 1, 0, 1, 0, float, 64)                                 \
    X(4, 2, 2, 4, 4, float, 8) X(4, 2, 2, 4, 4, float, 16) X(4, 2, 2, 4, 4, float, 64)                                \
    X(5, 1, 1, 1, 1, float, 4)                                                                                        \
    X(6, 1, 0, 1, 0, F32State, 2) X(7, 2, 2, 4, 4, F32State, 8) X(8, 1, 1, 1, 1, F32State, 4)
#endif
constexpr int RR_NUM_SHAPES = 3;
// default lanes per arena of a shape (the widths rr_rollout's multi-step variant is built for)
#if defined(RR_CUSTOM_SHAPE)
template <class C> constexpr int default_vw() { return RR_CVW; }
#else
template <class C> constexpr int default_vw() { return C::NR == 1 ? 2 : C::NR == 2 ? 4 : 8; }
#endif

// ---- split build: the same list once more, with the translation unit (part) each configuration's k_step instantiations are
// compiled in -- G kernels are the slow ones to compile, so they are spread first.  X(part, NRH, NRG, NBP, NBN, Real, VW, DEF)
// with DEF = 1 for the default lane width of its kind (the only ones rr_rollout's MULTI variant is built for).
#define RR_KSTEP_PARTS 7
#define RR_FOR_EACH_CFG_F64_PARTS(X)                                                                                   \
    X(5, 1, 0, 1, 0, double, 2, 1) X(5, 1, 0, 1, 0, double, 4, 0) X(6, 1, 0, 1, 0, double, 8, 0) X(6, 1, 0, 1, 0, double, 64, 0) \
    X(0, 2, 2, 4, 4, double, 8, 1) X(1, 2, 2, 4, 4, double, 16, 0) X(2, 2, 2, 4, 4, double, 32, 0) X(3, 2, 2, 4, 4, double, 64, 0) \
    X(1, 1, 1, 1, 1, double, 4, 1) X(2, 1, 1, 1, 1, double, 8, 0) X(3, 1, 1, 1, 1, double, 64, 0)
#define RR_FOR_EACH_CFG_F32_PARTS(X)                                                                                   \
    X(6, 1, 0, 1, 0, float, 2, 1) X(6, 1, 0, 1, 0, float, 4, 0) X(6, 1, 0, 1, 0, float, 64, 0)                         \
    X(4, 2, 2, 4, 4, float, 8, 1) X(4, 2, 2, 4, 4, float, 16, 0) X(5, 2, 2, 4, 4, float, 64, 0)                        \
    X(0, 1, 1, 1, 1, float, 4, 1)
// fp32 state / fp64 arithmetic: the plain single-step kernel only, float and double outputs
#define RR_FOR_EACH_CFG_F32S_PARTS(X) X(5, 1, 0, 1, 0, F32State, 2, 0) X(2, 2, 2, 4, 4, F32State, 8, 0) X(6, 1, 1, 1, 1, F32State, 4, 0)
#define RR_KSTEP_PLAIN(PREFIX, a, b, c, d, R_, vw_, O_) \
    PREFIX template __global__ void k_step<Cfg<a, b, c, d, R_, vw_>, O_, false, false> RR_KSTEP_SIG(RR_KSTEP_CFG(a, b, c, d, R_, vw_), O_);

#define RR_KSTEP_SIG(C_, O_)                                                                                                       \
    (SimParams<typename C_::Real>, typename C_::Store *, int32_t *, int, const int32_t *, const float *, int, O_ *, O_ *, uint8_t *, \
     O_ *, O_ *, int32_t *, const uint32_t *, uint32_t *, int, int, uint32_t *, int32_t *, uint32_t *, uint32_t)
// every variant step_impl can launch for one configuration and one output type: plain, budgeted, and (default widths, float
// outputs) rr_rollout's multi-step loop
#define RR_KSTEP_VARIANTS(PREFIX, a, b, c, d, R_, vw_, O_, MULTI_)                                                    \
    PREFIX template __global__ void k_step<Cfg<a, b, c, d, R_, vw_>, O_, false, false> RR_KSTEP_SIG(RR_KSTEP_CFG(a, b, c, d, R_, vw_), O_); \
    PREFIX template __global__ void k_step<Cfg<a, b, c, d, R_, vw_>, O_, false, true> RR_KSTEP_SIG(RR_KSTEP_CFG(a, b, c, d, R_, vw_), O_);  \
    RR_KSTEP_MULTI_##MULTI_(PREFIX, a, b, c, d, R_, vw_, O_)
#define RR_KSTEP_CFG(a, b, c, d, R_, vw_) Cfg<a, b, c, d, R_, vw_>
#define RR_KSTEP_MULTI_0(PREFIX, a, b, c, d, R_, vw_, O_)
#define RR_KSTEP_MULTI_1(PREFIX, a, b, c, d, R_, vw_, O_) \
    PREFIX template __global__ void k_step<Cfg<a, b, c, d, R_, vw_>, O_, true, false> RR_KSTEP_SIG(RR_KSTEP_CFG(a, b, c, d, R_, vw_), O_);

#if defined(RR_SPLIT_BUILD) && !defined(RR_CFG_SUBSET)
// rr_kernels.hip: the instantiations live in rr_kstep_inst.hip's objects
#define X(part, a, b, c, d, R_, vw_, def_) RR_KSTEP_VARIANTS(extern, a, b, c, d, R_, vw_, float, def_) RR_KSTEP_VARIANTS(extern, a, b, c, d, R_, vw_, double, 0)
RR_FOR_EACH_CFG_F64_PARTS(X)
#undef X
#define X(part, a, b, c, d, R_, vw_, def_) RR_KSTEP_VARIANTS(extern, a, b, c, d, R_, vw_, float, def_)
RR_FOR_EACH_CFG_F32_PARTS(X)
#undef X
#define X(part, a, b, c, d, R_, vw_, def_) RR_KSTEP_PLAIN(extern, a, b, c, d, R_, vw_, float) RR_KSTEP_PLAIN(extern, a, b, c, d, R_, vw_, double)
RR_FOR_EACH_CFG_F32S_PARTS(X)
#undef X
#endif
