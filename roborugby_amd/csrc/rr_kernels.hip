// rr_kernels.hip -- gfx950 kernels + the C-ABI (include/roborugby_amd.h) of the batched RoboRugby simulator.
//
// Launch geometry: 64-thread workgroups = 1 wavefront; a wavefront is cut into 64/VW virtual waves of VW
// lanes and every virtual wave owns one arena (VW = 64: one wavefront per arena; VW = 16: four arenas per
// wavefront, ...).  Each arena has a private LDS slice (Arena<C>) and never talks to another arena, so there
// is no workgroup barrier anywhere and no inter-workgroup traffic -- any blockIdx -> XCD placement is equally
// good (arenas share nothing, there is no L2 reuse to protect).  65,536 arenas = 2,048..65,536 workgroups.
//
// HBM layout: one record per arena -- the persistent reals (Arena::P) with the int32 bookkeeping (Arena::I) right behind
// them, padded to a 64-B multiple (G/fp64: 960 B, T/fp64: 256 B).  Inside a record the fields are entity-minor (SoA over
// robots / balls), and an arena's lanes load/store it with lane-strided accesses: lane k touches word k, so every
// wave-level access covers one contiguous run per arena that starts on a 64-B boundary.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <cmath>
#include <new>
#include <string>
#include <type_traits>
#include <cstdlib>

#include "../../include/roborugby_amd.h"
#include "rr_sim.hpp"
#include "rr_extras.hpp"
#include "rr_kstep.hpp"

using namespace rr;

// (load_record / store_record / k_step and the list of built configurations: rr_kstep.hpp, shared with rr_kstep_inst.hip)

// Slowest-first order for the next launch (longest-processing-time-first list scheduling).  A launch ends when its
// slowest wavefront does and wavefronts differ 3x and more in duration (contact-rich arenas); dispatched in index order a
// slow one may start in the last round and the chip idles behind it -- measured: 345 us of work per launch stretched
// to ~500 us.  Contact situations persist over steps, so last step's duration is a good key.  One workgroup, counting
// sort of the per-group costs into descending order (ties in arbitrary order: results never depend on the order).
#ifndef RR_ORDER_SHIFT
#define RR_ORDER_SHIFT 2 // cost unit is 256 clocks: 1,024 clocks per bin
#endif
constexpr int ORDER_BINS = 1024, ORDER_THREADS = 1024;
__global__ __launch_bounds__(ORDER_THREADS) void k_order(const uint32_t *cost, uint32_t *order, int ngroups) {
    __shared__ uint32_t hist[ORDER_BINS];
    for (int i = threadIdx.x; i < ORDER_BINS; i += ORDER_THREADS) hist[i] = 0;
    __syncthreads();
    for (int g = threadIdx.x; g < ngroups; g += ORDER_THREADS) {
        uint32_t c = cost[g] >> RR_ORDER_SHIFT; // 1,024 clocks per bin: 0 .. ~1M clocks (0.4 ms) resolved, slower groups share the first bin
        c = c > ORDER_BINS - 1 ? ORDER_BINS - 1 : c;
        atomicAdd(&hist[ORDER_BINS - 1 - c], 1u);
    }
    __syncthreads();
    // exclusive prefix sum over the bins (one thread per bin, Hillis-Steele in LDS)
    __shared__ uint32_t scan[ORDER_BINS];
    uint32_t v = hist[threadIdx.x];
    scan[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < ORDER_BINS; off <<= 1) {
        uint32_t add = threadIdx.x >= (unsigned)off ? scan[threadIdx.x - off] : 0u;
        __syncthreads();
        scan[threadIdx.x] += add;
        __syncthreads();
    }
    hist[threadIdx.x] = scan[threadIdx.x] - v; // start offset of the bin
    __syncthreads();
    for (int g = threadIdx.x; g < ngroups; g += ORDER_THREADS) {
        uint32_t c = cost[g] >> RR_ORDER_SHIFT;
        c = c > ORDER_BINS - 1 ? ORDER_BINS - 1 : c;
        order[atomicAdd(&hist[ORDER_BINS - 1 - c], 1u)] = (uint32_t)g;
    }
}
__global__ void k_iota(uint32_t *order, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = (uint32_t)i;
}

// env.reset() for masked arenas; also used (init = 1) to build the constructor's state
template <class C, typename O>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_reset(SimParams<typename C::Real> sp, typename C::Store *recs,
                                                               int32_t *irecs, int n, const uint8_t *mask, int init,
                                                               O *obs, O *obs_g) {
    using R = typename C::Real;
    __shared__ Arena<C> lds[arenas_per_block<C>()];
    const int wave = threadIdx.x / C::VW; // virtual wave = arena slot in this workgroup
    const int arena = blockIdx.x * arenas_per_block<C>() + wave;
    if (arena >= n || wave >= arenas_per_block<C>()) return;
    Arena<C> &A = lds[wave];
    typename C::Store *rec = recs + (size_t)arena * Arena<C>::P_STRIDE;
    int32_t *irec = irecs + (size_t)arena * Arena<C>::I_STRIDE;
    int st = 0;
    if (init) { // Robot(team, (0,0)) / Ball(color, (0,0)) as built by GameEnv.__init__ (RR_EnvBase.py:85-109)
        const int lane = threadIdx.x & (C::VW - 1);
        R *p = reinterpret_cast<R *>(&A.p);
        int32_t *q = reinterpret_cast<int32_t *>(&A.i);
        for (int k = lane; k < Arena<C>::P_REALS; k += C::VW) p[k] = (R)0;
        for (int k = lane; k < Arena<C>::I_INTS; k += C::VW) q[k] = 0;
        RR_SYNC();
        if (lane < C::NR) robot_set_clean_lane(A, sp, lane, (R)0, (R)0, lane < C::NRH ? (R)90 : (R)-90);
        if (lane < C::NB) ball_set_clean_lane(A, lane, (R)0, (R)0, (R)0, (R)0);
        if (lane == 0) A.i.episode = -1;
        RR_SYNC();
    } else {
        load_record(A, rec, irec);
    }
    const bool doit = init || !mask || mask[arena];
    if (doit) reset_arena(A, sp, sp.arena_offset + (uint64_t)arena, (uint64_t)(uint32_t)(A.i.episode + 1), st);
    else derive(A, sp);
    if (obs && doit) observe<C, O>(A, sp, 1, -1, -1, obs + (size_t)arena * 11, st);
    if (obs_g && doit) {
        if (!observe<C, O>(A, sp, -1, -1, -1, obs_g + (size_t)arena * 11, st)) {
            for (int k = threadIdx.x & (C::VW - 1); k < 11; k += C::VW) obs_g[(size_t)arena * 11 + k] = (O)NAN;
        }
    }
    store_record(A, rec, irec);
}

template <class C, typename O>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_observe(SimParams<typename C::Real> sp, const typename C::Store *recs,
                                                                 const int32_t *irecs, int n, int team, int ridx, int bidx,
                                                                 O *obs) {
    __shared__ Arena<C> lds[arenas_per_block<C>()];
    const int wave = threadIdx.x / C::VW; // virtual wave = arena slot in this workgroup
    const int arena = blockIdx.x * arenas_per_block<C>() + wave;
    if (arena >= n || wave >= arenas_per_block<C>()) return;
    Arena<C> &A = lds[wave];
    load_record(A, recs + (size_t)arena * Arena<C>::P_STRIDE, irecs + (size_t)arena * Arena<C>::I_STRIDE);
    derive(A, sp);
    int st = 0;
    if (!observe<C, O>(A, sp, team, ridx, bidx, obs + (size_t)arena * 11, st)) {
        for (int k = threadIdx.x & (C::VW - 1); k < 11; k += C::VW) obs[(size_t)arena * 11 + k] = (O)NAN;
    }
}

// canonical fp64 state <-> record (not hot: one thread per arena)
template <class C>
__global__ void k_set_state(typename C::Store *recs, int32_t *irecs, int n, const double *robots, const int32_t *ri,
                            const double *balls, const int32_t *step) {
    using R = typename C::Real;
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    typename C::Store *rec = recs + (size_t)a * Arena<C>::P_STRIDE;
    int32_t *irec = irecs + (size_t)a * Arena<C>::I_STRIDE;
    constexpr int NR = C::NR, NB = C::NB;
    for (int r = 0; r < NR; r++) {
        for (int f = 0; f < 10; f++) rec[f * NR + r] = (typename C::Store)robots[((size_t)a * NR + r) * 10 + f];
        for (int f = 0; f < 3; f++) irec[f * NR + r] = ri[((size_t)a * NR + r) * 3 + f];
    }
    for (int b = 0; b < NB; b++)
        for (int f = 0; f < 8; f++) rec[10 * NR + f * NB + b] = (typename C::Store)balls[((size_t)a * NB + b) * 8 + f];
    irec[3 * NR + 0] = step[a];
    irec[3 * NR + 5] = 0; // fault flag
    irec[3 * NR + 6] = 0; // no island carried over from the previous step (Arena::I::fzp)
#if RR_CARRY
    // the scratch rect where a sub-step leaves it, on the last ball (rr_set_scratch_rect overrides: a dumped reference state has its own)
    rec[10 * NR + 8 * NB + 4] = rec[10 * NR + 0 * NB + NB - 1]; rec[10 * NR + 8 * NB + 5] = rec[10 * NR + 1 * NB + NB - 1];
#endif
}
// centre of the reference's scratch rect (Arena::P::ic, parity build only): xy [N,2]
template <class C>
__global__ void k_scratch_rect(typename C::Store *recs, int n, double *xy, int set) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
#if RR_CARRY
    typename C::Store *rec = recs + (size_t)a * Arena<C>::P_STRIDE + 10 * C::NR + 8 * C::NB + 4;
    for (int k = 0; k < 2; k++) { if (set) rec[k] = (typename C::Store)xy[(size_t)a * 2 + k]; else xy[(size_t)a * 2 + k] = (double)rec[k]; }
#else
    (void)recs; (void)xy; (void)set;
#endif
}
template <class C>
__global__ void k_get_state(const typename C::Store *recs, const int32_t *irecs, int n, double *robots, int32_t *ri,
                            double *balls, int32_t *step) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const typename C::Store *rec = recs + (size_t)a * Arena<C>::P_STRIDE;
    const int32_t *irec = irecs + (size_t)a * Arena<C>::I_STRIDE;
    constexpr int NR = C::NR, NB = C::NB;
    for (int r = 0; r < NR; r++) {
        for (int f = 0; f < 10; f++) robots[((size_t)a * NR + r) * 10 + f] = (double)rec[f * NR + r];
        for (int f = 0; f < 3; f++) ri[((size_t)a * NR + r) * 3 + f] = irec[f * NR + r];
    }
    for (int b = 0; b < NB; b++)
        for (int f = 0; f < 8; f++) balls[((size_t)a * NB + b) * 8 + f] = (double)rec[10 * NR + f * NB + b];
    step[a] = irec[3 * NR + 0];
}
// rr_set_poses, and env.reset(bln_randomize_pos=False) = _set_starting_positions (RR_EnvBase.py:131-153,202-216) through
// rr_reset_to_poses: masked arenas only, first observations out
template <class C, typename O>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_set_poses(SimParams<typename C::Real> sp, typename C::Store *recs,
                                                                   int32_t *irecs, int n, const double *rxyr,
                                                                   const double *bxyv, const uint8_t *mask, O *obs, O *obs_g) {
    using R = typename C::Real;
    __shared__ Arena<C> lds[arenas_per_block<C>()];
    const int wave = threadIdx.x / C::VW, lane = threadIdx.x & (C::VW - 1);
    const int arena = blockIdx.x * arenas_per_block<C>() + wave;
    if (arena >= n || wave >= arenas_per_block<C>()) return;
    if (mask && !mask[arena]) return; // uniform per virtual wave
    Arena<C> &A = lds[wave];
    typename C::Store *rec = recs + (size_t)arena * Arena<C>::P_STRIDE;
    int32_t *irec = irecs + (size_t)arena * Arena<C>::I_STRIDE;
    load_record(A, rec, irec);
    if (lane < C::NR) {
        const double *q = rxyr + ((size_t)arena * C::NR + lane) * 3;
        robot_set_clean_lane(A, sp, lane, (R)q[0], (R)q[1], (R)q[2]);
    }
    if (lane < C::NB) {
        const double *q = bxyv + ((size_t)arena * C::NB + lane) * 4;
        ball_set_clean_lane(A, lane, (R)q[0], (R)q[1], (R)q[2], (R)q[3]);
    }
    if (lane == 0) { A.i.step = 0; A.i.ep_len = 0; A.i.fault = 0; A.i.fzp = 0; A.p.acc[0] = (R)0; A.p.acc[1] = (R)0; }
    RR_SYNC();
    if (obs || obs_g) {
        int st = 0;
        derive(A, sp);
        if (obs) observe<C, O>(A, sp, 1, -1, -1, obs + (size_t)arena * 11, st);
        if (obs_g && !observe<C, O>(A, sp, -1, -1, -1, obs_g + (size_t)arena * 11, st))
            for (int k = lane; k < 11; k += C::VW) obs_g[(size_t)arena * 11 + k] = (O)NAN;
    }
    store_record(A, rec, irec);
}
// episode bookkeeping <-> caller (checkpoint / resume): ints [N,5] = episode, ep_len, ep_count, last_len, fault; acc [N,4]
template <class C>
__global__ void k_episode_state(typename C::Store *recs, int32_t *irecs, int n, int32_t *ints, double *acc, int set) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    typename C::Store *rec = recs + (size_t)a * Arena<C>::P_STRIDE;
    int32_t *irec = irecs + (size_t)a * Arena<C>::I_STRIDE;
    constexpr int ACC = 10 * C::NR + 8 * C::NB, I0 = 3 * C::NR + 1; // after mc/thl/thr and step
    for (int k = 0; k < 5; k++) { if (set) irec[I0 + k] = ints[(size_t)a * 5 + k]; else ints[(size_t)a * 5 + k] = irec[I0 + k]; }
    for (int k = 0; k < 4; k++) { if (set) rec[ACC + k] = (typename C::Store)acc[(size_t)a * 4 + k]; else acc[(size_t)a * 4 + k] = (double)rec[ACC + k]; }
}
template <class C>
__global__ void k_episode_stats(const typename C::Store *recs, const int32_t *irecs, int n, float *lr, float *lrg,
                                int32_t *ll, int32_t *cnt) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const typename C::Store *rec = recs + (size_t)a * Arena<C>::P_STRIDE;
    const int32_t *irec = irecs + (size_t)a * Arena<C>::I_STRIDE;
    constexpr int ACC = 10 * C::NR + 8 * C::NB;
    if (lr) lr[a] = (float)rec[ACC + 2];
    if (lrg) lrg[a] = (float)rec[ACC + 3];
    if (ll) ll[a] = irec[3 * C::NR + 4];
    if (cnt) cnt[a] = irec[3 * C::NR + 3];
}

// ---- other mixins (rr_extras.hpp): thread-per-arena side kernels, straight from the HBM records
template <class C>
__global__ void k_extras_begin(const typename C::Store *recs, int n, typename C::Real *xs, const uint8_t *mask = nullptr) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n || (mask && !mask[a])) return;
    Rec<C> q = { recs + (size_t)a * Arena<C>::P_STRIDE };
    // budgeted step: an arena parked mid-step (bit 31 of the record's fzp word, step_arena) began its step in an earlier call -- the copies
    // taken then are the ones its on_step_end will need (a masked reset re-seeds them: rr_reset clears the mark first)
    const int32_t *irec = reinterpret_cast<const int32_t *>(q.p + Arena<C>::P_REALS);
    if (!mask && irec[3 * C::NR + 6] < 0) return;
    extras_begin<C>(q, xs + (size_t)a * xs_stride<C>());
}
// After k_step, when a non-default keeper program and / or prior-step tracking is on: the program's rewards replace the
// fused SimpleDuel3 ones AND feed the episode-return accumulators (k_step leaves them alone then: sp.acc_external), and an
// arena that k_step re-placed (auto-reset) gets its on_step_begin copies re-seeded from the new poses, like rr_reset does.
template <class C, typename O>
__global__ void k_extras_end(SimParams<typename C::Real> sp, typename C::Store *recs, int n, typename C::Real *xs,
                             Program pg, int rewrite, O *reward, O *reward_g, int32_t *status, const uint8_t *done) {
    using R = typename C::Real;
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const int32_t st = status[a];
    typename C::Store *rec = recs + (size_t)a * Arena<C>::P_STRIDE;
    Rec<C> q = { rec };
    if (st & ST_NOT_READY) return; // budgeted step: this arena's step is still in progress -- its on_step_end comes with the call that completes it
    if (st & ST_WAS_RESET) { extras_begin<C>(q, xs + (size_t)a * xs_stride<C>()); return; } // new episode: no prior step yet
    if (!rewrite || (st & ST_STEP_AFTER_DONE)) return; // nothing was stepped: the reward stays 0
    R rh, rg;
    extras_end<C, O>(q, sp, xs + (size_t)a * xs_stride<C>(), pg, (uint32_t)st >> 16, reward + a, reward_g ? reward_g + a : nullptr,
                     status + a, rh, rg);
    constexpr int ACC = 10 * C::NR + 8 * C::NB; // running return happy/grumpy, last finished return happy/grumpy
    rec[ACC + 0] += rh; rec[ACC + 1] += rg;
    if (done[a]) { rec[ACC + 2] = rec[ACC + 0]; rec[ACC + 3] = rec[ACC + 1]; }
}
// opt-in goal scoring (rr_extras.hpp: goal_step): thread per arena, after k_step (and k_extras_end)
template <class C, typename O>
__global__ void k_goal(SimParams<typename C::Real> sp, typename C::Store *recs, int32_t *irecs, int n, int32_t *gs, int base_destruction,
                       O *reward, O *reward_g, uint8_t *done, int32_t *status) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    goal_step<C, O>(recs + (size_t)a * Arena<C>::P_STRIDE, irecs + (size_t)a * Arena<C>::I_STRIDE, sp, gs + (size_t)a * gs_stride<C>(),
                    base_destruction != 0, reward + a, reward_g ? reward_g + a : nullptr, done + a, status + a);
}
template <class C>
__global__ void k_goal_clear(int n, int32_t *gs, const uint8_t *mask) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n || (mask && !mask[a])) return;
    goal_state_clear<C>(gs + (size_t)a * gs_stride<C>());
}
template <class C>
__global__ void k_goal_scores(int n, const int32_t *gs, int32_t *scores) { // Goal.get_score (RR_Goal.py:87-88) of the happy / grumpy goal
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const int32_t *g = gs + (size_t)a * gs_stride<C>() + 1 + 2 * C::NB;
    for (int k = 0; k < 2; k++) scores[2 * a + k] = 500 * (popcount8(g[k]) - popcount8(g[2 + k]));
}
template <class C, typename O>
__global__ void k_observe_kind(SimParams<typename C::Real> sp, const typename C::Store *recs, int n, int kind, int team, int ridx,
                               int bidx, O *obs, int dim, const typename C::Real *xs) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    Rec<C> q = { recs + (size_t)a * Arena<C>::P_STRIDE };
    O tmp[6 * C::NR + 4 * C::NB > 11 ? 6 * C::NR + 4 * C::NB : 11];
    const int m = observe_kind<C, O>(q, sp, kind, team, ridx, bidx, tmp, xs ? xs + (size_t)a * xs_stride<C>() : nullptr);
    for (int k = 0; k < dim; k++) obs[(size_t)a * dim + k] = k < m ? tmp[k] : (O)NAN;
}

// Scripted on-device policy of the contact-rich workload ("turn toward ball_angle, else forward, 10 %
// random"): robot 0 of every arena chases its ball from the arena's own observation row, the other robots act at random.
// One thread per arena, counter-based RNG keyed by (seed, global arena id, step) -- the policy is a function of the arena's
// observation and of the step index the caller passes, nothing else (so it is the same whatever the batch is doing around it).
__global__ void k_policy_chase(const float *obs, int n, int na, uint32_t noise_u32, uint64_t seed, uint64_t arena_offset,
                               const int32_t *step_of, uint32_t step, int32_t *actions) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const float d = fmodf(obs[(size_t)a * 11 + 1] - obs[(size_t)a * 11 + 0] + 540.0f, 360.0f) - 180.0f;
    int act = fabsf(d) < 8.0f ? 0 : (d > 0.0f ? 2 : 3); // FORWARD / LEFT / RIGHT (RR_EnvBase.py:583-591)
    const uint64_t gid = arena_offset + (uint64_t)a;
    const uint32_t st = step_of ? (uint32_t)step_of[a] : step;
    uint32_t c[4] = { (uint32_t)gid, (uint32_t)(gid >> 32), st, 0x70C1u };
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    if (c[0] < noise_u32) act = (int)(c[1] & 7u);
    actions[(size_t)a * na] = act;
    for (int k = 1; k < na; k++) actions[(size_t)a * na + k] = (int)((c[2 + (k & 1)] >> (4 * (k >> 1))) & 7u);
}

// HBM copy probe ("confirm on the box with a hipMemcpyDtoD / stream-triad probe"): ONE 16-byte element per
// thread, one workgroup per 4 KB, no loop -- the shape MI355X_MICROARCH.md quotes its ~6.3 TB/s for.  Measured on the MI355X
// (tools/probes/hbm_copy_probe.hip, profiles/r04/hbm_copy_probe.txt, 1 GiB, read + write bytes): this 6.15 TB/s; grid-stride loops
// with 1-8 elements in flight per lane 4.3-5.7 TB/s (the round-3 probe was one of those: 4.66); hipMemcpyDtoD 5.19.  bench.py times
// it with HIP events and reports it next to the 8 TB/s spec (roofline.peak_measured).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------ host side
struct rr_env {
    rr_config cfg;
    int kind; // shape + RR_NUM_SHAPES * dtype: 0 T64, 1 G64, 2 D64, 3 T32, 4 G32, 5 D32, 6-8 T / G / D with fp32 state and fp64 arithmetic
    int vw;   // lanes per arena
    void *recs;
    int32_t *irecs;      // the int part of the records: recs + P_REALS (same allocation, same stride)
    size_t rec_bytes, irec_bytes;
    uint32_t *snap;      // fixed-point snapshots (null when the shortcuts are off)
    int32_t *isnap;
    SimParams<double> spd;
    SimParams<float> spf;
    Program prog;        // reward keepers in execution order
    bool custom_prog;    // != SimpleDuel3's {Naughty, Chase, PushPos}
    bool track_prior;    // keep the on_step_begin snapshot up to date for AllCoords_WithPrior (rr_track_prior_step)
    int32_t *gs;         // goal bookkeeping of the opt-in goal-scoring mode (null: off)
    void *xs;            // on_step_begin snapshot for the side kernels (lazy)
    int32_t *status_buf; // internal status when the caller passes none but the side kernels need it (lazy)
    uint32_t *order;     // slowest-first dispatch order of the arena groups (null: index order)
    uint32_t *cost;      // last step's duration per group
    int ngroups;
    uint32_t *park;      // parked mid-step state of the budgeted step (null until a budget is first set)
    uint32_t budget;     // shader clocks; 0xFFFFFFFF = "no budget, but parked arenas may exist" (after the budget was switched off)
};

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename R> static void fill_params(SimParams<R> &sp, const rr_config &c) {
    const double W = c.arena_w, H = c.arena_h;
    sp.W = (R)W; sp.H = (R)H;
    const double mb = 200000.0 / std::pow(W * W + H * H, .5); // RR_Constants.py:44-46
    sp.mult_ball = (R)mb;
    sp.mult_robot = (R)(mb / 100);                              // RR_Constants.py:50
    sp.rob_cdist = (R)std::pow(10.0 * 10.0 + 20.0 * 20.0, .5);  // MyUtils.py:138 for the 20x40 robot rect
    const double hr = 7 * std::pow(2.0, .5) / 2;                // RR_TrashyPhysics.py:29
    sp.inner_h = (R)hr;
    sp.inner_cdist = (R)std::pow(hr * hr + hr * hr, .5);
    sp.game_len = c.game_len_steps; sp.game_mode = c.game_mode; sp.time_limit = c.time_limit; sp.auto_reset = c.auto_reset;
    sp.reset_on_fault = c.reset_on_fault;
    sp.acc_external = 0;
    { const char *nm = getenv("RR_NO_MEMO"); sp.memo = (nm && atoi(nm)) ? 0 : 1; } // fixed-point check of the sub-step loop (exact; the switch is for A/B runs)
    sp.seed = c.seed; sp.arena_offset = c.arena_offset;
}
template <typename R> static const SimParams<R> &params_of(const rr_env *e);
template <> const SimParams<double> &params_of<double>(const rr_env *e) { return e->spd; }
template <> const SimParams<float> &params_of<float>(const rr_env *e) { return e->spf; }

// calls f(Cfg<...>{}) for the configuration the handle was created with
template <class F> static int dispatch(const rr_env *e, F &&f) {
#define X(kind_, a, b, c, d, R_, vw_) \
    if (e->kind == kind_ && e->vw == vw_) return f(Cfg<a, b, c, d, R_, vw_>{});
    RR_FOR_EACH_CFG(X)
#undef X
    return fail(-1, "corrupt handle");
}
// The caller's current HIP device is left as found; launches go to the device the handle was created on.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int want) {
        if (hipGetDevice(&prev) == hipSuccess && prev != want) switched = (hipSetDevice(want) == hipSuccess);
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
template <class C> static dim3 arena_grid(int n) { return dim3((unsigned)((n + arenas_per_block<C>() - 1) / arenas_per_block<C>())); }
static inline dim3 wave_block() { return dim3(64 * WAVES_PER_BLOCK); }

extern "C" {

int rr_abi_version(void) { return RR_ABI_VERSION; }
int rr_exact_trig(void) { return RR_EXACT_TRIG; }
const char *rr_last_error(void) { return g_err.c_str(); }

int rr_create(const rr_config *cfg, rr_env **out) {
    if (!cfg || !out) return fail(-1, "rr_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(rr_config)) return fail(-1, "rr_create: rr_config.struct_size mismatch");
    if (cfg->num_envs <= 0) return fail(-1, "rr_create: num_envs must be positive");
    int shape;
#if defined(RR_CUSTOM_SHAPE)
    if (cfg->nr_happy == RR_NRH && cfg->nr_grumpy == RR_NRG && cfg->nb_pos == RR_NBP && cfg->nb_neg == RR_NBN) shape = 0;
    else return fail(-1, "rr_create: this library was built for one shape only (roborugby_amd.build.build_shape_library) and it is not this one");
    if (cfg->dtype == RR_DTYPE_F32) return fail(-1, "rr_create: a one-shape library is built for RR_DTYPE_F64 and RR_DTYPE_F32_STATE only");
#else
    if (cfg->nr_happy == 1 && cfg->nr_grumpy == 0 && cfg->nb_pos == 1 && cfg->nb_neg == 0) shape = 0;
    else if (cfg->nr_happy == 2 && cfg->nr_grumpy == 2 && cfg->nb_pos == 4 && cfg->nb_neg == 4) shape = 1;
    else if (cfg->nr_happy == 1 && cfg->nr_grumpy == 1 && cfg->nb_pos == 1 && cfg->nb_neg == 1) shape = 2;
    else return fail(-1, "rr_create: unsupported entity counts (built shapes: 1+0 robots/1+0 balls, 2+2 robots/4+4 balls, 1+1 robots/1+1 balls; "
                         "any other counts: a one-shape library, roborugby_amd.build.build_shape_library / BatchedRoboRugbyEnv does it on demand)");
#endif
    if (cfg->dtype != RR_DTYPE_F64 && cfg->dtype != RR_DTYPE_F32 && cfg->dtype != RR_DTYPE_F32_STATE) return fail(-1, "rr_create: bad dtype");
    if (cfg->dtype == RR_DTYPE_F32_STATE && cfg->step_budget_clocks) return fail(-1, "rr_create: no step budget with RR_DTYPE_F32_STATE");
    if (!(cfg->arena_w >= 300 && cfg->arena_h >= 300 && cfg->arena_w <= 8192 && cfg->arena_h <= 8192))
        return fail(-1, "rr_create: arena size out of range [300, 8192]");
    if (cfg->game_len_steps <= 0) return fail(-1, "rr_create: game_len_steps must be positive");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(-1, "rr_create: no such HIP device");
    DeviceGuard guard(cfg->device); // the caller's current device is left as found (like every other entry point)
    rr_env *e = new (std::nothrow) rr_env();
    if (!e) return fail(-3, "rr_create: out of host memory");
    e->cfg = *cfg;
    e->kind = shape + RR_NUM_SHAPES * cfg->dtype;
    e->vw = 0;
    e->prog.n = 3; e->prog.id[0] = KEEPER_NAUGHTY; e->prog.id[1] = KEEPER_CHASE; e->prog.id[2] = KEEPER_PUSHPOS;
    e->custom_prog = false; e->track_prior = false; e->xs = nullptr; e->status_buf = nullptr; e->gs = nullptr;
    e->park = nullptr; e->budget = 0;
    const char *want = getenv("RR_VW");
    const int want_vw = want ? atoi(want) : 0;
#define X(kind_, a, b, c, d, R_, vw_) \
    if (e->kind == kind_ && (e->vw == 0 || want_vw == vw_)) e->vw = vw_;
    RR_FOR_EACH_CFG(X)
#undef X
    fill_params(e->spd, *cfg);
    fill_params(e->spf, *cfg);
    size_t pstride = 0, preals = 0, snapw = 0, isnapw = 0;
    const size_t rsz = cfg->dtype == RR_DTYPE_F64 ? 8 : 4; // bytes per STORED real (the record)
    dispatch(e, [&](auto c) {
        using CC = decltype(c);
        pstride = Arena<CC>::P_STRIDE; preals = Arena<CC>::P_REALS; snapw = Arena<CC>::SNAP_WORDS; isnapw = Arena<CC>::ISNAP_WORDS;
        return 0;
    });
    e->rec_bytes = pstride * rsz;
    e->irec_bytes = 0; // inside the record
    e->snap = nullptr; e->isnap = nullptr;
    hipError_t he = hipMalloc(&e->recs, e->rec_bytes * (size_t)cfg->num_envs);
    if (he == hipSuccess && e->spd.memo) { // scratch of the exact shortcuts (RR_NO_MEMO=1: off)
        he = hipMalloc((void **)&e->snap, 4 * snapw * (size_t)cfg->num_envs);
        if (he == hipSuccess) he = hipMalloc((void **)&e->isnap, 4 * isnapw * (size_t)cfg->num_envs);
    }
    if (he != hipSuccess) {
        if (e->recs) (void)hipFree(e->recs);
        if (e->snap) (void)hipFree(e->snap);
        delete e;
        return fail(-3, std::string("rr_create: hipMalloc: ") + hipGetErrorString(he));
    }
    e->irecs = reinterpret_cast<int32_t *>(static_cast<char *>(e->recs) + preals * rsz);
    const int n = cfg->num_envs;
    // slowest-first dispatch pays once the groups outnumber the wavefronts resident at a time (a few thousand); one
    // workgroup sorts up to 65,536 keys in a few microseconds.  RR_NO_ORDER=1 switches it off (A/B runs).
    e->order = nullptr; e->cost = nullptr; e->ngroups = 0;
    {
        int apb = 1;
        dispatch(e, [&](auto c) { using CC = decltype(c); apb = arenas_per_block<CC>(); return 0; });
        const int ng = (n + apb - 1) / apb;
        const char *no = getenv("RR_NO_ORDER");
        if (ng > 2048 && ng <= 65536 && !(no && atoi(no))) {
            if (hipMalloc((void **)&e->order, sizeof(uint32_t) * ng) == hipSuccess && hipMalloc((void **)&e->cost, sizeof(uint32_t) * ng) == hipSuccess) {
                e->ngroups = ng;
                hipLaunchKernelGGL(k_iota, dim3((ng + 255) / 256), dim3(256), 0, 0, e->order, ng);
                (void)hipMemset(e->cost, 0, sizeof(uint32_t) * ng);
            } else {
                if (e->order) (void)hipFree(e->order);
                e->order = nullptr; e->cost = nullptr;
            }
        }
    }
    // constructor placement (RR_EnvBase.py:111-116): episode 0 of the counter RNG
    dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_reset<CC, float>), arena_grid<CC>(n), wave_block(), 0, 0, params_of<RR>(e), (typename CC::Store *)e->recs,
                           e->irecs, n, (const uint8_t *)nullptr, 1, (float *)nullptr, (float *)nullptr);
        return 0;
    });
    he = hipGetLastError();
    if (he == hipSuccess) he = hipDeviceSynchronize();
    if (he != hipSuccess) {
        (void)hipFree(e->recs);
        if (e->snap) (void)hipFree(e->snap);
        if (e->isnap) (void)hipFree(e->isnap);
        if (e->order) (void)hipFree(e->order);
        if (e->cost) (void)hipFree(e->cost);
        delete e;
        return fail(-2, std::string("rr_create: init kernel: ") + hipGetErrorString(he));
    }
    if (cfg->step_budget_clocks) {
        if (int rc = rr_set_step_budget(e, cfg->step_budget_clocks)) { rr_destroy(e); return rc; }
    }
    *out = e;
    return 0;
}

int rr_destroy(rr_env *e) {
    if (!e) return 0;
    DeviceGuard guard(e->cfg.device);
    (void)hipFree(e->recs);
    if (e->snap) (void)hipFree(e->snap);
    if (e->isnap) (void)hipFree(e->isnap);
    if (e->xs) (void)hipFree(e->xs);
    if (e->gs) (void)hipFree(e->gs);
    if (e->status_buf) (void)hipFree(e->status_buf);
    if (e->order) (void)hipFree(e->order);
    if (e->cost) (void)hipFree(e->cost);
    if (e->park) (void)hipFree(e->park);
    delete e;
    return 0;
}

int rr_reset(rr_env *e, const uint8_t *mask, float *obs, float *obs_g, void *stream) {
    if (!e) return fail(-1, "rr_reset: null handle");
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_reset<CC, float>), arena_grid<CC>(n), wave_block(), 0, (hipStream_t)stream, params_of<RR>(e),
                           (typename CC::Store *)e->recs, e->irecs, n, mask, 0, obs, obs_g);
        if (e->track_prior && e->xs) // a re-placed arena has no prior step yet: its copies restart from the new poses
            hipLaunchKernelGGL((k_extras_begin<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const typename CC::Store *)e->recs, n, (RR *)e->xs, mask);
        if (e->gs) hipLaunchKernelGGL((k_goal_clear<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, n, e->gs, mask); // Goal.on_reset
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

static int check_step_args(rr_env *e, const void *act, int32_t na, const void *obs, const void *reward, const void *done) {
    if (!e) return fail(-1, "rr_step: null handle");
    if (!act || !obs || !reward || !done) return fail(-1, "rr_step: actions/obs/reward/done must be non-null");
    const int nr = e->cfg.nr_happy + e->cfg.nr_grumpy;
    if (na < 0 || na > nr) // RR_EnvBase.py:621-622 raises "commands but only N robots"
        return fail(-1, "rr_step: more actions than robots");
    return 0;
}

// one GameEnv.step for every arena; with a non-default keeper program the side kernels bracket the step kernel
extern "C++" {
template <typename O>
static int step_impl(rr_env *e, const int32_t *actions, const float *thrust, int32_t na, O *obs, O *reward, uint8_t *done,
                     O *obs_g, O *reward_g, int32_t *status, void *stream, int nsteps = 1, int repeat = 0) {
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    hipStream_t s = (hipStream_t)stream;
    if ((e->custom_prog || e->track_prior || e->gs) && !status) status = e->status_buf; // the side kernels need the NaughtyBots / WAS_RESET bits
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        if constexpr (std::is_same<O, double>::value && !std::is_same<RR, double>::value) {
            return fail(-1, "fp64 outputs need a handle created with RR_DTYPE_F64");
        } else {
            if (e->custom_prog || e->track_prior)
                hipLaunchKernelGGL((k_extras_begin<CC>), dim3((n + 127) / 128), dim3(128), 0, s, (const typename CC::Store *)e->recs, n, (RR *)e->xs, (const uint8_t *)nullptr);
            if constexpr (CC::MIXED) { // fp32 state: the plain single-step kernel only (rr_set_step_budget / rr_rollout refuse the handle)
                if (nsteps != 1 || e->park) return fail(-1, "RR_DTYPE_F32_STATE: rr_step only (no rr_rollout, no step budget)");
                hipLaunchKernelGGL((k_step<CC, O, false>), arena_grid<CC>(n), wave_block(), 0, s, params_of<RR>(e), (typename CC::Store *)e->recs, e->irecs, n,
                                   actions, thrust, (int)na, obs, reward, done, obs_g, reward_g, status, (const uint32_t *)e->order, e->cost,
                                   1, 0, e->snap, e->isnap);
            } else
            if (nsteps == 1 && e->park)
                hipLaunchKernelGGL((k_step<CC, O, false, true>), arena_grid<CC>(n), wave_block(), 0, s, params_of<RR>(e), (typename CC::Store *)e->recs, e->irecs, n,
                                   actions, thrust, (int)na, obs, reward, done, obs_g, reward_g, status, (const uint32_t *)e->order, e->cost,
                                   1, 0, e->snap, e->isnap, e->park, e->budget);
            else if (nsteps == 1)
                hipLaunchKernelGGL((k_step<CC, O, false>), arena_grid<CC>(n), wave_block(), 0, s, params_of<RR>(e), (typename CC::Store *)e->recs, e->irecs, n,
                                   actions, thrust, (int)na, obs, reward, done, obs_g, reward_g, status, (const uint32_t *)e->order, e->cost,
                                   1, 0, e->snap, e->isnap);
            else if constexpr (std::is_same<O, float>::value && CC::VW == default_vw<CC>()) // default lane widths only (build time)
                hipLaunchKernelGGL((k_step<CC, O, true>), arena_grid<CC>(n), wave_block(), 0, s, params_of<RR>(e), (typename CC::Store *)e->recs, e->irecs, n,
                                   actions, thrust, (int)na, obs, reward, done, obs_g, reward_g, status, (const uint32_t *)e->order, e->cost,
                                   nsteps, repeat, e->snap, e->isnap);
            else
                return fail(-1, "rr_rollout: built for float outputs and the default lane widths (RR_VW unset) only");
            if (e->order) hipLaunchKernelGGL(k_order, dim3(1), dim3(ORDER_THREADS), 0, s, (const uint32_t *)e->cost, e->order, e->ngroups);
            if (e->custom_prog || e->track_prior)
                hipLaunchKernelGGL((k_extras_end<CC, O>), dim3((n + 127) / 128), dim3(128), 0, s, params_of<RR>(e),
                                   (typename CC::Store *)e->recs, n, (RR *)e->xs, e->prog, e->custom_prog ? 1 : 0, reward, reward_g, status,
                                   (const uint8_t *)done);
            if (e->gs) {
                int bd = 0;
                for (int k = 0; k < e->prog.n; k++) bd |= e->prog.id[k] == KEEPER_BASEDESTRUCTION;
                hipLaunchKernelGGL((k_goal<CC, O>), dim3((n + 127) / 128), dim3(128), 0, s, params_of<RR>(e), (typename CC::Store *)e->recs, e->irecs, n,
                                   e->gs, bd, reward, reward_g, done, status);
            }
            return 0;
        }
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
} // extern "C++"

int rr_step(rr_env *e, const int32_t *actions, int32_t na, float *obs, float *reward, uint8_t *done, float *obs_g,
            float *reward_g, int32_t *status, void *stream) {
    if (int rc = check_step_args(e, actions, na, obs, reward, done)) return rc;
    return step_impl<float>(e, actions, nullptr, na, obs, reward, done, obs_g, reward_g, status, stream);
}
int rr_rollout(rr_env *e, const int32_t *actions, int32_t na, int32_t nsteps, int32_t repeat, float *obs, float *reward,
               uint8_t *done, float *obs_g, float *reward_g, int32_t *status, void *stream) {
    if (int rc = check_step_args(e, actions, na, obs, reward, done)) return rc;
    if (nsteps < 1 || nsteps > 4096) return fail(-1, "rr_rollout: nsteps must be in 1..4096");
    if (e->custom_prog || e->track_prior || e->gs)
        return fail(-1, "rr_rollout: the side kernels of a custom reward program / prior-step tracking / goal scoring bracket single steps; use rr_step");
    if (e->park) return fail(-1, "rr_rollout: not on a handle that has had a step budget (arenas may be parked mid-step); use rr_step");
    return step_impl<float>(e, actions, nullptr, na, obs, reward, done, obs_g, reward_g, status, stream, (int)nsteps, repeat != 0);
}
int rr_step_thrust(rr_env *e, const float *thrust, int32_t nk, float *obs, float *reward, uint8_t *done, float *obs_g,
                   float *reward_g, int32_t *status, void *stream) {
    if (int rc = check_step_args(e, thrust, nk, obs, reward, done)) return rc;
    return step_impl<float>(e, nullptr, thrust, nk, obs, reward, done, obs_g, reward_g, status, stream);
}
int rr_step_f64(rr_env *e, const int32_t *actions, int32_t na, double *obs, double *reward, uint8_t *done, double *obs_g,
                double *reward_g, int32_t *status, void *stream) {
    if (int rc = check_step_args(e, actions, na, obs, reward, done)) return rc;
    return step_impl<double>(e, actions, nullptr, na, obs, reward, done, obs_g, reward_g, status, stream);
}
int rr_step_thrust_f64(rr_env *e, const float *thrust, int32_t nk, double *obs, double *reward, uint8_t *done, double *obs_g,
                       double *reward_g, int32_t *status, void *stream) {
    if (int rc = check_step_args(e, thrust, nk, obs, reward, done)) return rc;
    return step_impl<double>(e, nullptr, thrust, nk, obs, reward, done, obs_g, reward_g, status, stream);
}

static int ensure_snapshot_buffer(rr_env *e) { // on_step_begin snapshot of every arena (xs_stride reals each)
    if (e->xs) return 0;
    const size_t rsz = e->cfg.dtype == RR_DTYPE_F32 ? 4 : 8; // (arithmetic reals)
    const size_t nr = (size_t)(e->cfg.nr_happy + e->cfg.nr_grumpy), nb = (size_t)(e->cfg.nb_pos + e->cfg.nb_neg);
    HIP_TRY(hipMalloc(&e->xs, rsz * (3 * nr + 1 + 2 * nb) * (size_t)e->cfg.num_envs));
    return 0;
}

int rr_set_reward_program(rr_env *e, const int32_t *ids, int32_t n) {
    if (!e || (!ids && n > 0)) return fail(-1, "rr_set_reward_program: null argument");
    if (n < 0 || n > 8) return fail(-1, "rr_set_reward_program: at most 8 keepers");
    for (int i = 0; i < n; i++)
        if (ids[i] < KEEPER_NAUGHTY || ids[i] > KEEPER_PUSHNEG) return fail(-1, "rr_set_reward_program: unknown keeper id");
    DeviceGuard guard(e->cfg.device);
    e->prog.n = n;
    for (int i = 0; i < n; i++) e->prog.id[i] = ids[i];
    e->custom_prog = !(n == 3 && ids[0] == KEEPER_NAUGHTY && ids[1] == KEEPER_CHASE && ids[2] == KEEPER_PUSHPOS);
    e->spd.acc_external = e->spf.acc_external = e->custom_prog ? 1 : 0; // k_extras_end keeps the episode returns then
    if (e->custom_prog) { // allocated here, never inside rr_step (keeps the step launch-only)
        if (int rc = ensure_snapshot_buffer(e)) return rc;
        if (!e->status_buf) HIP_TRY(hipMalloc((void **)&e->status_buf, sizeof(int32_t) * (size_t)e->cfg.num_envs));
    }
    return 0;
}

int rr_set_step_budget(rr_env *e, uint32_t clocks) {
    if (!e) return fail(-1, "rr_set_step_budget: null handle");
    if (clocks && e->cfg.dtype == RR_DTYPE_F32_STATE) // (a parked arena's record would be rounded to fp32 in the middle of its step)
        return fail(-1, "rr_set_step_budget: not with RR_DTYPE_F32_STATE");
    DeviceGuard guard(e->cfg.device);
    if (clocks && !e->park) {
        size_t words = 0;
        dispatch(e, [&](auto c) { using CC = decltype(c); words = Arena<CC>::PARK_WORDS; return 0; });
        if (hipMalloc((void **)&e->park, 4 * words * (size_t)e->cfg.num_envs) != hipSuccess) {
            e->park = nullptr;
            return fail(-3, "rr_set_step_budget: out of device memory");
        }
    }
    // 0 after a budget: arenas may still be parked mid-step, so the budgeted kernel stays in charge with a budget nothing
    // exceeds (it resumes them and parks nothing); a handle that never had a budget keeps the default kernel
    e->budget = clocks ? clocks : 0xFFFFFFFFu;
    return 0;
}

int rr_set_goal_scoring(rr_env *e, int32_t on, void *stream) {
    if (!e) return fail(-1, "rr_set_goal_scoring: null handle");
    DeviceGuard guard(e->cfg.device);
    if (!on) {
        if (e->gs) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); (void)hipFree(e->gs); e->gs = nullptr; }
        return 0;
    }
    const int n = e->cfg.num_envs;
    if (!e->status_buf) HIP_TRY(hipMalloc((void **)&e->status_buf, sizeof(int32_t) * (size_t)n));
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c);
        if (!e->gs && hipMalloc((void **)&e->gs, sizeof(int32_t) * (size_t)gs_stride<CC>() * (size_t)n) != hipSuccess) {
            e->gs = nullptr;
            return fail(-3, "rr_set_goal_scoring: out of device memory");
        }
        hipLaunchKernelGGL((k_goal_clear<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, n, e->gs, (const uint8_t *)nullptr);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
int rr_goal_scores(rr_env *e, int32_t *scores, void *stream) {
    if (!e || !scores) return fail(-1, "rr_goal_scores: null argument");
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    if (!e->gs) { HIP_TRY(hipMemsetAsync(scores, 0, sizeof(int32_t) * 2 * (size_t)n, (hipStream_t)stream)); return 0; } // the live reference: identically 0
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c);
        hipLaunchKernelGGL((k_goal_scores<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, n, (const int32_t *)e->gs, scores);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

int rr_track_prior_step(rr_env *e, int32_t on, void *stream) {
    if (!e) return fail(-1, "rr_track_prior_step: null handle");
    DeviceGuard guard(e->cfg.device);
    e->track_prior = on != 0;
    if (!e->track_prior) return 0;
    if (int rc = ensure_snapshot_buffer(e)) return rc;
    if (!e->status_buf) HIP_TRY(hipMalloc((void **)&e->status_buf, sizeof(int32_t) * (size_t)e->cfg.num_envs));
    // until the first step the prior-step copies are copies of the current state (the reference's hold the stale
    // pre-placement pose there: Robot.on_reset / Ball.on_reset copy BEFORE _set_random_positions moves the sprites)
    const int n = e->cfg.num_envs;
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_extras_begin<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const typename CC::Store *)e->recs, n, (RR *)e->xs, (const uint8_t *)nullptr);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C++" {
template <typename O>
static int observe_kind_impl(rr_env *e, int32_t kind, int32_t team, int32_t ridx, int32_t bidx, O *obs, int32_t out_dim, void *stream) {
    if (!e || !obs) return fail(-1, "rr_observe_kind: null argument");
    const int nr = e->cfg.nr_happy + e->cfg.nr_grumpy, nb = e->cfg.nb_pos + e->cfg.nb_neg;
    if ((team != 1 && team != -1) || ridx >= nr || bidx >= nb) return fail(-1, "rr_observe_kind: bad team/robot/ball index");
    const int want = kind == OBS_V2 || kind == OBS_V1 ? 11 : kind == OBS_BASIC ? 5 : kind == OBS_ALLCOORDS ? 3 * nr + 2 * nb :
                     kind == OBS_ALLCOORDS_PRIOR ? 6 * nr + 4 * nb : -1;
    if (want < 0) return fail(-1, "rr_observe_kind: unknown observer kind");
    if (out_dim != want) return fail(-1, "rr_observe_kind: out_dim does not match the observer's size");
    if (kind == OBS_ALLCOORDS_PRIOR && !(e->track_prior && e->xs))
        return fail(-1, "rr_observe_kind: AllCoords_WithPrior needs rr_track_prior_step(env, 1) before the step it looks back on");
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        if constexpr (std::is_same<O, double>::value && !std::is_same<RR, double>::value) {
            return fail(-1, "fp64 outputs need a handle created with RR_DTYPE_F64");
        } else {
            if (kind == OBS_V2)
                hipLaunchKernelGGL((k_observe<CC, O>), arena_grid<CC>(n), wave_block(), 0, (hipStream_t)stream, params_of<RR>(e),
                                   (const typename CC::Store *)e->recs, (const int32_t *)e->irecs, n, (int)team, (int)ridx, (int)bidx, obs);
            else
                hipLaunchKernelGGL((k_observe_kind<CC, O>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, params_of<RR>(e),
                                   (const typename CC::Store *)e->recs, n, (int)kind, (int)team, (int)ridx, (int)bidx, obs, (int)out_dim,
                                   kind == OBS_ALLCOORDS_PRIOR ? (const RR *)e->xs : (const RR *)nullptr);
            return 0;
        }
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
} // extern "C++"
int rr_observe_kind(rr_env *e, int32_t kind, int32_t team, int32_t ridx, int32_t bidx, float *obs, int32_t out_dim, void *stream) {
    return observe_kind_impl<float>(e, kind, team, ridx, bidx, obs, out_dim, stream);
}
int rr_observe_kind_f64(rr_env *e, int32_t kind, int32_t team, int32_t ridx, int32_t bidx, double *obs, int32_t out_dim, void *stream) {
    return observe_kind_impl<double>(e, kind, team, ridx, bidx, obs, out_dim, stream);
}

static int check_obs_args(rr_env *e, const void *obs, int32_t team, int32_t ridx, int32_t bidx) {
    if (!e || !obs) return fail(-1, "rr_observe: null argument");
    const int nr = e->cfg.nr_happy + e->cfg.nr_grumpy, nb = e->cfg.nb_pos + e->cfg.nb_neg;
    if ((team != 1 && team != -1) || ridx >= nr || bidx >= nb) return fail(-1, "rr_observe: bad team/robot/ball index");
    return 0;
}
int rr_observe(rr_env *e, int32_t team, int32_t ridx, int32_t bidx, float *obs, void *stream) {
    if (int rc = check_obs_args(e, obs, team, ridx, bidx)) return rc;
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_observe<CC, float>), arena_grid<CC>(n), wave_block(), 0, (hipStream_t)stream, params_of<RR>(e),
                           (const typename CC::Store *)e->recs, (const int32_t *)e->irecs, n, (int)team, (int)ridx, (int)bidx, obs);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
int rr_observe_f64(rr_env *e, int32_t team, int32_t ridx, int32_t bidx, double *obs, void *stream) {
    if (int rc = check_obs_args(e, obs, team, ridx, bidx)) return rc;
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        if constexpr (std::is_same<RR, double>::value) {
            hipLaunchKernelGGL((k_observe<CC, double>), arena_grid<CC>(n), wave_block(), 0, (hipStream_t)stream,
                               params_of<RR>(e), (const typename CC::Store *)e->recs, (const int32_t *)e->irecs, n, (int)team, (int)ridx,
                               (int)bidx, obs);
            return 0;
        } else {
            return fail(-1, "rr_observe_f64: handle was created with RR_DTYPE_F32");
        }
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

int rr_set_state(rr_env *e, const double *robots, const int32_t *ri, const double *balls, const int32_t *step, void *stream) {
    if (!e || !robots || !ri || !balls || !step) return fail(-1, "rr_set_state: null argument");
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_set_state<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, (typename CC::Store *)e->recs, e->irecs,
                           n, robots, ri, balls, step);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
int rr_get_state(rr_env *e, double *robots, int32_t *ri, double *balls, int32_t *step, void *stream) {
    if (!e || !robots || !ri || !balls || !step) return fail(-1, "rr_get_state: null argument");
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_get_state<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const typename CC::Store *)e->recs,
                           (const int32_t *)e->irecs, n, robots, ri, balls, step);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
static int set_poses_impl(rr_env *e, const uint8_t *mask, const double *rxyr, const double *bxyv, float *obs, float *obs_g, void *stream) {
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_set_poses<CC, float>), arena_grid<CC>(n), wave_block(), 0, (hipStream_t)stream, params_of<RR>(e),
                           (typename CC::Store *)e->recs, e->irecs, n, rxyr, bxyv, mask, obs, obs_g);
        if (e->track_prior && e->xs) // like rr_reset: a re-placed arena has no prior step yet
            hipLaunchKernelGGL((k_extras_begin<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const typename CC::Store *)e->recs, n, (RR *)e->xs, mask);
        if (e->gs) hipLaunchKernelGGL((k_goal_clear<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, n, e->gs, mask);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
int rr_set_poses(rr_env *e, const double *rxyr, const double *bxyv, void *stream) {
    if (!e || !rxyr || !bxyv) return fail(-1, "rr_set_poses: null argument");
    return set_poses_impl(e, nullptr, rxyr, bxyv, nullptr, nullptr, stream);
}
int rr_reset_to_poses(rr_env *e, const uint8_t *mask, const double *rxyr, const double *bxyv, float *obs, float *obs_g, void *stream) {
    if (!e || !rxyr || !bxyv) return fail(-1, "rr_reset_to_poses: null argument");
    return set_poses_impl(e, mask, rxyr, bxyv, obs, obs_g, stream);
}
static int episode_state_impl(rr_env *e, int32_t *ints, double *acc, int set, void *stream) {
    if (!e || !ints || !acc) return fail(-1, "rr_get/set_episode_state: null argument");
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_episode_state<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, (typename CC::Store *)e->recs, e->irecs, n, ints, acc, set);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
static int scratch_rect_impl(rr_env *e, double *xy, int set, void *stream) {
    if (!e || !xy) return fail(-1, "rr_get/set_scratch_rect: null argument");
    if (!RR_CARRY) return fail(-3, "rr_get/set_scratch_rect: only the parity build (libroborugby_amd_exact.so) carries the scratch rect");
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_scratch_rect<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, (typename CC::Store *)e->recs, n, xy, set);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
int rr_get_scratch_rect(rr_env *e, double *xy, void *stream) { return scratch_rect_impl(e, xy, 0, stream); }
int rr_set_scratch_rect(rr_env *e, const double *xy, void *stream) { return scratch_rect_impl(e, const_cast<double *>(xy), 1, stream); }
int rr_get_episode_state(rr_env *e, int32_t *ints, double *acc, void *stream) { return episode_state_impl(e, ints, acc, 0, stream); }
int rr_set_episode_state(rr_env *e, const int32_t *ints, const double *acc, void *stream) {
    return episode_state_impl(e, const_cast<int32_t *>(ints), const_cast<double *>(acc), 1, stream);
}
int rr_episode_stats(rr_env *e, float *lr, float *lrg, int32_t *ll, int32_t *cnt, void *stream) {
    if (!e) return fail(-1, "rr_episode_stats: null handle");
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    int rc = dispatch(e, [&](auto c) {
        using CC = decltype(c); using RR = typename CC::Real;
        hipLaunchKernelGGL((k_episode_stats<CC>), dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const typename CC::Store *)e->recs,
                           (const int32_t *)e->irecs, n, lr, lrg, ll, cnt);
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
int rr_policy_chase(rr_env *e, const float *obs, const int32_t *step_of, uint32_t step, float noise, uint64_t seed, int32_t *actions,
                    int32_t na, void *stream) {
    if (!e || !obs || !actions) return fail(-1, "rr_policy_chase: null argument");
    const int nr = e->cfg.nr_happy + e->cfg.nr_grumpy;
    if (na < 1 || na > nr || na > 8) return fail(-1, "rr_policy_chase: between 1 action and one per robot");
    if (!(noise >= 0.0f && noise <= 1.0f)) return fail(-1, "rr_policy_chase: noise must be in [0, 1]");
    const int n = e->cfg.num_envs;
    DeviceGuard guard(e->cfg.device);
    const uint32_t nz = noise >= 1.0f ? 0xFFFFFFFFu : (uint32_t)((double)noise * 4294967296.0);
    hipLaunchKernelGGL(k_policy_chase, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, obs, n, (int)na, nz, seed,
                       e->cfg.arena_offset, step_of, step, actions);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rr_probe_hbm_copy(void *dst, const void *src, size_t bytes, void *stream) {
    if (!dst || !src || bytes < 16 || (bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15))
        return fail(-1, "rr_probe_hbm_copy: 16-byte aligned buffers and a multiple of 16 bytes, please");
    const size_t n16 = bytes / 16;
    const size_t blocks = (n16 + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return fail(-1, "rr_probe_hbm_copy: at most 8 TiB per call");
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)src, (u32x4 *)dst, n16);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rr_state_bytes_per_env(const rr_env *e, int64_t *bytes) {
    if (!e || !bytes) return fail(-1, "rr_state_bytes_per_env: null argument");
    *bytes = (int64_t)(e->rec_bytes + e->irec_bytes);
    return 0;
}

int rr_lanes_per_env(const rr_env *e, int32_t *lanes) {
    if (!e || !lanes) return fail(-1, "rr_lanes_per_env: null argument");
    *lanes = e->vw;
    return 0;
}

#ifdef RR_PROFILE_PHASES
// diagnostic build only: copy out / clear the per-phase cycle totals
int rr_debug_phase_cycles(unsigned long long *host32, int clear) {
    if (hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_rr_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return -2;
    if (clear) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rr_prof), z, sizeof z) != hipSuccess) return -2;
    }
    return 0;
}
// diagnostic build only: [start, end] clock stamps of the first n wavefronts of the last k_step launch
int rr_debug_wave_times(unsigned long long *host, int n) {
    if (n > 65536) n = 65536;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rr_wave_t), (size_t)2 * n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

} // extern "C"
