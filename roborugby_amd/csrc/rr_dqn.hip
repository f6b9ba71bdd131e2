// rr_dqn.hip -- fused DQN update for the reference's network (Training_DQN_pytorch.py:25-67: Linear 11 -> 256 -> 256 -> 8,
// ReLU, MSE loss, Adam) at the wide batches of the batched trainer (config 5: B = 32,768 samples per gradient step).
//
// The reference's learn() (Training_DQN_pytorch.py:151-191) is, per sample: Q_eval(s)[a], max_a' Q_target(s') (0 on terminal),
// y = r + gamma * max, loss = mean (y - Q_eval(s)[a])^2, backward, Adam.  Stock PyTorch runs that as ~20 small kernels per update
// (three GEMM pairs + bias / ReLU / gather / mask / max / MSE / threshold / column sums / index kernels: 74 % of a config-5 vector
// step, profiles/r02/dqn_stage_timings.txt).  Here it is TWO launches:
//
//   k_dqn_fwd_bwd   one workgroup per CU, persistent over 64-sample tiles.  A tile's activations never leave the CU: the two
//                   256-wide hidden layers live in LDS (transposed, [feature][sample], row stride 65 words: conflict-free as MFMA
//                   A operand and as epilogue target), the 256 x 256 layer runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32;
//                   fp32 because the reference trains in fp32 -- no bf16 shortcut), weights stream from L2 (282 KB for both nets).
//                   Order per tile: target net forward (s') -> y; eval net forward (s) -> h1, h2, q; dq; dW3 / dh2 (the loss
//                   gradient touches ONE action per sample: VALU); dW2 += dh2^T h1 (MFMA, accumulators stay in registers across
//                   tiles: 256 VGPRs per lane at one wave per SIMD); dh1 = (dh2 W2) * relu' (MFMA); dW1 / biases (VALU).
//                   Ends by writing its partial gradient (70,921 floats) once.
//   k_dqn_reduce_adam   sums the partials in workgroup order (deterministic), applies torch.optim.Adam's update to the
//                   parameters in place (PyTorch owns them: plain device pointers), keeps the moments, emits the loss.
//
// Same math as torch autograd up to fp32 summation order (tests/test_gpu_dqn_fused.py: every parameter gradient within 1e-5 of
// autograd's, relative to the gradient's scale).  C-ABI: rr_dqn_update / rr_dqn_grads in include/roborugby_amd.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <string>

#include "../../include/roborugby_amd.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int H = 256, IN = 11, NA = 8;
constexpr int TM = 64;        // samples per tile
constexpr int LDX = TM + 1;   // LDS row stride of the transposed activations (words)
constexpr int NT = 256;       // threads per workgroup: 4 wavefronts, one per SIMD
// partial-gradient layout (floats), also the layout of the flat Adam moments
constexpr int OFF_W2 = 0, OFF_W1 = OFF_W2 + H * H, OFF_W3 = OFF_W1 + H * IN, OFF_B1 = OFF_W3 + NA * H, OFF_B2 = OFF_B1 + H,
              OFF_B3 = OFF_B2 + H, OFF_LOSS = OFF_B3 + NA, P_COUNT = OFF_LOSS, P_STRIDE = ((OFF_LOSS + 1 + 63) / 64) * 64;

struct Net { const float *w1, *b1, *w2, *b2, *w3, *b3; };
struct Batch {
    const float *state, *new_state, *reward;
    const int64_t *action;
    const uint8_t *terminal;
    const int64_t *idx; // sampled rows of the replay memory (nullptr: rows 0..batch-1)
    int batch;
    float gamma;
};

__device__ __forceinline__ f16v mfma(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// row of a 32 x 32 accumulator tile held in register r of a lane in half `half` (lanes 32-63: half = 1); its column is lane & 31
__device__ __forceinline__ int acc_row(int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); }

// hidden layers of one net for the tile in Sb ([12][LDX], row 11 = 0): X0 = relu(fc1), X1 = relu(fc2), Q[a][m] = fc3
__device__ __forceinline__ void forward_tile(const Net &net, const float *Sb, float *X0, float *X1, float *Q, int w, int half, int c, int l) {
    f16v acc[4];
    // ---- fc1: K = 11 (padded to 12), output tiles: n-tiles {2w, 2w+1} x m-tiles {0, 1}
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = (f16v)0.0f;
#pragma unroll
    for (int p = 0; p < 6; p++) {
        const int k = 2 * p + half;
        const float a0 = Sb[k * LDX + c], a1 = Sb[k * LDX + 32 + c];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int n = (2 * w + nt) * 32 + c;
            const float b = k < IN ? net.w1[n * IN + k] : 0.0f;
            acc[nt * 2 + 0] = mfma(a0, b, acc[nt * 2 + 0]);
            acc[nt * 2 + 1] = mfma(a1, b, acc[nt * 2 + 1]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int n = (2 * w + nt) * 32 + c;
        const float bias = net.b1[n];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float v = acc[nt * 2 + mt][r] + bias;
                X0[n * LDX + mt * 32 + acc_row(r, half)] = v > 0.0f ? v : 0.0f;
            }
    }
    __syncthreads();
    // ---- fc2: K = 256, eight k per round: a lane's float4 of W2 covers k = 8 kg + 4 half .. + 3 (the two halves of the
    // wavefront are the two k slots of an MFMA; which k a slot holds is free as long as A and B agree)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = (f16v)0.0f;
    float4 bw[2], bw_next[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++) bw[nt] = *reinterpret_cast<const float4 *>(&net.w2[((2 * w + nt) * 32 + c) * H + 4 * half]);
#pragma unroll 2
    for (int kg = 0; kg < H / 8; kg++) {
        const int kb = kg * 8 + 4 * half;
        if (kg + 1 < H / 8) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++) bw_next[nt] = *reinterpret_cast<const float4 *>(&net.w2[((2 * w + nt) * 32 + c) * H + kb + 8]);
        }
        const float bv[2][4] = { { bw[0].x, bw[0].y, bw[0].z, bw[0].w }, { bw[1].x, bw[1].y, bw[1].z, bw[1].w } };
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float a0 = X0[(kb + j) * LDX + c], a1 = X0[(kb + j) * LDX + 32 + c];
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                acc[nt * 2 + 0] = mfma(a0, bv[nt][j], acc[nt * 2 + 0]);
                acc[nt * 2 + 1] = mfma(a1, bv[nt][j], acc[nt * 2 + 1]);
            }
        }
        bw[0] = bw_next[0]; bw[1] = bw_next[1];
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int n = (2 * w + nt) * 32 + c;
        const float bias = net.b2[n];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float v = acc[nt * 2 + mt][r] + bias;
                X1[n * LDX + mt * 32 + acc_row(r, half)] = v > 0.0f ? v : 0.0f;
            }
    }
    __syncthreads();
    // ---- fc3 (8 outputs: VALU): wavefront w computes actions 2w, 2w+1 for sample m = lane; the weights are wave-uniform
    {
        const float *w3a = net.w3 + (2 * w) * H, *w3b = net.w3 + (2 * w + 1) * H;
        float q0 = 0.0f, q1 = 0.0f;
#pragma unroll 8
        for (int k = 0; k < H; k++) {
            const float x = X1[k * LDX + l];
            q0 = fmaf(x, w3a[k], q0);
            q1 = fmaf(x, w3b[k], q1);
        }
        Q[(2 * w) * TM + l] = q0 + net.b3[2 * w];
        Q[(2 * w + 1) * TM + l] = q1 + net.b3[2 * w + 1];
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT, 1) void k_dqn_fwd_bwd(Net ev, Net tg, Batch bt, float *partials) {
    __shared__ float X0[H * LDX], X1[H * LDX], S[12 * LDX], S2[12 * LDX], Q[NA * TM], Y[TM], DQ[TM], RED[NT];
    __shared__ int ACT[TM];
    const int t = threadIdx.x, w = t >> 6, l = t & 63, half = l >> 5, c = l & 31;
    // gradient accumulators that stay in registers over all of this workgroup's tiles
    f16v dw2[16]; // [it (i-tile 2w+it)][jt]: dW2[i][j], i = out feature of fc2, j = in feature
#pragma unroll
    for (int q = 0; q < 16; q++) dw2[q] = (f16v)0.0f;
    float dw1[IN], dw3[NA], w3r[NA];
#pragma unroll
    for (int q = 0; q < IN; q++) dw1[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < NA; q++) { dw3[q] = 0.0f; w3r[q] = ev.w3[q * H + t]; }
    float db1 = 0.0f, db2 = 0.0f, db3 = 0.0f, loss = 0.0f;
    const float inv_b2 = 2.0f / (float)bt.batch;
    const int ntiles = bt.batch / TM;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ---- gather the tile: s, s' transposed into LDS, action, reward, terminal
        for (int e = t; e < TM * IN; e += NT) {
            const int m = e / IN, ci = e - m * IN;
            const int64_t row = bt.idx ? bt.idx[tile * TM + m] : (int64_t)(tile * TM + m);
            S[ci * LDX + m] = bt.state[row * IN + ci];
            S2[ci * LDX + m] = bt.new_state[row * IN + ci];
        }
        if (t < TM) {
            const int64_t row = bt.idx ? bt.idx[tile * TM + t] : (int64_t)(tile * TM + t);
            S[11 * LDX + t] = 0.0f; S2[11 * LDX + t] = 0.0f;
            ACT[t] = (int)bt.action[row];
            Y[t] = bt.reward[row];
            DQ[t] = bt.terminal[row] ? 0.0f : 1.0f; // (mask of the bootstrap term, replaced by dq below)
        }
        __syncthreads();
        // ---- target net on s': y = r + gamma * max_a Q_target(s')  (0 for a terminal transition)
        forward_tile(tg, S2, X0, X1, Q, w, half, c, l);
        if (t < TM) {
            float mx = Q[t];
#pragma unroll
            for (int a = 1; a < NA; a++) mx = fmaxf(mx, Q[a * TM + t]);
            Y[t] = Y[t] + bt.gamma * (DQ[t] != 0.0f ? mx : 0.0f);
        }
        __syncthreads();
        // ---- eval net on s: h1 -> X0, h2 -> X1, q -> Q; loss gradient dq = 2 (q[a] - y) / B
        forward_tile(ev, S, X0, X1, Q, w, half, c, l);
        if (t < TM) {
            const float diff = Q[ACT[t] * TM + t] - Y[t];
            DQ[t] = inv_b2 * diff;
            loss += diff * diff;
        }
        __syncthreads();
        // ---- fc3 backward (one action per sample): thread t owns hidden unit n = t.  dW3[a][n] += dq h2, then X1 <- dh2
        {
            float *row = &X1[t * LDX];
            for (int m = 0; m < TM; m++) {
                const int am = ACT[m];      // wave-uniform: LDS broadcast, scalar branch below
                const float d = DQ[m];
                const float h = row[m];
                const float dh = d * h;
                float wsel = w3r[0];
#pragma unroll
                for (int a = 0; a < NA; a++) { dw3[a] += (a == am) ? dh : 0.0f; wsel = (a == am) ? w3r[a] : wsel; }
                const float g = h > 0.0f ? d * wsel : 0.0f;
                row[m] = g;
                db2 += g;
            }
            if (t < NA) {
                for (int m = 0; m < TM; m++) db3 += (ACT[m] == t) ? DQ[m] : 0.0f;
            }
        }
        __syncthreads();
        // ---- dW2[i][j] += sum_m dh2[m][i] h1[m][j]: reduction over the tile's 64 samples, two per MFMA
#pragma unroll 2
        for (int p = 0; p < TM / 2; p++) {
            const int mm = 2 * p + half;
            const float a0 = X1[((2 * w) * 32 + c) * LDX + mm], a1 = X1[((2 * w + 1) * 32 + c) * LDX + mm];
#pragma unroll
            for (int jt = 0; jt < 8; jt++) {
                const float b = X0[(jt * 32 + c) * LDX + mm];
                dw2[jt] = mfma(a0, b, dw2[jt]);
                dw2[8 + jt] = mfma(a1, b, dw2[8 + jt]);
            }
        }
        // ---- dh1[m][k] = sum_n dh2[m][n] W2[n][k] (then * relu'(h1)): output k-tiles {2w, 2w+1} x m-tiles {0, 1}
        {
            f16v acc[4];
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = (f16v)0.0f;
            const float *w2c0 = ev.w2 + (2 * w) * 32 + c, *w2c1 = ev.w2 + (2 * w + 1) * 32 + c;
#pragma unroll 4
            for (int np = 0; np < H / 2; np++) {
                const int n = 2 * np + half;
                const float a0 = X1[n * LDX + c], a1 = X1[n * LDX + 32 + c];
                const float b0 = w2c0[n * H], b1 = w2c1[n * H];
                acc[0] = mfma(a0, b0, acc[0]);
                acc[1] = mfma(a1, b0, acc[1]);
                acc[2] = mfma(a0, b1, acc[2]);
                acc[3] = mfma(a1, b1, acc[3]);
            }
            __syncthreads(); // every wavefront has read h1 (X0) for dW2: it can be overwritten with dh1 now
#pragma unroll
            for (int kt = 0; kt < 2; kt++) {
                const int k = (2 * w + kt) * 32 + c;
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float *px = &X0[k * LDX + mt * 32 + acc_row(r, half)];
                        *px = *px > 0.0f ? acc[kt * 2 + mt][r] : 0.0f;
                    }
            }
        }
        __syncthreads();
        // ---- fc1 backward: thread t owns hidden unit k = t: db1, dW1[k][:] += dh1[m][k] s[m][:]
        {
            const float *row = &X0[t * LDX];
            for (int m = 0; m < TM; m++) {
                const float g = row[m];
                db1 += g;
#pragma unroll
                for (int ci = 0; ci < IN; ci++) dw1[ci] = fmaf(g, S[ci * LDX + m], dw1[ci]);
            }
        }
        __syncthreads();
    }
    // ---- this workgroup's partial gradient
    float *out = partials + (size_t)blockIdx.x * P_STRIDE;
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
        for (int jt = 0; jt < 8; jt++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                out[OFF_W2 + ((2 * w + it) * 32 + acc_row(r, half)) * H + jt * 32 + c] = dw2[it * 8 + jt][r];
#pragma unroll
    for (int ci = 0; ci < IN; ci++) out[OFF_W1 + t * IN + ci] = dw1[ci];
#pragma unroll
    for (int a = 0; a < NA; a++) out[OFF_W3 + a * H + t] = dw3[a];
    out[OFF_B1 + t] = db1;
    out[OFF_B2 + t] = db2;
    if (t < NA) out[OFF_B3 + t] = db3;
    RED[t] = loss;
    __syncthreads();
    if (t == 0) {
        float s = 0.0f;
        for (int q = 0; q < TM; q++) s += RED[q];
        out[OFF_LOSS] = s;
    }
}

// DQNAgent.choose_action (Training_DQN_pytorch.py:138-149) for a whole batch of observations in ONE launch: the same tiled forward
// as the learn step, argmax over the eight Q values (first maximum, like torch.argmax), and the epsilon-greedy draw from a
// counter-based generator keyed by (seed, row, call counter).
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__global__ __launch_bounds__(NT, 1) void k_dqn_act(Net ev, const float *obs, int n, float epsilon, uint64_t seed, uint32_t call,
                                                   int32_t *actions, float *qvalues) {
    __shared__ float X0[H * LDX], X1[H * LDX], S[12 * LDX], Q[NA * TM];
    const int t = threadIdx.x, w = t >> 6, l = t & 63, half = l >> 5, c = l & 31;
    const int ntiles = n / TM;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int e = t; e < TM * IN; e += NT) {
            const int m = e / IN, ci = e - m * IN;
            S[ci * LDX + m] = obs[(size_t)(tile * TM + m) * IN + ci];
        }
        if (t < TM) S[11 * LDX + t] = 0.0f;
        __syncthreads();
        forward_tile(ev, S, X0, X1, Q, w, half, c, l);
        if (t < TM) {
            const int row = tile * TM + t;
            int best = 0;
            float mx = Q[t];
#pragma unroll
            for (int a = 1; a < NA; a++) { const float q = Q[a * TM + t]; if (q > mx) { mx = q; best = a; } }
            if (epsilon > 0.0f) {
                uint32_t r[4] = { (uint32_t)row, call, 0x0AC7u, 0u };
                philox4x32_10(r, (uint32_t)seed, (uint32_t)(seed >> 32));
                if ((float)r[0] * 2.3283064365386963e-10f <= epsilon) best = (int)(r[1] & 7u); // np.random.random() <= epsilon: explore
            }
            actions[row] = best;
            if (qvalues) {
#pragma unroll
                for (int a = 0; a < NA; a++) qvalues[(size_t)row * NA + a] = Q[a * TM + t];
            }
        }
        __syncthreads();
    }
}

// DQNAgent.store_transition (Training_DQN_pytorch.py:126-136) for N transitions at once: the rows whose `valid` byte is set are
// appended to the ring in row order.  Stable compaction in two launches: k_dqn_store_scan (one workgroup) counts the valid rows of
// every 64-row chunk and scans the counts; k_dqn_store_write gives a chunk to a wavefront -- a row's rank inside its chunk is a
// ballot + popcount -- and every lane writes its own row.
constexpr int STORE_THREADS = 1024;
__global__ __launch_bounds__(STORE_THREADS) void k_dqn_store_scan(const uint8_t *valid, int n, uint32_t *chunk_off, int32_t *count_out) {
    __shared__ uint32_t scan[STORE_THREADS];
    const int t = threadIdx.x;
    const int nchunks = (n + 63) / 64;
    const int per = (nchunks + STORE_THREADS - 1) / STORE_THREADS;
    const int lo = t * per < nchunks ? t * per : nchunks, hi = lo + per < nchunks ? lo + per : nchunks;
    uint32_t cnt = 0;
    for (int ch = lo; ch < hi; ch++) {
        uint32_t c = 0;
        const int r0 = ch * 64, r1 = r0 + 64 < n ? r0 + 64 : n;
        if (valid) { for (int i = r0; i < r1; i++) c += valid[i] ? 1u : 0u; } else c = (uint32_t)(r1 - r0);
        chunk_off[ch] = c; // (count for now; turned into the exclusive prefix below)
        cnt += c;
    }
    scan[t] = cnt;
    __syncthreads();
    for (int off = 1; off < STORE_THREADS; off <<= 1) {
        const uint32_t add = t >= off ? scan[t - off] : 0u;
        __syncthreads();
        scan[t] += add;
        __syncthreads();
    }
    uint32_t run = scan[t] - cnt;
    for (int ch = lo; ch < hi; ch++) { const uint32_t c = chunk_off[ch]; chunk_off[ch] = run; run += c; }
    if (t == STORE_THREADS - 1 && count_out) *count_out = (int32_t)scan[t];
}
__global__ __launch_bounds__(256) void k_dqn_store_write(const float *s, const int32_t *a, const float *r, const float *s2, const uint8_t *done,
                                                         const uint8_t *valid, int n, const uint32_t *chunk_off, int64_t mem_cntr,
                                                         int64_t mem_size, float *sm, float *nsm, int64_t *am, float *rm, uint8_t *tm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // blockDim is a multiple of 64 and chunks are 64 rows: a wavefront = a chunk
    const bool ok = i < n && (!valid || valid[i]);
    const uint64_t m = __ballot(ok);
    if (!ok) return;
    const int lane = threadIdx.x & 63;
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const int64_t q = (mem_cntr + (int64_t)chunk_off[i >> 6] + (int64_t)rank) % mem_size;
#pragma unroll
    for (int k = 0; k < IN; k++) { sm[q * IN + k] = s[(size_t)i * IN + k]; nsm[q * IN + k] = s2[(size_t)i * IN + k]; }
    am[q] = (int64_t)a[i];
    rm[q] = r[i];
    tm[q] = done[i] ? 1 : 0;
}

struct Params { float *w1, *b1, *w2, *b2, *w3, *b3; };
__device__ __forceinline__ float *param_at(const Params &p, int i) {
    if (i < OFF_W1) return p.w2 + (i - OFF_W2);
    if (i < OFF_W3) return p.w1 + (i - OFF_W1);
    if (i < OFF_B1) return p.w3 + (i - OFF_W3);
    if (i < OFF_B2) return p.b1 + (i - OFF_B1);
    if (i < OFF_B3) return p.b2 + (i - OFF_B2);
    return p.b3 + (i - OFF_B3);
}
// Sums the partial gradients in workgroup order and (apply != 0) makes torch.optim.Adam's step (no weight decay, no amsgrad):
// m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).  grads_out (nullable)
// receives the reduced gradient in the partial layout (the parity test's window).
__global__ void k_dqn_reduce_adam(const float *partials, int nwg, Params p, float *m1, float *m2, float lr, float beta1, float beta2,
                                  float eps, float bc1, float sqrt_bc2, int apply, float *grads_out, float *loss_out, int batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > P_COUNT) return;
    // (the per-workgroup partials are fp32 sums over 128 samples; their sum is taken in fp64 and rounded once.  Eight independent
    // chains: a thread's 256 loads are 284 KB apart -- one dependent chain would pay the memory latency 256 times)
    double acc8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    int wg = 0;
    for (; wg + 8 <= nwg; wg += 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) acc8[q] += (double)partials[(size_t)(wg + q) * P_STRIDE + i];
    }
    for (; wg < nwg; wg++) acc8[0] += (double)partials[(size_t)wg * P_STRIDE + i];
    const double gd = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
    const float g = (float)gd;
    if (i == P_COUNT) { if (loss_out) *loss_out = (float)(gd / (double)batch); return; }
    if (grads_out) grads_out[i] = g;
    if (!apply) return;
    const float mm = beta1 * m1[i] + (1.0f - beta1) * g;
    const float vv = beta2 * m2[i] + (1.0f - beta2) * g * g;
    m1[i] = mm; m2[i] = vv;
    float *q = param_at(p, i);
    const float denom = sqrtf(vv) / sqrt_bc2 + eps;
    *q = *q - (lr / bc1) * (mm / denom);
}

thread_local std::string g_dqn_err;
int dfail(int code, const char *msg) { g_dqn_err = msg; return code; }

} // namespace

struct rr_dqn {
    int device, nwg;
    float *partials, *m1, *m2;
    long long step;
    uint32_t *chunk_off; // rr_dqn_store's per-chunk offsets (grown on demand; a call with a larger batch than any before allocates)
    size_t chunk_cap;
};

extern "C" {

const char *rr_dqn_last_error(void) { return g_dqn_err.c_str(); }

int rr_dqn_create(int32_t device, rr_dqn **out) {
    if (!out) return dfail(-1, "rr_dqn_create: null argument");
    int ndev = 0, prev = -1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return dfail(-1, "rr_dqn_create: no such HIP device");
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    // RR_DQN_WGS: workgroups of the update / act kernels (default: one per CU).  A workgroup takes a whole CU (every register, 143 KB of
    // LDS), so fewer of them leave CUs to whatever runs on another stream -- config 5 steps the simulator there (roborugby_amd/dqn.py)
    { const char *w = getenv("RR_DQN_WGS"); const int k = w ? atoi(w) : 0; if (k > 0 && k < cus) cus = k; }
    rr_dqn *d = new rr_dqn();
    d->device = device; d->nwg = cus; d->step = 0; d->partials = nullptr; d->m1 = nullptr; d->m2 = nullptr;
    d->chunk_off = nullptr; d->chunk_cap = 0;
    hipError_t e = hipMalloc((void **)&d->partials, sizeof(float) * (size_t)P_STRIDE * (size_t)d->nwg);
    if (e == hipSuccess) e = hipMalloc((void **)&d->m1, sizeof(float) * P_STRIDE);
    if (e == hipSuccess) e = hipMalloc((void **)&d->m2, sizeof(float) * P_STRIDE);
    if (e == hipSuccess) e = hipMemset(d->m1, 0, sizeof(float) * P_STRIDE);
    if (e == hipSuccess) e = hipMemset(d->m2, 0, sizeof(float) * P_STRIDE);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        if (d->partials) (void)hipFree(d->partials);
        if (d->m1) (void)hipFree(d->m1);
        if (d->m2) (void)hipFree(d->m2);
        delete d;
        return dfail(-3, "rr_dqn_create: out of device memory");
    }
    *out = d;
    return 0;
}

int rr_dqn_destroy(rr_dqn *d) {
    if (!d) return 0;
    (void)hipFree(d->partials); (void)hipFree(d->m1); (void)hipFree(d->m2);
    if (d->chunk_off) (void)hipFree(d->chunk_off);
    delete d;
    return 0;
}

int32_t rr_dqn_param_count(void) { return P_COUNT; }

static int dqn_run(rr_dqn *d, const rr_dqn_args *a, int apply, float *grads_out, void *stream) {
    if (!d || !a) return dfail(-1, "rr_dqn_update: null argument");
    if (a->struct_size != (int32_t)sizeof(rr_dqn_args)) return dfail(-1, "rr_dqn_update: rr_dqn_args.struct_size mismatch");
    for (int k = 0; k < 6; k++)
        if (!a->eval_params[k] || !a->target_params[k]) return dfail(-1, "rr_dqn_update: null parameter pointer");
    if (!a->state_memory || !a->new_state_memory || !a->action_memory || !a->reward_memory || !a->terminal_memory)
        return dfail(-1, "rr_dqn_update: null replay pointer");
    if (a->batch <= 0 || a->batch % TM) return dfail(-1, "rr_dqn_update: batch must be a positive multiple of 64");
    int prev = -1;
    const bool sw = hipGetDevice(&prev) == hipSuccess && prev != d->device && hipSetDevice(d->device) == hipSuccess;
    Net ev = { a->eval_params[0], a->eval_params[1], a->eval_params[2], a->eval_params[3], a->eval_params[4], a->eval_params[5] };
    Net tg = { a->target_params[0], a->target_params[1], a->target_params[2], a->target_params[3], a->target_params[4], a->target_params[5] };
    Batch bt = { a->state_memory, a->new_state_memory, a->reward_memory, a->action_memory, a->terminal_memory, a->batch_index, a->batch, a->gamma };
    const int ntiles = a->batch / TM;
    const int nwg = ntiles < d->nwg ? ntiles : d->nwg;
    hipLaunchKernelGGL(k_dqn_fwd_bwd, dim3(nwg), dim3(NT), 0, (hipStream_t)stream, ev, tg, bt, d->partials);
    float bc1 = 1.0f, sbc2 = 1.0f;
    if (apply) {
        d->step += 1;
        bc1 = (float)(1.0 - std::pow((double)a->beta1, (double)d->step));
        sbc2 = (float)std::sqrt(1.0 - std::pow((double)a->beta2, (double)d->step));
    }
    Params p = { const_cast<float *>(a->eval_params[0]), const_cast<float *>(a->eval_params[1]), const_cast<float *>(a->eval_params[2]),
                 const_cast<float *>(a->eval_params[3]), const_cast<float *>(a->eval_params[4]), const_cast<float *>(a->eval_params[5]) };
    hipLaunchKernelGGL(k_dqn_reduce_adam, dim3((P_COUNT + 1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float *)d->partials, nwg, p,
                       d->m1, d->m2, a->lr, a->beta1, a->beta2, a->eps, bc1, sbc2, apply, grads_out, a->loss_out, a->batch);
    const hipError_t e = hipGetLastError();
    if (sw) (void)hipSetDevice(prev);
    if (e != hipSuccess) return dfail(-2, hipGetErrorString(e));
    return 0;
}

int rr_dqn_update(rr_dqn *d, const rr_dqn_args *a, void *stream) { return dqn_run(d, a, 1, nullptr, stream); }
int rr_dqn_grads(rr_dqn *d, const rr_dqn_args *a, float *grads, void *stream) {
    if (!grads) return dfail(-1, "rr_dqn_grads: null output");
    return dqn_run(d, a, 0, grads, stream);
}
int rr_dqn_act(rr_dqn *d, const float *const params[6], const float *obs, int32_t n, float epsilon, uint64_t seed, uint32_t call,
               int32_t *actions, float *qvalues, void *stream) {
    if (!d || !params || !obs || !actions) return dfail(-1, "rr_dqn_act: null argument");
    for (int k = 0; k < 6; k++) if (!params[k]) return dfail(-1, "rr_dqn_act: null parameter pointer");
    if (n <= 0 || n % TM) return dfail(-1, "rr_dqn_act: the number of observations must be a positive multiple of 64");
    if (!(epsilon >= 0.0f && epsilon <= 1.0f)) return dfail(-1, "rr_dqn_act: epsilon must be in [0, 1]");
    int prev = -1;
    const bool sw = hipGetDevice(&prev) == hipSuccess && prev != d->device && hipSetDevice(d->device) == hipSuccess;
    Net ev = { params[0], params[1], params[2], params[3], params[4], params[5] };
    const int ntiles = n / TM, nwg = ntiles < d->nwg ? ntiles : d->nwg;
    hipLaunchKernelGGL(k_dqn_act, dim3(nwg), dim3(NT), 0, (hipStream_t)stream, ev, obs, (int)n, epsilon, seed, call, actions, qvalues);
    const hipError_t e = hipGetLastError();
    if (sw) (void)hipSetDevice(prev);
    if (e != hipSuccess) return dfail(-2, hipGetErrorString(e));
    return 0;
}
int rr_dqn_store(rr_dqn *d, const float *state, const int32_t *action, const float *reward, const float *new_state, const uint8_t *done,
                 const uint8_t *valid, int32_t n, int64_t mem_cntr, int64_t mem_size, float *state_memory, float *new_state_memory,
                 int64_t *action_memory, float *reward_memory, uint8_t *terminal_memory, int32_t *count_out, void *stream) {
    if (!d || !state || !action || !reward || !new_state || !done || !state_memory || !new_state_memory || !action_memory || !reward_memory ||
        !terminal_memory)
        return dfail(-1, "rr_dqn_store: null argument");
    if (n <= 0 || mem_size <= 0 || mem_cntr < 0 || n > mem_size) return dfail(-1, "rr_dqn_store: bad sizes");
    int prev = -1;
    const bool sw = hipGetDevice(&prev) == hipSuccess && prev != d->device && hipSetDevice(d->device) == hipSuccess;
    const size_t need = ((size_t)n + 63) / 64;
    if (need > d->chunk_cap) {
        if (d->chunk_off) (void)hipFree(d->chunk_off);
        d->chunk_off = nullptr; d->chunk_cap = 0;
        if (hipMalloc((void **)&d->chunk_off, sizeof(uint32_t) * need) != hipSuccess) { if (sw) (void)hipSetDevice(prev); return dfail(-3, "rr_dqn_store: out of device memory"); }
        d->chunk_cap = need;
    }
    hipLaunchKernelGGL(k_dqn_store_scan, dim3(1), dim3(STORE_THREADS), 0, (hipStream_t)stream, valid, (int)n, d->chunk_off, count_out);
    hipLaunchKernelGGL(k_dqn_store_write, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, state, action, reward, new_state, done, valid,
                       (int)n, (const uint32_t *)d->chunk_off, mem_cntr, mem_size, state_memory, new_state_memory, action_memory, reward_memory,
                       terminal_memory);
    const hipError_t e = hipGetLastError();
    if (sw) (void)hipSetDevice(prev);
    if (e != hipSuccess) return dfail(-2, hipGetErrorString(e));
    return 0;
}
int rr_dqn_adam_state(rr_dqn *d, float *exp_avg, float *exp_avg_sq, int64_t *step, int32_t set, void *stream) {
    if (!d || !exp_avg || !exp_avg_sq || !step) return dfail(-1, "rr_dqn_adam_state: null argument");
    const size_t bytes = sizeof(float) * P_COUNT;
    hipError_t e;
    if (set) {
        e = hipMemcpyAsync(d->m1, exp_avg, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d->m2, exp_avg_sq, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        d->step = *step;
    } else {
        e = hipMemcpyAsync(exp_avg, d->m1, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e == hipSuccess) e = hipMemcpyAsync(exp_avg_sq, d->m2, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        *step = d->step;
    }
    return e == hipSuccess ? 0 : dfail(-2, hipGetErrorString(e));
}

} // extern "C"
