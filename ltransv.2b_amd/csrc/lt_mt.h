// lt_mt.h -- the reference's single serial Mersenne-Twister stream
// (random_module.f90:86-197, MT19937 of Matsumoto & Nishimura) produced in
// parallel on the device, bit-identical to sequential generation.
//
// The reference draws every random number of a run from ONE generator in
// particle order.  Here the same stream is MATERIALISED in HBM in blocks of
// S raw words: block b holds stream words [b*S, (b+1)*S).  Block b belongs to
// generator state b mod P, positioned at the start of the block after the one it
// wrote last; a block nobody asks for is never generated.  The owner of a needed
// block first jumps ahead by a whole number of blocks with the polynomial method
// of Haramoto, Matsumoto, Nishimura, Panneton & L'Ecuyer (2008) (a single engine:
// by P-1 blocks after every block written; a shard of a multi-GPU run: k*P-1
// blocks, over the other shards' part of the stream
// instead of generating it): F^J s = g(F) s with g(x) = x^J mod phi(x), phi =
// characteristic polynomial of the MT19937 recurrence (degree 19937), evaluated
// as  XOR_{i: g_i = 1} window_i(E)  over the sequence E generated from s.
//
// State representation ("history window"): h[0..623] = the 624 most recent raw
// words; the next raw word is  h[397] ^ twist(h[0], h[1]).  Only the top bit of
// h[0] is ever used again, so the 31 low bits of h[0] (the part of the state the
// polynomial identity does not preserve) are irrelevant to every later output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace ltmt {

constexpr int N = 624, M = 397;
constexpr int DEG = 19937;
constexpr int GW = 624;  // uint32 words that hold a polynomial of degree < 19968

__host__ __device__ inline uint32_t twist(uint32_t a, uint32_t b) {
  uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__host__ __device__ inline uint32_t temper(uint32_t y) {  // random_module.f90:190-193
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// ------------------------------------------------------------------ host side
// GF(2) polynomials as little-endian bit arrays in uint64 words.
using Poly = std::vector<uint64_t>;
inline bool pbit(const Poly& p, int i) { return (p[i >> 6] >> (i & 63)) & 1ull; }
inline void pflip(Poly& p, int i) { p[i >> 6] ^= 1ull << (i & 63); }

// advance a history window by one word (host reference of the device code)
inline uint32_t host_next(uint32_t* h, int& p) {  // circular, p = index of h[0]
  uint32_t nw = h[(p + M) % N] ^ twist(h[p], h[(p + 1) % N]);
  h[p] = nw;
  p = (p + 1) % N;
  return nw;
}

// characteristic polynomial of the recurrence by Berlekamp-Massey on one output
// bit of 2*DEG+2 consecutive words; phi[i] = coefficient of x^i, degree DEG
inline Poly characteristic_polynomial() {
  const int LEN = 2 * DEG + 64;
  std::vector<uint32_t> h(N);
  h[0] = 5489u;
  for (int i = 1; i < N; i++) h[i] = 1812433253u * (h[i - 1] ^ (h[i - 1] >> 30)) + (uint32_t)i;
  int p = 0;
  std::vector<uint8_t> s(LEN);
  for (int i = 0; i < LEN; i++) s[i] = (uint8_t)(host_next(h.data(), p) & 1u);
  const int W = (DEG + 64 + 63) / 64 + 1;
  Poly C(W, 0), B(W, 0), T(W, 0), srev(W, 0);
  C[0] = 1; B[0] = 1;
  int L = 0, m = 1;
  for (int n = 0; n < LEN; n++) {
    // srev bit j = s[n-j]
    uint64_t carry = s[n];
    for (int w = 0; w < W; w++) {
      uint64_t nc = srev[w] >> 63;
      srev[w] = (srev[w] << 1) | carry;
      carry = nc;
    }
    uint64_t acc = 0;
    const int lw = (L >> 6) + 1;
    for (int w = 0; w < lw && w < W; w++) acc ^= C[w] & srev[w];
    int d = __builtin_parityll(acc);
    if (!d) { m++; continue; }
    auto xor_shift = [&](Poly& dst, const Poly& src, int sh) {  // dst ^= src << sh
      int ws = sh >> 6, bs = sh & 63;
      for (int w = W - 1; w >= ws; w--) {
        uint64_t v = src[w - ws] << bs;
        if (bs && w - ws - 1 >= 0) v |= src[w - ws - 1] >> (64 - bs);
        dst[w] ^= v;
      }
    };
    if (2 * L <= n) {
      T = C;
      xor_shift(C, B, m);
      L = n + 1 - L;
      B = T;
      m = 1;
    } else {
      xor_shift(C, B, m);
      m++;
    }
  }
  // connection polynomial C (degree L) -> characteristic polynomial x^L C(1/x)
  Poly phi((DEG + 64) / 64 + 1, 0);
  if (L != DEG) return Poly();  // cannot happen for MT19937
  for (int i = 0; i <= L; i++)
    if (pbit(C, i)) pflip(phi, L - i);
  return phi;
}

// g(x) = x^e mod phi(x) as GW uint32 words (bit i = coefficient of x^i)
inline std::vector<uint32_t> jump_polynomial(const Poly& phi, uint64_t e) {
  const int W2 = (2 * DEG + 64) / 64 + 2;
  Poly r(W2, 0), t(W2, 0);
  r[0] = 1;  // 1
  auto reduce = [&](Poly& a, int top) {  // a mod phi, a has bits < top
    for (int i = top - 1; i >= DEG; i--) {
      if (!pbit(a, i)) continue;
      int sh = i - DEG, ws = sh >> 6, bs = sh & 63;
      for (size_t w = 0; w < phi.size() && w + ws < a.size(); w++) {
        a[w + ws] ^= phi[w] << bs;
        if (bs && w + ws + 1 < a.size()) a[w + ws + 1] ^= phi[w] >> (64 - bs);
      }
    }
  };
  int hb = 63;
  while (hb > 0 && !((e >> hb) & 1ull)) hb--;
  if (e == 0) hb = -1;
  for (int b = hb; b >= 0; b--) {
    // square: spread bits
    std::fill(t.begin(), t.end(), 0);
    for (int i = 0; i < DEG; i++)
      if (pbit(r, i)) pflip(t, 2 * i);
    reduce(t, 2 * DEG);
    if ((e >> b) & 1ull) {  // multiply by x
      uint64_t carry = 0;
      for (size_t w = 0; w < t.size(); w++) {
        uint64_t nc = t[w] >> 63;
        t[w] = (t[w] << 1) | carry;
        carry = nc;
      }
      reduce(t, DEG + 1);
    }
    r = t;
  }
  std::vector<uint32_t> g(GW, 0);
  for (int i = 0; i < DEG; i++)
    if (pbit(r, i)) g[i >> 5] |= 1u << (i & 31);
  return g;
}

// a(x) * b(x) mod phi(x), polynomials as GW uint32 words (bit i = coefficient of x^i, degree < DEG):
// the jump over m + n words is the product of the jumps over m and over n words
inline std::vector<uint32_t> poly_mulmod(const Poly& phi, const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
  const int W2 = (2 * DEG + 64) / 64 + 2;
  Poly bb(W2, 0), r(W2, 0);
  for (int i = 0; i < GW; i++) bb[i >> 1] |= (uint64_t)b[i] << (32 * (i & 1));
  const int bw = (DEG + 63) / 64;
  for (int i = 0; i < DEG; i++) {
    if (!((a[i >> 5] >> (i & 31)) & 1u)) continue;
    const int ws = i >> 6, bs = i & 63;
    for (int w = 0; w < bw; w++) {
      r[w + ws] ^= bb[w] << bs;
      if (bs) r[w + ws + 1] ^= bb[w] >> (64 - bs);
    }
  }
  for (int i = 2 * DEG - 1; i >= DEG; i--) {  // mod phi
    if (!pbit(r, i)) continue;
    const int sh = i - DEG, ws = sh >> 6, bs = sh & 63;
    for (size_t w = 0; w < phi.size() && w + ws < r.size(); w++) {
      r[w + ws] ^= phi[w] << bs;
      if (bs && w + ws + 1 < r.size()) r[w + ws + 1] ^= phi[w] >> (64 - bs);
    }
  }
  std::vector<uint32_t> g(GW, 0);
  for (int i = 0; i < DEG; i++)
    if (pbit(r, i)) g[i >> 5] |= 1u << (i & 31);
  return g;
}

// host reference of the jump: out = g(F) h  (linear windows, h[0] oldest)
inline void host_jump(const uint32_t* h, const uint32_t* g, uint32_t* out) {
  std::vector<uint32_t> E(DEG + N);
  for (int i = 0; i < N; i++) E[i] = h[i];
  for (int i = 0; i + N < DEG + N; i++) E[i + N] = E[i + M] ^ twist(E[i], E[i + 1]);
  for (int j = 0; j < N; j++) out[j] = 0;
  for (int i = 0; i < DEG; i++)
    if ((g[i >> 5] >> (i & 31)) & 1u)
      for (int j = 0; j < N; j++) out[j] ^= E[i + j];
}

// ---------------------------------------------------------------- device side
#ifndef LT_GEN_THREADS
#define LT_GEN_THREADS 256
#endif
#ifndef LT_GEN_PER_WG
#define LT_GEN_PER_WG 4
#endif
#ifndef LT_GEN_ROUNDS
#define LT_GEN_ROUNDS 16
#endif
constexpr int CHUNK = 227 * LT_GEN_ROUNDS;      // words generated per LDS pass (227 = N - M independent words)
constexpr int JUMP_CHUNK = ((CHUNK / 32) < 112 ? (CHUNK / 32) : 112) * 32;  // coefficients of the jump polynomial handled per LDS pass (whole words, <= CHUNK)
constexpr int GEN_THREADS = LT_GEN_THREADS;      // threads that drive one generator state
constexpr int GEN_PER_WG = LT_GEN_PER_WG;        // generator states per workgroup of k_mt_block
constexpr int GEN_LDS_WORDS = N + CHUNK;
constexpr int GEN_NQ = (N + GEN_THREADS - 1) / GEN_THREADS;   // words of the 624-word window a thread owns

// window (lds[0..N)) <- g(F) window, for the GEN_THREADS threads t of one generator.  The sequence E generated
// from the window is produced JUMP_CHUNK words at a time behind a sliding window (N + JUMP_CHUNK words of LDS
// instead of DEG + N): new window[j] = XOR_{i: g_i = 1} E[i + j]; thread t owns j = t, t+256, t+512.
// Every thread of the workgroup calls it (barriers); `active` = this generator really jumps.
// (T = the threads that drive the generator: GEN_THREADS in k_mt_block, 256 in the start-up kernels; thread t owns
// j = t, t + T, ...)
template <int T>
__device__ inline void jump_in_lds(uint32_t* lds, const uint32_t* __restrict__ g, int t, bool active) {
  constexpr int NQ = (N + T - 1) / T;
  uint32_t acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0u;
  for (int c = 0; c < DEG; c += JUMP_CHUNK) {
    const int todo = (DEG - c < JUMP_CHUNK) ? (DEG - c) : JUMP_CHUNK;
    for (int base = 0; base < todo; base += (N - M)) {
#pragma unroll
      for (int r = 0; r < (N - M + T - 1) / T; r++) {
        const int k = t + r * T, i = base + k;
        if (active && k < (N - M) && i < todo) lds[N + i] = lds[M + i] ^ twist(lds[i], lds[i + 1]);
      }
      __syncthreads();
    }
    if (active) {
      const int w1 = (c + todo + 31) >> 5;  // (the bits of the last word beyond DEG - 1 are zero: degree < DEG)
      for (int w = c >> 5; w < w1; w++) {
        uint32_t bits = g[w];
        const int i0 = w * 32 - c;
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1;
          const int i = i0 + b;
#pragma unroll
          for (int q = 0; q < NQ; q++)
            if (t + q * T < N) acc[q] ^= lds[i + t + q * T];
        }
      }
    }
    __syncthreads();
    // slide the window by todo words
    uint32_t keep[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int i = t + q * T;
      keep[q] = (active && i < N) ? lds[todo + i] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int i = t + q * T;
      if (active && i < N) lds[i] = keep[q];
    }
    __syncthreads();
  }
  if (active) {
#pragma unroll
    for (int q = 0; q < NQ; q++)
      if (t + q * T < N) lds[t + q * T] = acc[q];
  }
  __syncthreads();
}

// GEN_THREADS threads per job, GEN_PER_WG jobs per workgroup (the 16 waves of a workgroup fill the wave slots of
// ONE compute unit with four generators, where a workgroup per generator spread them over four times as many
// compute units, each of which then lost a wave slot per SIMD -- half of k_rk's occupancy, which runs beside
// this kernel -- to a generator's wave).  Generator state `state` is positioned at the start of some stream
// block; the job jumps it ahead with polynomial `poly` (slot in `polys`, -1: it already sits at block `blk`),
// writes the `count` raw words of stream block `blk` to the block's ring slot (store = 0: only advances the
// state) and leaves the state positioned at block blk + 1.
// Stream block 0 starts with the 624 words of the state itself (= the words already in the reference's mt
// array, "virtual stream" V[0..623]); the host launches that job alone (its loop counts differ).
struct BlockJob {
  long long blk;
  int state, poly, store, pad_;
};
__global__ __launch_bounds__(GEN_THREADS * GEN_PER_WG) void k_mt_block(uint32_t* __restrict__ states,
                                                                       uint32_t* __restrict__ ring,
                                                                       const BlockJob* __restrict__ jobs, int njobs,
                                                                       int ring_blocks,
                                                                       const uint32_t* __restrict__ polys, int count) {
  extern __shared__ uint32_t lds_all[];  // GEN_PER_WG x GEN_LDS_WORDS
  const int sub = threadIdx.x / GEN_THREADS, tid = threadIdx.x % GEN_THREADS;
  uint32_t* lds = lds_all + (size_t)sub * GEN_LDS_WORDS;
  const int jn = blockIdx.x * GEN_PER_WG + sub;
  const bool active = jn < njobs;
  BlockJob job;
  job.blk = 1; job.state = 0; job.poly = -1; job.store = 0; job.pad_ = 0;
  if (active) job = jobs[jn];
  const long long blk = job.blk;
  uint32_t* st = states + (size_t)job.state * N;
  uint32_t* out = ring + (size_t)(blk % ring_blocks) * (size_t)count;
  // (workgroup-uniform: block 0 comes alone)
  const int first_copy = __syncthreads_or(active && blk == 0) ? N : 0;
  const bool store = active && job.store != 0;
  if (active)
    for (int i = tid; i < N; i += GEN_THREADS) lds[i] = st[i];
  __syncthreads();
  if (__syncthreads_or(active && job.poly >= 0))
    jump_in_lds<GEN_THREADS>(lds, polys + (size_t)(job.poly >= 0 ? job.poly : 0) * GW, tid, active && job.poly >= 0);
  int done = 0;
  if (first_copy > 0) {
    if (store)
      for (int i = tid; i < first_copy; i += GEN_THREADS) out[i] = lds[N - first_copy + i];
    done = first_copy;
  }
  while (done < count) {
    const int todo = (count - done < CHUNK) ? (count - done) : CHUNK;
    // rounds of up to 227 independent words: E[N+i] = E[M+i] ^ twist(E[i], E[i+1])
    for (int base = 0; base < todo; base += (N - M)) {
#pragma unroll
      for (int r = 0; r < (N - M + GEN_THREADS - 1) / GEN_THREADS; r++) {
        const int k = tid + r * GEN_THREADS, i = base + k;
        if (active && k < (N - M) && i < todo) lds[N + i] = lds[M + i] ^ twist(lds[i], lds[i + 1]);
      }
      __syncthreads();
    }
    if (store)
      for (int i = tid; i < todo; i += GEN_THREADS) out[done + i] = lds[N + i];
    __syncthreads();
    // slide the history window
    uint32_t keep[GEN_NQ];
#pragma unroll
    for (int q = 0; q < GEN_NQ; q++) {
      int i = tid + q * GEN_THREADS;
      keep[q] = (active && i < N) ? lds[todo + i] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < GEN_NQ; q++) {
      int i = tid + q * GEN_THREADS;
      if (active && i < N) lds[i] = keep[q];
    }
    __syncthreads();
    done += todo;
  }
  if (active)
    for (int i = tid; i < N; i += GEN_THREADS) st[i] = lds[i];
}

// states[dst] = g(F) states[src]   (positioning of the generator states at start-up)
__global__ __launch_bounds__(256) void k_mt_jump(uint32_t* __restrict__ states, int src, int dst,
                                                 const uint32_t* __restrict__ g) {
  extern __shared__ uint32_t lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < N; i += blockDim.x) lds[i] = states[(size_t)src * N + i];
  __syncthreads();
  jump_in_lds<256>(lds, g, tid, true);
  for (int i = tid; i < N; i += blockDim.x) states[(size_t)dst * N + i] = lds[i];
}

// jump_in_lds for a workgroup of JUMP_WIDE threads that drives ONE generator: thread t accumulates the single word
// j = t of the new window (t < N) instead of three, so the pass over the ~10^4 set bits of g reads a third as much
// per thread; the generation rounds use the first N - M threads as before.  Same result as jump_in_lds.
constexpr int JUMP_WIDE = 1024;
__device__ inline void jump_in_lds_wide(uint32_t* lds, const uint32_t* __restrict__ g, int t) {
  uint32_t acc = 0;
  for (int c = 0; c < DEG; c += JUMP_CHUNK) {
    const int todo = (DEG - c < JUMP_CHUNK) ? (DEG - c) : JUMP_CHUNK;
    for (int base = 0; base < todo; base += (N - M)) {
      const int i = base + t;
      if (t < (N - M) && i < todo) lds[N + i] = lds[M + i] ^ twist(lds[i], lds[i + 1]);
      __syncthreads();
    }
    if (t < N) {
      const int w1 = (c + todo + 31) >> 5;
      for (int w = c >> 5; w < w1; w++) {
        uint32_t bits = g[w];
        const int i0 = w * 32 - c;
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1;
          acc ^= lds[i0 + b + t];
        }
      }
    }
    __syncthreads();
    const uint32_t keep = (t < N) ? lds[todo + t] : 0u;   // slide the window by todo words
    __syncthreads();
    if (t < N) lds[t] = keep;
    __syncthreads();
  }
  if (t < N) lds[t] = acc;
  __syncthreads();
}

// states[dst0 + b] = g(F) states[src0 + b], b = blockIdx.x: one round of the start-up positioning by doubling (the
// states 1 .. 2^r sit at the starts of blocks 1 .. 2^r; the jump over 2^r blocks places 2^r more of them)
__global__ __launch_bounds__(JUMP_WIDE) void k_mt_jump_fan(uint32_t* __restrict__ states, int src0, int dst0,
                                                          const uint32_t* __restrict__ g) {
  extern __shared__ uint32_t lds[];
  const int tid = threadIdx.x;
  const uint32_t* src = states + (size_t)(src0 + (int)blockIdx.x) * N;
  uint32_t* dst = states + (size_t)(dst0 + (int)blockIdx.x) * N;
  if (tid < N) lds[tid] = src[tid];
  __syncthreads();
  jump_in_lds_wide(lds, g, tid);
  if (tid < N) dst[tid] = lds[tid];
}

// states[b] = g(F) states[b] for every state b = blockIdx.x (a shard moves its whole pool to its part of the stream)
__global__ __launch_bounds__(256) void k_mt_jump_each(uint32_t* __restrict__ states, const uint32_t* __restrict__ g) {
  extern __shared__ uint32_t lds[];
  const int tid = threadIdx.x;
  uint32_t* st = states + (size_t)blockIdx.x * N;
  for (int i = tid; i < N; i += blockDim.x) lds[i] = st[i];
  __syncthreads();
  jump_in_lds<256>(lds, g, tid, true);
  for (int i = tid; i < N; i += blockDim.x) st[i] = lds[i];
}

constexpr size_t LDS_BYTES = (size_t)GEN_LDS_WORDS * 4;  // per generator state

}  // namespace ltmt
