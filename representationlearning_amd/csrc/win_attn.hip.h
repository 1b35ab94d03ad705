// Shared pieces of the fused 7x7-window cross-attention kernels (forward and backward).
#pragma once
#include "common.hip.h"

namespace rssf {
namespace wa {

constexpr int WIN = 7;       // window side (multihead_isa_pool_attention.py: window_size=7)
constexpr int LP = 64;      // window tokens padded to 4 MFMA tiles (49 live)
constexpr int NT = LP / 16;
constexpr int MAX_MT = 4;   // virtual channel tiles: heads * ceil16(d) / 16  (Base 2, Tiny 2, Large 4)

template <typename T> struct Pad;           // LDS row padding (elements) that keeps fragment reads conflict-free
template <> struct Pad<bf16_t> { static constexpr int X = 8; };
template <> struct Pad<float> { static constexpr int X = 4; };

// compile-time tiling of one (C, heads) configuration
template <int C_, int HEADS_> struct Dims {
  static constexpr int C = C_, HEADS = HEADS_;
  static constexpr int D = C / HEADS;
  static constexpr int DP = (D + 15) / 16 * 16;      // head dim padded to MFMA tiles
  static constexpr int CP = (C + 15) / 16 * 16;      // input channels padded (K of the projections)
  static constexpr int CV = HEADS * DP;              // "virtual" channels: per-head padded layout
  static constexpr int MT = CV / 16, CT = CP / 16, TPH = DP / 16;
  static_assert(C % HEADS == 0, "embed_dim must be divisible by num_heads");
};

struct Geom {           // runtime geometry (image / window grid)
  int B, H, W, win, L, QH, QW, padT, padL, nWin, N;
  int xcd_major;      // forward launcher: workgroups numbered XCD-major (win_attn_fwd.hip)
};

inline Geom make_geom(int B, int H, int W, int win) {
  Geom g;
  g.B = B; g.H = H; g.W = W; g.win = win; g.L = win * win;
  const int ph = (win - H % win) % win, pw = (win - W % win) % win;
  g.QH = (H + ph) / win; g.QW = (W + pw) / win;
  g.padT = ph / 2; g.padL = pw / 2;          // center pad: floor(P/2) before (multihead_isa_attention.py:375-381)
  g.nWin = B * g.QH * g.QW;
  g.N = H * W;
  g.xcd_major = 0;
  return g;
}

// wave-local LDS ordering: LDS ops of one wave complete in order; this only stops the compiler from moving
// cross-lane-dependent LDS accesses across the point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// 4 consecutive elements to an 8-byte (bf16) / 16-byte (f32) aligned address
__device__ __forceinline__ void store4(bf16_t* p, const f32x4& v) { *reinterpret_cast<s16x4*>(p) = pack_bf16x4(v); }
__device__ __forceinline__ void store4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

// real channel of virtual channel m (= h*dp + dc), or -1 for a padding row
template <typename DM>
__device__ __forceinline__ int real_ch(int m) {
  const int h = m / DM::DP, dc = m % DM::DP;
  return (dc < DM::D && h < DM::HEADS) ? h * DM::D + dc : -1;
}

// token index (within the image) of window slot t, or -1 if the slot lies in the zero padding / beyond 49
__device__ __forceinline__ int slot_token(const Geom& g, int qh, int qw, int t) {
  // the window is always 7x7 (checked by the entry points): compile-time divisors, no integer division per token
  if (t >= WIN * WIN) return -1;
  const int u = qh * WIN + t / WIN - g.padT;
  const int v = qw * WIN + t % WIN - g.padL;
  return (u >= 0 && u < g.H && v >= 0 && v < g.W) ? u * g.W + v : -1;
}

// fragment from 4 accumulator values (k-slots (l>>4)*4 + r): chaining C-layout registers into the next MFMA
template <typename T> struct Chain;
template <> struct Chain<bf16_t> {
  static constexpr int STEPS = 1;
  static __device__ __forceinline__ s16x4 frag(const f32x4& v, int) { return pack_bf16x4(v); }
};
template <> struct Chain<float> {
  static constexpr int STEPS = 4;
  static __device__ __forceinline__ float frag(const f32x4& v, int r) { return v[r]; }
};

// Packed form of a C-layout tile that is only ever used again as an MFMA operand (halves the registers in bf16).
template <typename T> struct Packed;
template <> struct Packed<bf16_t> {
  typedef s16x4 type;
  static __device__ __forceinline__ type pack(const f32x4& v) { return pack_bf16x4(v); }
  static __device__ __forceinline__ f32x4 mma(type a, type b, f32x4 acc) { return Mma<bf16_t>::mma(a, b, acc); }
  static __device__ __forceinline__ f32x4 mma_lds_a(const bf16_t* A, int lda, int k0, type b, f32x4 acc) {
    const int lane = threadIdx.x & 63;
    return Mma<bf16_t>::mma(Mma<bf16_t>::load(A + (lane & 15) * lda + k0 + (lane >> 4) * 4), b, acc);
  }
};
template <> struct Packed<float> {
  typedef f32x4 type;
  static __device__ __forceinline__ type pack(const f32x4& v) { return v; }
  static __device__ __forceinline__ f32x4 mma(const type& a, const type& b, f32x4 acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = Mma<float>::mma(a[r], b[r], acc);
    return acc;
  }
  static __device__ __forceinline__ f32x4 mma_lds_a(const float* A, int lda, int k0, const type& b, f32x4 acc) {
    const int lane = threadIdx.x & 63;
    const float* pa = A + (lane & 15) * lda + k0 + (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = Mma<float>::mma(pa[r], b[r], acc);
    return acc;
  }
};

// D += X * Y where both operands come from C-layout register tiles whose row axis (4g+r) is the K axis.
template <typename T>
__device__ __forceinline__ f32x4 mma_chain(const f32x4& a, const f32x4& b, f32x4 acc) {
#pragma unroll
  for (int s = 0; s < Chain<T>::STEPS; ++s) acc = Mma<T>::mma(Chain<T>::frag(a, s), Chain<T>::frag(b, s), acc);
  return acc;
}

// A operand from LDS (k-contiguous rows), B operand chained from a C-layout register tile covering 16 k-slots
// starting at k0 (slot order within the tile: 4g + r).
template <typename T>
__device__ __forceinline__ f32x4 mma_lds_chain(const T* A, int lda, int k0, const f32x4& b, f32x4 acc);
template <>
__device__ __forceinline__ f32x4 mma_lds_chain<bf16_t>(const bf16_t* A, int lda, int k0, const f32x4& b, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const bf16_t* pa = A + (lane & 15) * lda + k0 + (lane >> 4) * 4;
  return Mma<bf16_t>::mma(Mma<bf16_t>::load(pa), pack_bf16x4(b), acc);
}
template <>
__device__ __forceinline__ f32x4 mma_lds_chain<float>(const float* A, int lda, int k0, const f32x4& b, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const float* pa = A + (lane & 15) * lda + k0 + (lane >> 4) * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc = Mma<float>::mma(pa[r], b[r], acc);
  return acc;
}
// B operand from LDS, A operand chained from registers.
template <typename T>
__device__ __forceinline__ f32x4 mma_chain_lds(const f32x4& a, const T* Bm, int ldb, int k0, f32x4 acc);
template <>
__device__ __forceinline__ f32x4 mma_chain_lds<bf16_t>(const f32x4& a, const bf16_t* Bm, int ldb, int k0, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const bf16_t* pb = Bm + (lane & 15) * ldb + k0 + (lane >> 4) * 4;
  return Mma<bf16_t>::mma(pack_bf16x4(a), Mma<bf16_t>::load(pb), acc);
}
template <>
__device__ __forceinline__ f32x4 mma_chain_lds<float>(const f32x4& a, const float* Bm, int ldb, int k0, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const float* pb = Bm + (lane & 15) * ldb + k0 + (lane >> 4) * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc = Mma<float>::mma(a[r], pb[r], acc);
  return acc;
}

// Fragment whose K axis is the ROW axis of a row-major LDS tile buf[row][ld] (e.g. a token-major activation tile used
// with K = tokens, or a weight matrix used transposed): 16 columns c0.. (fragment row/col index = l15) x 16 rows k0..
// in chain slot order (slot 4g + r).  bf16: ONE ds_read_b64_tr_b16 (lanes 4j..4j+3 of a 16-lane group address the four
// 8-byte quarters of row k0 + 4g + j; lane i receives column i of that [4][16] sub-tile - measured semantics, see
// conv_wgrad.hip); f32: four plain reads (the 16x16x4 MFMA takes one K value per lane).  This replaces the second,
// channel-major copy of every operand the backward used to stage.
typedef __attribute__((ext_vector_type(4))) short v4s_t;
template <typename T> struct RowFrag;
template <> struct RowFrag<bf16_t> {
  static __device__ __forceinline__ s16x4 load(const bf16_t* buf, int ld, int k0, int c0) {
    const int lane = threadIdx.x & 63, grp = lane >> 4, i = lane & 15;
    const bf16_t* p = buf + (k0 + grp * 4 + (i >> 2)) * ld + c0 + (i & 3) * 4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(p));
  }
};
template <> struct RowFrag<float> {
  static __device__ __forceinline__ f32x4 load(const float* buf, int ld, int k0, int c0) {
    const int lane = threadIdx.x & 63;
    const float* p = buf + (k0 + (lane >> 4) * 4) * ld + c0 + (lane & 15);
    return f32x4{p[0], p[ld], p[2 * ld], p[3 * ld]};
  }
};
// operand fragment in CHAIN slot order (slot 4g + r) straight from a k-contiguous LDS tile: row l15, elements k0 + 4g .. + 3
template <typename T>
__device__ __forceinline__ typename Packed<T>::type chain_frag_lds(const T* buf, int ld, int k0) {
  const int lane = threadIdx.x & 63;
  return *reinterpret_cast<const typename Packed<T>::type*>(buf + (lane & 15) * ld + k0 + (lane >> 4) * 4);
}
// D += A^T-view * chain:  A operand = columns c0.. of buf with K = rows k0..k0+15, B operand chained from registers
template <typename T>
__device__ __forceinline__ f32x4 mma_row_chain(const T* buf, int ld, int c0, int k0, const f32x4& b, f32x4 acc) {
  return Packed<T>::mma(RowFrag<T>::load(buf, ld, k0, c0), Packed<T>::pack(b), acc);
}
// both operands with K along the rows of their LDS tiles
template <typename T>
__device__ __forceinline__ f32x4 mma_row_row(const T* A, int lda, int ca, const T* Bm, int ldb, int cb, int k0, f32x4 acc) {
  return Packed<T>::mma(RowFrag<T>::load(A, lda, k0, ca), RowFrag<T>::load(Bm, ldb, k0, cb), acc);
}

// Load the two 49xC tiles of one window, apply LayerNorm (precomputed {mean,rstd}) and the gate weight
// omega[(n*C+c) mod N] (the reference's view-scramble), write them to LDS as T with
// zero rows for padded / dead slots and zero columns C..Cp.  16-byte lane accesses when C % VEC == 0.
// NPARTS cooperating waves deal the 64-lane chunks of the tile round-robin (chunk index = part + NPARTS * j): same
// branch-free code, a runtime chunk offset.
template <typename T, typename DM, int NPARTS = 1>
__device__ __forceinline__ void load_gated_tiles(const rssf_winattn_fwd_params& p, const Geom& g, const float* sLn,
                                                 const T* X, const T* Y, const float* om0, int64_t img, int qh, int qw,
                                                 T* xs, T* ys, int ldx, int lane, int part = 0) {
  constexpr int V = Vec<T>::N;
  if constexpr (DM::C % V == 0) {
    constexpr int cpr = DM::CP / V;
    constexpr int ITERS = ((LP * cpr + 63) / 64 + NPARTS - 1) / NPARTS;
    // branch-free: every lane issues all its global loads back to back (dead / padded slots read token 0 and are
    // zeroed afterwards), so the HBM latency is paid once per window instead of once per iteration.
    Vec<T> vx[ITERS], vy[ITERS];
    float2 sx[ITERS], sy[ITERS];
    int pp[ITERS];
    bool ok[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int e = lane + (part + NPARTS * it) * 64;
      const int t = e / cpr, c0 = (e % cpr) * V;
      const int n = slot_token(g, qh, qw, t);
      ok[it] = n >= 0 && c0 < DM::C && e < LP * cpr;
      const int nn = ok[it] ? n : 0, cc = ok[it] ? c0 : 0;
      const int64_t f = (int64_t)nn * DM::C + cc;
      vx[it].load(X + img * DM::C + f);
      vy[it].load(Y + img * DM::C + f);
      sx[it] = *reinterpret_cast<const float2*>(p.stats_x + (img + nn) * 2);
      sy[it] = *reinterpret_cast<const float2*>(p.stats_y + (img + nn) * 2);
      pp[it] = (int)((unsigned)f % (unsigned)g.N);          // N*C < 2^31 (checked by the entry points): 32-bit modulo
    }
    const bool contiguous = g.N % DM::C == 0;   // the V gate weights of a chunk are contiguous (no wrap inside a token row)
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int e = lane + (part + NPARTS * it) * 64;
      if (e >= LP * cpr) break;
      const int t = e / cpr, c0 = (e % cpr) * V;
      float w0[V], w1[V];
      if (contiguous) {
#pragma unroll
        for (int i = 0; i < V; i += 4) {
          const float4 a = *reinterpret_cast<const float4*>(om0 + pp[it] + i);
          const float4 c = *reinterpret_cast<const float4*>(om0 + g.N + pp[it] + i);
          w0[i] = a.x; w0[i + 1] = a.y; w0[i + 2] = a.z; w0[i + 3] = a.w;
          w1[i] = c.x; w1[i + 1] = c.y; w1[i + 2] = c.z; w1[i + 3] = c.w;
        }
      } else {
        int q = pp[it];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          w0[i] = om0[q]; w1[i] = om0[g.N + q];
          if (++q == g.N) q = 0;
        }
      }
      float fx[V], fy[V];
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float ga = sLn[c0 + i], be = sLn[DM::CP + c0 + i];
        fx[i] = ok[it] ? ((vx[it].get(i) - sx[it].x) * sx[it].y * ga + be) * w0[i] : 0.f;
        fy[i] = ok[it] ? ((vy[it].get(i) - sy[it].x) * sy[it].y * ga + be) * w1[i] : 0.f;
      }
      Vec<T> ox, oy;
      ox.set_all(fx); oy.set_all(fy);
      ox.store(xs + t * ldx + c0);
      oy.store(ys + t * ldx + c0);
    }
  } else {
    for (int e = lane + 64 * part; e < LP * DM::CP; e += 64 * NPARTS) {
      const int t = e / DM::CP, c = e % DM::CP;
      const int n = slot_token(g, qh, qw, t);
      float vx = 0.f, vy = 0.f;
      if (n >= 0 && c < DM::C) {
        const int64_t f = (int64_t)n * DM::C + c;
        const int pp = (int)((unsigned)f % (unsigned)g.N);
        const float* sx = p.stats_x + (img + n) * 2;
        const float* sy = p.stats_y + (img + n) * 2;
        vx = ((ldf(X + img * DM::C + f) - sx[0]) * sx[1] * sLn[c] + sLn[DM::CP + c]) * om0[pp];
        vy = ((ldf(Y + img * DM::C + f) - sy[0]) * sy[1] * sLn[c] + sLn[DM::CP + c]) * om0[g.N + pp];
      }
      stf(xs + t * ldx + c, vx);
      stf(ys + t * ldx + c, vy);
    }
  }
}

}  // namespace wa
}  // namespace rssf
