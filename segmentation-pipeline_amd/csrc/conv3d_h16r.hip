// 3x3x3 / stride 1 / pad 1 convolution with 16-bit operands, REGISTER-RESIDENT WEIGHTS variant for the wide levels
// (W a multiple of 32 lanes, tiles of 8 x 4 x 32 voxels fill the chip): forward and data gradient, c8 in -> c8 out.
// OPT-IN (M355_H16R=1|2; default off): it measures EQUAL to conv3_h16_kernel on the cfg2 layers (d0.c1 150 vs 140 us,
// u0.c0 389 vs 396, d1.c1 74 vs 74, u1.c0 149 vs 151 -- tools/h16r_probe.py), and it is the vehicle of the diagnosis
// of what bounds BOTH kernels (profiles/r03_h16r_diagnosis.txt; M355_H16R_DBG selects loop variants):
//   * the matrix cores run this loop at an in-kernel clock of 1.5-1.9 GHz, not 2.4 (s_memtime / s_memrealtime): dense
//     bf16 MFMAs on random data are power-limited, so the "peak" a launch can see is 1.6-2.0 PFLOP/s;
//   * the MFMA + LDS-read loop alone takes ~9.1k cycles per chunk (216 MFMAs = 6.9k): 76 % busy;
//   * the halo loads add ~2.0k cycles per chunk although they are requested a full chunk ahead and wherever in the
//     chunk they are placed -- but only 0.5k when the same instructions read a cache-resident region: a vector-memory
//     instruction that finds the CU's miss path busy (it sustains ~10 B/clk, and the 64 KB halo tile of a chunk needs
//     ~6.5k cycles of it) blocks the wave's whole in-order instruction stream, MFMAs included.  Two workgroups per CU
//     (conv3_h16_kernel) hide that behind each other's MFMAs; one wave per SIMD cannot, which cancels what it gains
//     from 0.33 instead of 0.75 LDS fragment reads per MFMA.
//
// Design.  conv3_h16_kernel (conv3d_h16.hip) keeps the 27 x 32-channel weight fragments of a 16-channel chunk in LDS
// beside the halo tile and every wave reads them again for its 4 output rows.  Here
//   * one wave per SIMD owns the whole 512-entry register file: TWO output planes x 4 rows x 32 voxels = 8
//     accumulator tiles per wave (128 registers), so every halo fragment read from LDS feeds up to 6 MFMAs (two
//     planes x three tap rows): 72 fragment reads per 216 MFMAs;
//   * the weights never touch LDS: a lane's 27 A fragments of the chunk (one 16-byte item each, the packed layout of
//     pack_w3_h16_kernel, L2-resident and shared by every workgroup) live in 108 registers and are re-loaded IN PLACE
//     for the next chunk, each right after its last use -- a full chunk ahead of its next one;
//   * the halo tile is double-buffered in LDS (2 x 64 KB) and travels in two halves, each requested half a chunk
//     before it is written to the buffer of its position (see the pipeline comment in the kernel): a chunk costs ONE
//     barrier;
//   * items are assigned statically (XCD label -> contiguous region, stride = workgroups of the XCD): all items of a
//     launch cost the same, and neighbouring tiles run at the same time on one XCD.
// Reference ops replaced: nn.Conv3d of Block3d (segmentation_pipeline/models/components.py:36,42,51) forward and its
// autograd data gradient under BASELINE cfg3 / cfg5 (16-bit operands, fp32 accumulate) -- same arithmetic as
// conv3_h16_kernel: products exact in fp32, fp32 accumulation over (chunk, dx, plane, row) in a fixed order.
#include "conv3d_common.hpp"
#include "h16_epilogue.hpp"

namespace m355 {

// DBG (diagnostic variants of the loop, only in the --stamps build: M355_H16R_DBG -- wrong results, for timing the parts only): bit 0 = no halo loads
// / LDS commits, 1 = no weight re-loads, 2 = no MFMAs, 3 = clock stamps over the output, 4 = no LDS commits, 5 = no halo
// loads, 6 = halo loads from a cache-resident region
template <typename HT, int DBG = 0>
__global__ __launch_bounds__(256, 1) void conv3_h16r_kernel(
    const HT* __restrict__ x16, const HT* __restrict__ wp, const float* __restrict__ bias, HT* __restrict__ y16, int CB,
    int Cout, int D, int H, int W, int cout_pad, int tz_tiles, int ty_tiles, int tx_tiles, int otiles, int nchunks,
    int nbatch, int64_t xbs16, int64_t ybs16, float* __restrict__ stat, int stagger) {
  using hx8 = typename H16<HT>::x8;
  constexpr int TZ = 8, TY = 4, TX = 32, RS = TX + 2, PS = (TY + 2) * RS, HV = (TZ + 2) * PS, XI = 2 * HV;
  constexpr int NT = 256, XPER = 16;               // XI = 4080 items in 4096 slots
  static_assert(XI <= NT * XPER, "halo tile does not fit the staging registers");
  __shared__ __attribute__((aligned(16))) hx8 xs[2][NT * XPER];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l32 = lane & 31;
  const int iHW = H * W;
  const int S = D * iHW;
  const int sp_tiles = tz_tiles * ty_tiles * tx_tiles;
  const int total = sp_tiles * otiles * nbatch;

  struct Item {
    int z0, y0, x0, o0, n, sp;
  };
  auto decode = [&](int it) {
    Item q;
    const int ot = it % otiles;                    // channel tiles of one input tile are adjacent (same XCD, same time)
    int sp = (it / otiles) % sp_tiles;
    q.n = it / (otiles * sp_tiles);
    q.sp = sp;
    const int txt = sp % tx_tiles;
    sp /= tx_tiles;
    q.x0 = txt * TX;
    q.y0 = (sp % ty_tiles) * TY;
    q.z0 = (sp / ty_tiles) * TZ;
    q.o0 = ot * 32;
    return q;
  };

  constexpr unsigned OOB = 0x80000000u;
  // halo item e = tid + 256 * i is staged by register xr[i]; the tile travels in two HALVES (A: i < 8, B: i >= 8) that
  // are requested half a chunk apart, see the pipeline below
  unsigned goff[XPER];
  auto compute_goff = [&](const Item& q, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < XPER; ++i) {
      if (i < i0 || i >= i1) continue;
      const int e = tid + NT * i;
      const int h = e / HV, r = e - h * HV;
      const int zz = r / PS, r2 = r - zz * PS;
      const int yy = r2 / RS, xx = r2 - yy * RS;
      const int gz = q.z0 + zz - 1, gy = q.y0 + yy - 1, gx = q.x0 + xx - 1;
      const bool ok = e < XI && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      goff[i] = ok ? (unsigned)(h * S + gz * iHW + gy * W + gx) * 16u : OOB;
      if constexpr ((DBG & 64) != 0) goff[i] = (unsigned)e * 16u;   // (diagnostic: always the same, cache-resident, 64 KB)
    }
  };

  f32x4 xr[XPER];
  hx8 wr[27];
  __amdgpu_buffer_rsrc_t rxa, rxb;
  const f32x4* wsrc;
  const int wlane = half * cout_pad + l32;
  auto x_desc = [&](const Item& q, int ch, bool live) {  // !live: zero-sized descriptor, no memory traffic
    const int nb = min(2, CB - 2 * ch);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(x16 + (int64_t)q.n * xbs16 + (int64_t)(2 * ch) * S * 8), 0,
                                             live ? nb * S * 16 : 0, 0x00020000);
  };
  auto w_setup = [&](const Item& q, int ch) {
    wsrc = reinterpret_cast<const f32x4*>(wp) + (int64_t)ch * 54 * cout_pad + q.o0 + wlane;
  };
  auto fetch_x = [&](int k) {
    xr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k < 8 ? rxa : rxb, goff[k], 0, 0));
  };
  auto fetch_w = [&](int t) { wr[t] = __builtin_bit_cast(hx8, wsrc[(int64_t)t * 2 * cout_pad]); };

  // ---- static item assignment: XCD label xl owns the contiguous region [xl * cpx, ...), its workgroups stride it
  const int G = (int)gridDim.x;
  const int xl = blockIdx.x & 7;
  const int cpx = (total + 7) >> 3;
  const int gq = (G + 7 - xl) >> 3;                // workgroups with this label
  const int rend = min(total, (xl + 1) * cpx);
  int it0 = xl * cpx + (int)(blockIdx.x >> 3);
  if (it0 >= rend) return;                         // (uniform per workgroup; no barrier has been executed)
  if (stagger > 0) {                               // (diagnostic: start the workgroups of an XCD spread in time)
    const int phase = (int)(blockIdx.x >> 3) & 7;
    for (int i = 0; i < phase * stagger; ++i) __builtin_amdgcn_s_sleep(16);   // 16 x 64 cycles each
  }
  unsigned long long t_begin = 0, r_begin = 0, n_chunks = 0;
  if constexpr ((DBG & 8) != 0) {
    t_begin = __builtin_amdgcn_s_memtime();
    r_begin = __builtin_amdgcn_s_memrealtime();
  }
  // The stream of (item, chunk) positions p = 0, 1, 2, ... of this workgroup.  A vector-memory instruction that finds
  // the CU's miss queue full blocks the wave's whole instruction stream, MFMAs included, and a CU sustains ~10 B/clk of
  // misses: the 64 KB halo tile of a chunk is therefore requested at that pace over a whole chunk period -- half B of
  // position p + 1 during the first half of chunk p, half A of position p + 2 during its second half -- and a half is
  // written to the LDS buffer of its position half a chunk after its last request (A(p+1) in the first half of chunk p,
  // B(p+1) in the second).
  auto advance = [&](int it, int ch, int& nit, int& nch) {   // -> live
    if (ch + 1 < nchunks) { nit = it; nch = ch + 1; return true; }
    nit = it + gq; nch = 0;
    return nit < rend;
  };
  int ch0 = 0;
  Item q0 = decode(it0);
  compute_goff(q0, 0, XPER);
  rxa = rxb = x_desc(q0, 0, true);
  w_setup(q0, 0);
#pragma unroll
  for (int k = 0; k < XPER; ++k) fetch_x(k);
#pragma unroll
  for (int t = 0; t < 27; ++t) fetch_w(t);
#pragma unroll
  for (int i = 0; i < XPER; ++i) reinterpret_cast<f32x4*>(xs[0])[tid + NT * i] = xr[i];
  int it1, ch1;
  bool live1 = advance(it0, ch0, it1, ch1);
  Item q1 = q0;
  if (it1 != it0) {
    q1 = decode(live1 ? it1 : it0);
    compute_goff(q1, 0, 8);
  }
  rxa = x_desc(q1, ch1, live1);
  if constexpr (!(DBG & 1)) {
#pragma unroll
    for (int k = 0; k < 8; ++k) fetch_x(k);        // A(1)
  }
  __syncthreads();
  int buf = 0;
  const int xoff = half * HV + 2 * wave * PS + l32;

  f32x16 acc[2][4];
#pragma unroll
  for (int op = 0; op < 2; ++op)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[op][g][r] = 0.f;
  while (true) {
    // chunk at position p = (it0, ch0); p + 1 = (it1, ch1): its half A is in flight, its half B and weights start here
    if (it1 != it0) compute_goff(q1, 8, XPER);
    rxb = x_desc(q1, ch1, live1);
    w_setup(q1, ch1);
    int it2 = it1, ch2 = ch1;
    bool live2 = false;
    Item q2 = q1;
    const hx8* xb = xs[buf] + xoff;
    f32x4* xw = reinterpret_cast<f32x4*>(xs[buf ^ 1]) + tid;
    // step s = (dx, ip): the 6 row fragments of input plane 2 * wave + ip at column offset dx feed output plane op
    // (op = 0, 1) through tap plane dz = ip - op, rows g = j - dy
    hx8 f[2][6];
    auto lds_step = [&](int s, int slot) {
      const int dx = s >> 2, ip = s & 3;
#pragma unroll
      for (int j = 0; j < 6; ++j) f[slot][j] = xb[ip * PS + j * RS + dx];
    };
    lds_step(0, 0);
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const int dx = s >> 2, ip = s & 3;
      if (s == 6) {   // second half: position p + 2 becomes known, its half A starts
        live2 = live1 && advance(it1, ch1, it2, ch2);
        if (it2 != it1) {
          q2 = decode(live2 ? it2 : it1);
          compute_goff(q2, 0, 8);
        }
        rxa = x_desc(q2, ch2, live2);
      }
      if (s + 1 < 12) lds_step(s + 1, (s + 1) & 1);
      // 8 requests and 8 LDS writes per half chunk, over its 6 steps: 2 1 1 2 1 1
      const int h6 = s % 6, k0 = h6 + (h6 >= 1) + (h6 >= 4) - (h6 >= 1), kn = (h6 == 0 || h6 == 3) ? 2 : 1;
      const int kb = (h6 == 0 ? 0 : h6 == 1 ? 2 : h6 == 2 ? 3 : h6 == 3 ? 4 : h6 == 4 ? 6 : 7);
      (void)k0;
      if constexpr (!(DBG & 1)) {
#pragma unroll
        for (int k = 0; k < kn; ++k) {
          if (s < 6) {
            if constexpr (!(DBG & 16)) xw[NT * (kb + k)] = xr[kb + k];           // A(p+1) -> LDS
            if constexpr (!(DBG & 32)) fetch_x(8 + kb + k);                      // B(p+1) requested
            if constexpr ((DBG & 16) != 0) asm volatile("" ::"v"(xr[kb + k]));
          } else {
            if constexpr (!(DBG & 16)) xw[NT * (8 + kb + k)] = xr[8 + kb + k];   // B(p+1) -> LDS
            if constexpr (!(DBG & 32)) fetch_x(kb + k);                          // A(p+2) requested
            if constexpr ((DBG & 16) != 0) asm volatile("" ::"v"(xr[8 + kb + k]));
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int dz = ip - op, g = j - dy;
            if (dz >= 0 && dz <= 2 && g >= 0 && g < 4 && !(DBG & 4))
              acc[op][g] = H16<HT>::mfma(wr[(dz * 3 + dy) * 3 + dx], f[s & 1][j], acc[op][g]);
          }
      if (ip >= 1 && !(DBG & 2)) {  // taps (dz = ip - 1, *, dx) are done for this chunk: their registers take the next chunk's
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) fetch_w(((ip - 1) * 3 + dy) * 3 + dx);
      }
      {
        const int nm = (ip == 0 || ip == 3) ? 12 : 24;
        const int nvm = ((DBG & 33) ? 0 : kn) + ((ip >= 1 && !(DBG & 2)) ? 3 : 0);
        const int nwr = (DBG & 17) ? 0 : kn;
        const int nrd = s + 1 < 12 ? 6 : 0;
#pragma unroll
        for (int m = 0; m < nm; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // MFMA
          if (m < nrd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // DS read
          if (m >= 1 && m - 1 < nwr) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
          if (m >= 2 && (m - 2) % 2 == 0 && (m - 2) / 2 < nvm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    buf ^= 1;
    if constexpr ((DBG & 8) != 0) ++n_chunks;

    if (ch0 == nchunks - 1) {   // ---- output tiles of item it0: two planes per wave ----
      const int xg = q0.x0 + l32;
#pragma unroll
      for (int op = 0; op < 2; ++op) {
        const int z = q0.z0 + 2 * wave + op;
        float* st = stat ? stat + (((int64_t)q0.n * sp_tiles + q0.sp) * TZ + 2 * wave + op) * Cout * 2 : nullptr;
        store_conv_tile_c8<4, 1, HT>(acc[op], y16 + (int64_t)q0.n * ybs16, bias, q0.o0, Cout, z, q0.y0, xg, 0, half, H,
                                     W, (int64_t)S, z < D && xg < W, st);
      }
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[op][g][r] = 0.f;
    }
    if (!live1) break;
    it0 = it1; ch0 = ch1; q0 = q1;
    it1 = it2; ch1 = ch2; q1 = q2; live1 = live2;
  }
  if constexpr ((DBG & 8) != 0) {   // diagnostic: (core cycles, 100 MHz ticks, chunks) of this workgroup over the output
    __syncthreads();
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(y16) + (int64_t)blockIdx.x * 4;
      o[0] = __builtin_amdgcn_s_memtime() - t_begin;
      o[1] = __builtin_amdgcn_s_memrealtime() - r_begin;
      o[2] = n_chunks;
    }
  }
}

int launch_h16r(const FwdPlan& p, int compute, const void* x16, int64_t xbs16, const void* wp, const float* bias, void* y16,
                int N, int kin, int mout, int D, int H, int W, int64_t ybs16, hipStream_t st, float* stat) {
  const int64_t items = (int64_t)p.tz_tiles * p.ty_tiles * p.tx_tiles * p.otiles * N;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(items, num_cus()));
#define M355_H16R_LAUNCH(HT, DBG)                                                                                          \
  hipLaunchKernelGGL((conv3_h16r_kernel<HT, DBG>), dim3(grid), dim3(256), 0, st, (const HT*)x16, (const HT*)wp, bias,       \
                     (HT*)y16, (int)c8_blocks(kin), mout, D, H, W, p.mout_pad, p.tz_tiles, p.ty_tiles, p.tx_tiles, p.otiles, \
                     p.nchunks, N, xbs16, ybs16, stat, stagger)
  static const int stagger = getenv("M355_H16R_STAGGER") ? atoi(getenv("M355_H16R_STAGGER")) : 0;
#ifdef M355_H16_STAMPS
  // diagnostic build (build.py --stamps, M355_LIB_PATH): loop variants with WRONG results, for timing the parts
  // (tools/h16r_probe.py; bit 3 adds the in-kernel clock stamps)
  static const int dbg = getenv("M355_H16R_DBG") ? atoi(getenv("M355_H16R_DBG")) : 0;
  if (compute == M355_COMPUTE_BF16 && dbg) {
    switch (dbg) {
      case 8: M355_H16R_LAUNCH(__bf16, 8); break;      // everything
      case 9: M355_H16R_LAUNCH(__bf16, 9); break;      // no halo loads / LDS commits
      case 10: M355_H16R_LAUNCH(__bf16, 10); break;    // no weight re-loads
      case 11: M355_H16R_LAUNCH(__bf16, 11); break;    // neither: MFMAs + LDS fragment reads only
      case 12: M355_H16R_LAUNCH(__bf16, 12); break;    // no MFMAs: the memory side alone
      case 24: M355_H16R_LAUNCH(__bf16, 24); break;    // halo loads without the LDS commits
      case 40: M355_H16R_LAUNCH(__bf16, 40); break;    // LDS commits without the halo loads
      case 72: M355_H16R_LAUNCH(__bf16, 72); break;    // halo loads from a cache-resident region
      default: M355_H16R_LAUNCH(__bf16, 0);
    }
    return check_launch("conv3_h16r(diagnostic)");
  }
#endif
  if (compute == M355_COMPUTE_BF16)
    M355_H16R_LAUNCH(__bf16, 0);
  else
    M355_H16R_LAUNCH(_Float16, 0);
#undef M355_H16R_LAUNCH
  return check_launch("conv3_h16r");
}

}  // namespace m355
