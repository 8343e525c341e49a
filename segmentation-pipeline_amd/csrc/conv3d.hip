// conv3d for gfx950: fp32 implicit-GEMM on v_mfma_f32_32x32x2_f32 (exact fp32,
// k-ordered fma chain) for the 3x3x3 / stride 1 / pad 1 convolutions that carry
// ~97 % of the U-Net FLOPs, plus generic direct kernels for any other
// kernel-size / stride / padding (the reference's BlurConv3d, k=4 s=2, reaches the
// 3x3x3 kernels through a space-to-depth rearrangement, models/components.py).
//
// Reference ops replaced: nn.Conv3d in Block3d (models/components.py:36,42,51),
// out conv (models/modular_unet.py:83,99), F.conv3d in BlurConv3d
// (components.py:119); autograd of those for bwd-data / bwd-weight.
//
// GEMM view of the 3x3x3 forward:  Y[o, v] = sum_k Wp[k, o] * X[k, v]
//   M = Cout (A operand = packed weights), N = voxels (B operand = input read
//   from an LDS halo tile with compile-time tap offsets), K = 27 * Cin.
// The MFMA's k-pair (lanes 0-31 / 32-63) is (channel c, channel c+1) at the same
// tap, so both halves use one immediate offset.
#include "conv3d_common.hpp"

namespace m355 {

// ---------------------------------------------------------------- weight pack
// fwd:      wp[(c*27 + tap)*cout_pad + o]          = w[(o*Cin + c)*27 + tap]
// bwd-data: wp[(o*27 + (26-tap))*cin_pad32 + c]    = w[(o*Cin + c)*27 + tap]
//           (the data-gradient is a conv of dy with the flipped, transposed filter)
// Padded rows/cols are zero-filled.
__global__ void pack_w3_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                               int Cin, int kin_pad, int mout_pad, int transpose, int* __restrict__ counter) {
  // logical conv being run: K-channels = kin (padded to kin_pad), M-channels = mout_pad
  if (counter && blockIdx.x == 0 && threadIdx.x < 16) counter[threadIdx.x] = 0;  // work queues of the persistent conv kernel
  pack_w3_body(w, wp, Cout, Cin, kin_pad, mout_pad, transpose, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------ MFMA fwd kernel
// One output tile per workgroup (used when a launch has no more tiles than resident workgroups;
// otherwise conv3_mfma_fwd_p_kernel below).  Two workgroups per CU up to NTW = 4, one for NTW = 8,
// i.e. one or two waves per SIMD: the next chunk's input halo tile and weights travel
// global -> registers -> the other LDS buffer while the current chunk is multiplied, with the
// prefetch instructions interleaved into the MFMA stream; one barrier per chunk (~14k MFMA cycles).
template <int NTW, int GX>
__global__ __launch_bounds__(256, 1) void conv3_mfma_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ add, float* __restrict__ y, float* __restrict__ slab, int Cin,
    int Cout, int D, int H, int W, int cout_pad, int ty_tiles, int tx_tiles, int nchunks,
    int ksplit, int64_t xbs, int64_t ybs, int64_t slab_stride, float* __restrict__ stat, int otiles, int order) {
  using T = FwdTile<NTW, GX>;
  constexpr int GY = T::GY, TZ = T::TZ, TY = T::TY, TX = T::TX, RS = T::RS, PS = T::PS,
                CS = T::CS, CC = T::CC;
  constexpr int XE = CC * CS;                 // floats of one staged input chunk
  constexpr int XPER = (XE + 255) / 256;      // per-thread elements
  constexpr int WE4 = CC * 27 * 8;            // float4s of one staged weight chunk
  constexpr int WPER = (WE4 + 255) / 256;
  __shared__ float xs[2][XE];
  __shared__ __attribute__((aligned(16))) float ws[2][CC * 27 * 32];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l32 = lane & 31;
  const int ly = l32 / GX, lx = l32 % GX;

  // XCD-aware placement (speed only, a bijection of the grid's x range): workgroups are dealt to the 8 XCDs round-robin
  // in launch order (x fastest); when the workgroup count along x is a multiple of 8 every XCD gets a contiguous run
  // of tiles, whose shared halos then hit in its L2.  The channel tile is the FASTEST index of x: the workgroups that
  // need the same input tile for different output channels run side by side on one XCD (the channel tile as the
  // outer index made every pass over the channel tiles re-read the whole input from HBM).
  int bt = blockIdx.x;
  if ((gridDim.x & 7) == 0) bt = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  const int sp_count = (int)gridDim.x / otiles;
  const int otile = (order & 2) ? bt % otiles : bt / sp_count;
  bt = (order & 2) ? bt / otiles : bt % sp_count;
  const int sp_index = bt;   // spatial tile (statistics slot)
  const int txt = bt % tx_tiles;
  bt /= tx_tiles;
  const int tyt = bt % ty_tiles;
  const int tzt = bt / ty_tiles;
  const int z0 = tzt * TZ, y0 = tyt * TY, x0 = txt * TX;
  const int o0 = otile * 32;
  const int n = blockIdx.z / ksplit;
  const int ks = blockIdx.z % ksplit;

  const int cps = (nchunks + ksplit - 1) / ksplit;
  const int ch_begin = ks * cps;
  const int ch_end = min(nchunks, ch_begin + cps);

  const float* xn = x + (int64_t)n * xbs;
  const int64_t HW = (int64_t)H * W;
  const int DHW = (int)(HW * D);

  // chunk-invariant gather offsets of this thread's elements, in bytes; zero padding and elements
  // past the tile get an offset no buffer descriptor covers, so the hardware returns 0 for them
  constexpr unsigned OOB = 0x80000000u;
  unsigned goff[XPER];
#pragma unroll
  for (int i = 0; i < XPER; ++i) {
    const int e = tid + 256 * i;
    unsigned off = OOB;
    if (e < XE) {
      const int c = e / CS, r = e - c * CS;
      const int zz = r / PS, r2 = r - zz * PS;
      const int yy = r2 / RS, xx = r2 - yy * RS;
      const int gz = z0 + zz - 1, gy = y0 + yy - 1, gx = x0 + xx - 1;
      if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
        off = (unsigned)(c * DHW + gz * (int)HW + gy * W + gx) * 4u;
    }
    goff[i] = off;
  }

  f32x16 acc[NTW];
#pragma unroll
  for (int g = 0; g < NTW; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  // Prefetch of the next chunk into registers.  x comes through a raw buffer descriptor that
  // covers exactly the channels of the chunk that exist (base at channel c0, num_records =
  // min(CC, Cin - c0) channels): padding, tile overhang and channels past Cin are out of range and
  // read as 0, so there are no validity masks, no branches, and the commit is plain LDS writes.
  // The prefetch is cut into per-step items issued inside the MFMA loop (see below).
  float xr[XPER];
  f32x4 wr[WPER];
  __amdgpu_buffer_rsrc_t rx;
  const float* wsrc = wp;
  auto chunk_setup = [&](int ch, bool live) {  // !live: zero-sized descriptor, every load returns 0
    const int c0 = ch * CC;
    rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + (int64_t)c0 * DHW), 0,
                                           live ? min(CC, Cin - c0) * DHW * 4 : 0, 0x00020000);
    wsrc = wp + (int64_t)c0 * 27 * cout_pad + o0;
  };
  auto fetch_item = [&](int s) {  // s is a compile-time constant wherever this is called
    if (s < XPER) xr[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, goff[s], 0, 0));
    if (s < WPER) {
      const int idx = tid + 256 * s;
      const int idc = idx < WE4 ? idx : WE4 - 1;  // clamp: keeps the array fully scalarised
      wr[s] = *reinterpret_cast<const f32x4*>(wsrc + (int64_t)(idc >> 3) * cout_pad + (idc & 7) * 4);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XPER; ++i)
      if (tid + 256 * i < XE) xs[buf][tid + 256 * i] = xr[i];
#pragma unroll
    for (int j = 0; j < WPER; ++j)
      if (tid + 256 * j < WE4) *reinterpret_cast<f32x4*>(&ws[buf][(tid + 256 * j) * 4]) = wr[j];
  };
  constexpr int NSTEP = (CC / 2) * 27;
  static_assert(XPER <= NSTEP && WPER <= NSTEP, "one prefetch item per MFMA step");

  if (ch_begin < ch_end) {
    chunk_setup(ch_begin, true);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) fetch_item(s);
    commit(0);
  }
  __syncthreads();

  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int cur = (ch - ch_begin) & 1;
    const bool more = ch + 1 < ch_end;
    chunk_setup(more ? ch + 1 : ch, more);  // unconditional, so the loop body has no branch
    const float* xb = xs[cur] + half * CS + wave * PS + ly * RS + lx;
    const float* wb = ws[cur] + half * (27 * 32) + l32;
    // ---- MFMA over K = CC * 27, software-pipelined over the (channel pair, tap) steps: the LDS
    // reads of step s+1 and one item of the next chunk's prefetch are issued before the MFMAs of
    // step s.  The sched_barriers pin that order; left alone, hipcc sinks LDS reads to just before
    // their MFMA (waiting lgkmcnt(0) on them) and hoists the whole prefetch in front of the loop,
    // which with one wave per SIMD leaves the matrix core idle meanwhile.
    float av[2], bv[2][NTW];
    auto lds_step = [&](int s, int slot) {  // s is a compile-time constant after unrolling
      const int cp = s / 27, tap = s % 27;
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      av[slot] = wb[(2 * cp * 27 + tap) * 32];
#pragma unroll
      for (int g = 0; g < NTW; ++g) bv[slot][g] = xb[2 * cp * CS + dz * PS + (g * GY + dy) * RS + dx];
    };
    lds_step(0, 0);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + 1 < NSTEP) lds_step(s + 1, (s + 1) & 1);
      fetch_item(s);
#pragma unroll
      for (int g = 0; g < NTW; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1], bv[s & 1][g], acc[g], 0, 0, 0);
      // issue order inside the step: each MFMA is followed by up to two of the LDS reads of step
      // s+1, then the prefetch loads -- they go out while the matrix core is busy
#pragma unroll
      for (int g = 0; g < NTW; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);    // VMEM read
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
  const int z = z0 + wave;
  const int xg = x0 + lx;
  const bool lane_ok = z < D && xg < W;
  if (ksplit == 1) {
    // statistics slot of this wave: stat[n][spatial tile * 4 + wave][Cout][2]
    float* st = stat ? stat + (((int64_t)n * sp_count + sp_index) * 4 + wave) * Cout * 2 : nullptr;
    store_conv_tile<NTW, GY>(acc, y + (int64_t)n * ybs, add ? add + (int64_t)n * ybs : nullptr, bias, o0, Cout, z,
                             y0, xg, ly, half, D, H, W, lane_ok, st);
  } else {
    store_conv_tile<NTW, GY>(acc, slab + (int64_t)ks * slab_stride + (int64_t)n * Cout * D * HW, nullptr, nullptr,
                             o0, Cout, z, y0, xg, ly, half, D, H, W, lane_ok, nullptr);
  }
}

// ---- persistent variant of conv3_mfma_fwd_kernel ----
// A workgroup of the kernel above lives for nchunks/ksplit chunks and pays ~2.5 chunks of fixed
// cost around them (measured: 8-chunk layers reach 112 TFLOP/s, 48-chunk layers 140): the first
// chunk's load latency + commit, the output stores, the launch of the next workgroup.  Here the grid
// is one residency (2 workgroups per CU) and every workgroup takes items from a queue (item = output
// tile x 32-channel tile x sample x split, same order as the grid above; XCD-aware, see "work queue"
// below), so a workgroup that starts late -- another kernel, e.g. an RCCL collective, still holding
// its CU -- simply takes fewer items; every item's result is independent of who computes it, so
// this stays bit-reproducible.  The
// (item, chunk) sequence is flattened: during the last chunk of an item the FIRST chunk of the next
// item is prefetched, so the MFMA stream only stops for the output stores.
// M16: the 16-row remainder tile (see store_conv_tile16): the items are the spatial tiles x ONE 16-channel tile at
// o_base; one MFMA 16x16x4 per (tap, 16-voxel half group) covers the chunk's 4 channels, 27 steps per chunk.
// The LDS channel stride is padded to == 16 (mod 32) so the four channels of a B fragment (lane >> 4) and the two
// voxel halves fall on distinct banks; the weights of odd channels are read from the upper 16 columns of their
// LDS row (both halves of a row are staged with the same 16 output channels).
template <int NTW, int GX, bool M16 = false>
__global__ __launch_bounds__(256, (NTW <= 4 ? 2 : 1)) void conv3_mfma_fwd_p_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ add, float* __restrict__ y, float* __restrict__ slab, int Cin,
    int Cout, int D, int H, int W, int cout_pad, int tz_tiles, int ty_tiles, int tx_tiles, int otiles,
    int nchunks, int ksplit, int nbatch, int64_t xbs, int64_t ybs, int64_t slab_stride,
    float* __restrict__ stat, int* __restrict__ work_counter, int o_base, int order) {
  using T = FwdTile<NTW, GX>;
  constexpr int GY = T::GY, TZ = T::TZ, TY = T::TY, TX = T::TX, RS = T::RS, PS = T::PS,
                CS0 = T::CS, CC = T::CC;
  constexpr int CS = M16 ? ((CS0 + 31) / 32) * 32 + 16 : CS0;   // LDS channel stride (elements past CS0: padding)
  constexpr int XE = CC * CS;
  constexpr int XPER = (XE + 255) / 256;
  constexpr int WE4 = CC * 27 * 8;
  constexpr int WPER = (WE4 + 255) / 256;
  constexpr int NSTEP = M16 ? 27 : (CC / 2) * 27;
  static_assert(XPER <= NSTEP && WPER <= NSTEP, "one prefetch item per MFMA step");
  __shared__ float xs[2][XE];
  __shared__ __attribute__((aligned(16))) float ws[2][CC * 27 * 32];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l32 = lane & 31;
  const int ly = l32 / GX, lx = l32 % GX;
  const int64_t HW = (int64_t)H * W;
  const int DHW = (int)(HW * D);
  const int iHW = (int)HW;

  const int sp_tiles = tz_tiles * ty_tiles * tx_tiles;
  const int total = sp_tiles * otiles * nbatch * ksplit;
  const int cps = (nchunks + ksplit - 1) / ksplit;

  struct Item {
    int z0, y0, x0, o0, n, ks, ch_begin, ch_end, sp;
  };
  // Tile order inside a region: the ~64 items in flight on one XCD should form a CUBE of tiles (all x, 4 y, 4 z), not
  // a slab one tile thick: with the linear (x, y, z) order every tile's two z-halo planes (a third of its input) are
  // re-read from HBM when the next slab comes around (PMC: 1.6x the algorithmic bytes); in the blocked order they
  // are shared with a z neighbour that is in flight at the same time.  A bijection of the item range: results unchanged.
  const bool cube = (ty_tiles & 3) == 0 && (tz_tiles & 3) == 0 && (order & 1);
  auto decode = [&](int it) {
    Item q;
    // channel tile fastest (the items that share an input tile are adjacent: same XCD, same time), then the spatial tile
    const int ot = (order & 2) ? it % otiles : (it / sp_tiles) % otiles;
    int sp = (order & 2) ? (it / otiles) % sp_tiles : it % sp_tiles, r = it / (otiles * sp_tiles);
    const int txt = sp % tx_tiles;
    sp /= tx_tiles;
    int tyt, tzt;
    if (cube) {   // sp = ((tz_hi * (ty_tiles / 4) + ty_hi) * 4 + tz_lo) * 4 + ty_lo
      const int ty_lo = sp & 3, tz_lo = (sp >> 2) & 3, hi = sp >> 4;
      const int tyh = ty_tiles >> 2;
      tyt = (hi % tyh) * 4 + ty_lo;
      tzt = (hi / tyh) * 4 + tz_lo;
    } else {
      tyt = sp % ty_tiles;
      tzt = sp / ty_tiles;
    }
    q.sp = (tzt * ty_tiles + tyt) * tx_tiles + txt;   // canonical tile index (statistics slot)
    q.x0 = txt * TX;
    q.y0 = tyt * TY;
    q.z0 = tzt * TZ;
    q.o0 = o_base + ot * 32;
    q.ks = r % ksplit;
    q.n = r / ksplit;
    q.ch_begin = q.ks * cps;
    q.ch_end = min(nchunks, q.ch_begin + cps);
    return q;
  };

  // gather offsets of this thread's halo elements for item q, in bytes from the chunk's first channel;
  // zero padding and elements past the tile get an offset no descriptor covers (hardware returns 0).
  // Recomputed per item from the element index (divisions by compile-time constants) rather than
  // kept as per-thread tables: 40 registers this kernel does not have.
  constexpr unsigned OOB = 0x80000000u;
  unsigned goff[XPER];
  auto compute_goff = [&](const Item& q) {
#pragma unroll
    for (int i = 0; i < XPER; ++i) {
      const int e = tid + 256 * i;
      const int c = e / CS, r = e - c * CS;
      const int zz = r / PS, r2 = r - zz * PS;
      const int yy = r2 / RS, xx = r2 - yy * RS;
      const int gz = q.z0 + zz - 1, gy = q.y0 + yy - 1, gx = q.x0 + xx - 1;
      const bool ok = e < XE && r < CS0 && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H &&
                      (unsigned)gx < (unsigned)W;
      goff[i] = ok ? (unsigned)(c * DHW + gz * iHW + gy * W + gx) * 4u : OOB;
    }
  };

  float xr[XPER];
  f32x4 wr[WPER];
  __amdgpu_buffer_rsrc_t rx;
  const float* wsrc = wp;
  auto chunk_setup = [&](const Item& q, int ch, bool live) {  // !live: zero-sized descriptor
    const int c0 = ch * CC;
    rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)q.n * xbs + (int64_t)c0 * DHW), 0,
                                           live ? min(CC, Cin - c0) * DHW * 4 : 0, 0x00020000);
    wsrc = wp + (int64_t)c0 * 27 * cout_pad + q.o0;
  };
  auto fetch_item = [&](int s) {
    if (s < XPER) xr[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, goff[s], 0, 0));
    if (s < WPER) {
      const int idx = tid + 256 * s;
      const int idc = idx < WE4 ? idx : WE4 - 1;
      wr[s] = *reinterpret_cast<const f32x4*>(wsrc + (int64_t)(idc >> 3) * cout_pad + (idc & (M16 ? 3 : 7)) * 4);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XPER; ++i)
      if (tid + 256 * i < XE) xs[buf][tid + 256 * i] = xr[i];
#pragma unroll
    for (int j = 0; j < WPER; ++j)
      if (tid + 256 * j < WE4) *reinterpret_cast<f32x4*>(&ws[buf][(tid + 256 * j) * 4]) = wr[j];
  };

  // ---- work queue.  The items are dealt in eight contiguous regions, one per XCD label
  // (blockIdx % 8: blocks b and b + 8 are observed to share an XCD and therefore an L2), so the
  // tiles in flight on one L2 are neighbours and their shared input halos hit in it.  A workgroup
  // takes its region's items in order from an atomic counter (its first item is static) and steals
  // from the other regions once its own is empty.  Only thread 0 talks to the counters.
  __shared__ int next_item_s;
  const int G = (int)gridDim.x;
  const int xl = blockIdx.x & 7;
  const int cpx = (total + 7) >> 3;  // items per region
  auto region_size = [&](int r) { return max(0, min(cpx, total - r * cpx)); };
  auto region_static = [&](int r) { return min(region_size(r), (G - r + 7) >> 3); };  // workgroups starting there
  auto steal = [&]() {  // thread 0; `total` = nothing left anywhere
    // One 32-byte read of all eight ticket counters first: when every region is drained (what every
    // workgroup finds once, at the end of its life) that is one memory round trip instead of seven
    // dependent atomics (~15-40k cycles, a quarter of the lifetime of a short 16-bit launch).
    int seen[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) seen[r] = __hip_atomic_load(work_counter + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int a = 1; a < 8; ++a) {
      const int r = (xl + a) & 7;
      if (region_static(r) + seen[r] >= region_size(r)) continue;  // nothing dynamic left there (counters only grow)
      const int k = region_static(r) + atomicAdd(work_counter + r, 1);
      if (k < region_size(r)) return r * cpx + k;
    }
    return total;
  };
  auto resolve = [&](int taken) {  // thread 0: `taken` = this workgroup's ticket in its own region
    const int k = region_static(xl) + taken;
    return k < region_size(xl) ? xl * cpx + k : steal();
  };
  int it = xl * cpx + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= region_size(xl)) {  // more workgroups than items in this region (uniform)
    if (tid == 0) next_item_s = steal();
    __syncthreads();
    it = next_item_s;
    __syncthreads();
    if (it >= total) {
      queue_leave(work_counter);
      return;
    }
  }
  Item cur = decode(it);
  compute_goff(cur);
  chunk_setup(cur, cur.ch_begin, true);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) fetch_item(s);
  commit(0);
  __syncthreads();
  int buf = 0;

  f32x16 acc[M16 ? 1 : NTW];
  f32x4v acc16[M16 ? NTW : 1][2];
  while (true) {
    // thread 0 takes a ticket for this workgroup's next item now; the item is published through
    // LDS after the first chunk (a barrier later) and consumed at the start of the last chunk
    int pending = 0;
    if (tid == 0) pending = atomicAdd(work_counter + xl, 1);
    if (cur.ch_end - cur.ch_begin == 1) {  // single-chunk items: no chunk to hide the round trip behind
      if (tid == 0) next_item_s = resolve(pending);
      __syncthreads();
    }
    if constexpr (M16) {
#pragma unroll
      for (int g = 0; g < NTW; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[g][0][r] = acc16[g][1][r] = 0.f;
    } else {
#pragma unroll
      for (int g = 0; g < NTW; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    }
    Item nxt = cur;
    int nit = total;
    for (int ch = cur.ch_begin; ch < cur.ch_end; ++ch) {
      if (ch + 1 < cur.ch_end) {
        chunk_setup(cur, ch + 1, true);
      } else {  // last chunk of this item: prefetch the first chunk of the next one
        nit = next_item_s;
        const bool live = nit < total;
        nxt = decode(live ? nit : it);
        compute_goff(nxt);
        chunk_setup(nxt, nxt.ch_begin, live);
      }
      if constexpr (M16) {
        const int kq = lane >> 4, l16 = lane & 15;                       // channel of the chunk, voxel / output row
        const float* wb = ws[buf] + kq * (27 * 32) + l16 + 16 * (kq & 1);
        const float* xb0 = xs[buf] + kq * CS + wave * PS + (l16 / GX) * RS + l16 % GX;                  // x-half 0
        const float* xb1 = xs[buf] + kq * CS + wave * PS + ((16 + l16) / GX) * RS + (16 + l16) % GX;    // x-half 1
        float av[2], bv[2][NTW][2];
        auto lds_step = [&](int s, int slot) {  // s = tap
          const int dz = s / 9, dy = (s / 3) % 3, dx = s % 3;
          av[slot] = wb[s * 32];
#pragma unroll
          for (int g = 0; g < NTW; ++g) {
            bv[slot][g][0] = xb0[dz * PS + (g * GY + dy) * RS + dx];
            bv[slot][g][1] = xb1[dz * PS + (g * GY + dy) * RS + dx];
          }
        };
        lds_step(0, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
          if (s + 1 < NSTEP) lds_step(s + 1, (s + 1) & 1);
          fetch_item(s);
#pragma unroll
          for (int g = 0; g < NTW; ++g)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
              acc16[g][h2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s & 1], bv[s & 1][g][h2], acc16[g][h2], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 2 * NTW; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
          }
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);    // VMEM read
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      const float* xb = xs[buf] + half * CS + wave * PS + ly * RS + lx;
      const float* wb = ws[buf] + half * (27 * 32) + l32;
      float av[2], bv[2][NTW];
      auto lds_step = [&](int s, int slot) {
        const int cp = s / 27, tap = s % 27;
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        av[slot] = wb[(2 * cp * 27 + tap) * 32];
#pragma unroll
        for (int g = 0; g < NTW; ++g) bv[slot][g] = xb[2 * cp * CS + dz * PS + (g * GY + dy) * RS + dx];
      };
      lds_step(0, 0);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        if (s + 1 < NSTEP) lds_step(s + 1, (s + 1) & 1);
        fetch_item(s);
#pragma unroll
        for (int g = 0; g < NTW; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1], bv[s & 1][g], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NTW; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);    // VMEM read
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      if (ch == cur.ch_begin && cur.ch_end - cur.ch_begin > 1 && tid == 0) next_item_s = resolve(pending);
      commit(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }

    // ---- output tile of `cur` ----
    if constexpr (M16) {
      const int z = cur.z0 + wave;
      if (ksplit == 1) {
        float* st = stat ? stat + (((int64_t)cur.n * sp_tiles + cur.sp) * 4 + wave) * Cout * 2 : nullptr;
        store_conv_tile16<NTW, GX>(acc16, y + (int64_t)cur.n * ybs, add ? add + (int64_t)cur.n * ybs : nullptr, bias,
                                   cur.o0, Cout, z, cur.y0, cur.x0, lane, D, H, W, st);
      } else {
        store_conv_tile16<NTW, GX>(acc16, slab + (int64_t)cur.ks * slab_stride + (int64_t)cur.n * Cout * D * HW, nullptr,
                                   nullptr, cur.o0, Cout, z, cur.y0, cur.x0, lane, D, H, W, nullptr);
      }
    } else {
      const int z = cur.z0 + wave;
      const int xg = cur.x0 + lx;
      const bool lane_ok = z < D && xg < W;
      if (ksplit == 1) {
        float* st = stat ? stat + (((int64_t)cur.n * sp_tiles + cur.sp) * 4 + wave) * Cout * 2 : nullptr;
        store_conv_tile<NTW, GY>(acc, y + (int64_t)cur.n * ybs, add ? add + (int64_t)cur.n * ybs : nullptr, bias,
                                 cur.o0, Cout, z, cur.y0, xg, ly, half, D, H, W, lane_ok, st);
      } else {
        store_conv_tile<NTW, GY>(acc, slab + (int64_t)cur.ks * slab_stride + (int64_t)cur.n * Cout * D * HW, nullptr,
                                 nullptr, cur.o0, Cout, z, cur.y0, xg, ly, half, D, H, W, lane_ok, nullptr);
      }
    }
    if (nit >= total) break;
    it = nit;
    cur = nxt;
  }
  queue_leave(work_counter);
}

// A 32-row MFMA tile would carry only Cout useful rows.  z-Toeplitz packing fills the rows
// with (o, s), s = 0..7 being eight consecutive output planes of one (y, x) column:
//   y[o, z0+s, y, x] = sum_{c,dy,dx} sum_{u=0..9} Wz[(c,dy,dx,u), (o,s)] * x[c, z0-1+u, y+dy-1, x+dx-1]
//   Wz[(c,dy,dx,u), (o,s)] = w[o,c,u-s,dy,dx] if 0 <= u-s <= 2 else 0
// K grows by 10/3 but each MFMA column now yields 8 outputs: 2.4x fewer MFMAs than the padded tile.
// Tile = 8 z x 8 y x 32 x; wave w owns rows y = 2w, 2w+1; 2 input channels (one MFMA k-pair) per chunk.
constexpr int TZ_K = 90;  // (dy, dx, u) combinations
__global__ void pack_w3_toeplitz_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                        int Cin, int kin_pad) {
  const int64_t total = (int64_t)kin_pad * TZ_K * 32;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i & 31);
    int64_t r = i >> 5;
    const int t = (int)(r % TZ_K);
    const int c = (int)(r / TZ_K);
    const int u = t % 10, dyx = t / 10;  // dyx = dy*3 + dx
    const int o = m >> 3, sft = m & 7;
    const int dz = u - sft;
    float v = 0.f;
    if (c < Cin && o < Cout && dz >= 0 && dz <= 2) v = w[((int64_t)o * Cin + c) * 27 + dz * 9 + dyx];
    wp[i] = v;
  }
}

__global__ __launch_bounds__(256, 2) void conv3_mfma_fwd_smallcout_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ add, float* __restrict__ y, int Cin, int Cout, int D, int H, int W,
    int ty_tiles, int tx_tiles, int nchunks, int64_t xbs, int64_t ybs) {
  constexpr int TZ = 8, TY = 8, TX = 32, RS = TX + 2, PS = (TY + 2) * RS, CS = (TZ + 2) * PS;
  constexpr int XE = 2 * CS, XPER = (XE + 255) / 256;   // 6800 floats
  constexpr int WE4 = 2 * TZ_K * 8, WPER = (WE4 + 255) / 256;
  __shared__ float xs[XE];
  __shared__ __attribute__((aligned(16))) float ws[2 * TZ_K * 32];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  int bt = blockIdx.x;
  const int txt = bt % tx_tiles;
  bt /= tx_tiles;
  const int tyt = bt % ty_tiles;
  const int tzt = bt / ty_tiles;
  const int z0 = tzt * TZ, y0 = tyt * TY, x0 = txt * TX;
  const int n = blockIdx.y;
  const float* xn = x + (int64_t)n * xbs;
  const int iHW = H * W, iDHW = D * H * W;

  int goff[XPER];
#pragma unroll
  for (int i = 0; i < XPER; ++i) {
    const int e = tid + 256 * i;
    int off = -1;
    if (e < XE) {
      const int c = e / CS, r = e - c * CS;
      const int zz = r / PS, r2 = r - zz * PS;
      const int yy = r2 / RS, xx = r2 - yy * RS;
      const int gz = z0 + zz - 1, gy = y0 + yy - 1, gx = x0 + xx - 1;
      if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) off = c * iDHW + gz * iHW + gy * W + gx;
    }
    goff[i] = off;
  }
  f32x16 acc[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  const float* xb = xs + half * CS + (wave * 2) * RS + l32;
  const float* wb = ws + half * (TZ_K * 32) + l32;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * 2;
    __syncthreads();
    {
      float xr[XPER];
      const float* xc = xn + (int64_t)c0 * iDHW;
      const bool full = c0 + 2 <= Cin;
#pragma unroll
      for (int i = 0; i < XPER; ++i) {
        bool ok = goff[i] >= 0;
        if (!full) ok = ok && (c0 + (tid + 256 * i) / CS < Cin);
        xr[i] = xc[ok ? goff[i] : 0];
      }
      f32x4 wr[WPER];
      const f32x4* wsrc = reinterpret_cast<const f32x4*>(wp + (int64_t)c0 * TZ_K * 32);
#pragma unroll
      for (int j = 0; j < WPER; ++j) wr[j] = wsrc[min(tid + 256 * j, WE4 - 1)];
#pragma unroll
      for (int i = 0; i < XPER; ++i) {
        bool ok = goff[i] >= 0;
        if (!full) ok = ok && (c0 + (tid + 256 * i) / CS < Cin);
        if (tid + 256 * i < XE) xs[tid + 256 * i] = ok ? xr[i] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < WPER; ++j)
        if (tid + 256 * j < WE4) reinterpret_cast<f32x4*>(ws)[tid + 256 * j] = wr[j];
    }
    __syncthreads();
#pragma unroll
    for (int dyx = 0; dyx < 9; ++dyx) {
      const int dy = dyx / 3, dx = dyx % 3;
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const float a = wb[(dyx * 10 + u) * 32];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const float b = xb[u * PS + (g + dy) * RS + dx];
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
        }
      }
    }
  }
  const int xg = x0 + l32;
  if (xg >= W) return;
  float* yn = y + (int64_t)n * ybs;
  const float* an = add ? add + (int64_t)n * ybs : nullptr;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int yg = y0 + wave * 2 + g;
    if (yg >= H) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * half;  // row = o*8 + s
      const int o = m >> 3, zg = z0 + (m & 7);
      if (o < Cout && zg < D) {
        const int64_t idx = (int64_t)o * iDHW + (int64_t)zg * iHW + (int64_t)yg * W + xg;
        float v = acc[g][r];
        if (bias) v += bias[o];
        if (an) v += an[idx];
        yn[idx] = v;
      }
    }
  }
}

// ---- Cout <= 4 forward on the vector ALU ----
// With three output channels the MFMA formulations above still spend 4.4x the useful flops (row
// padding x z-Toeplitz inflation).  The packed fp32 FMA (v_pk_fma_f32: two FMAs per lane per issue)
// runs at the same peak rate as the fp32 MFMA and wastes only the 4th channel: each thread owns a run
// of 8 x-adjacent voxels and all (padded to 4) output channels -- 16 float2 accumulators; an input
// row segment of 10 floats is read from LDS once per (channel, dz, dy) and used for 3 taps x 8 voxels
// x 4 channels; the weights of a tap are one uniform 16-byte scalar load (SGPR operands of the FMA,
// the voxel value is broadcast with op_sel).  Tile = 4 z x 8 y x 64 x, 4 input channels per LDS chunk,
// the next chunk prefetched into registers through a buffer descriptor; two workgroups per CU.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int VS_TZ = 4, VS_TY = 8, VS_TX = 64, VS_CC = 4;
constexpr int VS_RS = VS_TX + 4;                 // row: [left halo][64][right halo][2 pad] -> 16-byte aligned segments
constexpr int VS_PS = (VS_TY + 2) * VS_RS, VS_CS = (VS_TZ + 2) * VS_PS;
// wq[(c*27 + tap)*4 + o] = w[o][c][tap] (o >= Cout: 0)
__global__ void pack_w3_valu_kernel(const float* __restrict__ w, float* __restrict__ wq, int Cout, int Cin) {
  const int total = Cin * 27 * 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int o = i & 3, r = i >> 2;
    wq[i] = o < Cout ? w[((int64_t)o * Cin + r / 27) * 27 + r % 27] : 0.f;
  }
}

__global__ __launch_bounds__(256, 2) void conv3_valu_smallcout_kernel(
    const float* __restrict__ x, const float* __restrict__ wq, const float* __restrict__ bias,
    const float* __restrict__ add, float* __restrict__ y, int Cin, int Cout, int D, int H, int W,
    int ty_tiles, int tx_tiles, int64_t xbs, int64_t ybs, int softmax) {
  // softmax: nn.Softmax(dim=1) of the hypothesis (models/modular_unet.py:100) applied in the epilogue -- all
  // (<= 4) channels of a voxel are in this thread's registers, so the logits never travel through HBM.  Same
  // formula and order as softmax_fwd_kernel (max, sum of expf in channel order, one reciprocal): identical bits.
  constexpr int XE = VS_CC * VS_CS, XPER = (XE + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) float xs[XE];
  const int tid = threadIdx.x;
  const int tx = tid & 7, ty = (tid >> 3) & 7, tz = tid >> 6;
  // XCD-aware placement (speed only, a bijection of the grid): blocks b and b + 8 share an XCD and its L2; give each
  // XCD a contiguous run of tiles, so that the 1.9x halo overlap of neighbouring tiles hits in that L2 instead of
  // being fetched from HBM again (PMC: 543 MB per launch for 293 MB of algorithmic bytes before)
  int bt;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int txt = bt % tx_tiles;
  bt /= tx_tiles;
  const int tyt = bt % ty_tiles;
  const int tzt = bt / ty_tiles;
  const int z0 = tzt * VS_TZ, y0 = tyt * VS_TY, x0 = txt * VS_TX;
  const int n = blockIdx.y;
  const float* xn = x + (int64_t)n * xbs;
  const int iHW = H * W, iDHW = D * H * W;

  // chunk-invariant byte offsets of this thread's halo elements (padding / past the tile: out of range)
  unsigned goff[XPER];
#pragma unroll
  for (int i = 0; i < XPER; ++i) {
    const int e = tid + 256 * i;
    const int c = e / VS_CS, r = e - c * VS_CS;
    const int zz = r / VS_PS, r2 = r - zz * VS_PS;
    const int yy = r2 / VS_RS, xx = r2 - yy * VS_RS;
    const int gz = z0 + zz - 1, gy = y0 + yy - 1, gx = x0 + xx - 1;
    const bool ok = e < XE && xx < VS_TX + 2 && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H &&
                    (unsigned)gx < (unsigned)W;
    goff[i] = ok ? (unsigned)(c * iDHW + gz * iHW + gy * W + gx) * 4u : OOB;
  }

  f32x2 acc[8][2];
#pragma unroll
  for (int v = 0; v < 8; ++v) acc[v][0] = acc[v][1] = (f32x2){0.f, 0.f};

  float xr[XPER];
  auto fetch = [&](int c0, bool live) {  // !live: zero-sized descriptor, no memory traffic
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(xn + (int64_t)c0 * iDHW), 0, live ? min(VS_CC, Cin - c0) * iDHW * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < XPER; ++i) xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, goff[i], 0, 0));
  };
  fetch(0, true);
  for (int c0 = 0; c0 < Cin; c0 += VS_CC) {
    __syncthreads();  // the previous chunk has been consumed
#pragma unroll
    for (int i = 0; i < XPER; ++i)
      if (tid + 256 * i < XE) xs[tid + 256 * i] = xr[i];
    __syncthreads();
    fetch(c0 + VS_CC < Cin ? c0 + VS_CC : c0, c0 + VS_CC < Cin);  // next chunk in flight during the FMAs
    const int cn = min(VS_CC, Cin - c0);
    for (int c = 0; c < cn; ++c) {
      const float* wc = wq + (int64_t)(c0 + c) * 27 * 4;
      const float* xc = xs + c * VS_CS + tz * VS_PS + ty * VS_RS + tx * 8;
      // software-pipelined over the 9 (dz, dy) rows: the LDS segment and the 12 scalar weights of row
      // r + 1 are requested before the 48 packed FMAs of row r (sched_barrier: hipcc otherwise issues
      // each scalar load right before its first use and waits on it)
      float seg[2][10];
      f32x4 wrow[2][3];
      auto request = [&](int r, int slot) {  // r is a compile-time constant after unrolling
        const float* row = xc + (r / 3) * VS_PS + (r % 3) * VS_RS;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(row), s1 = *reinterpret_cast<const f32x4*>(row + 4);
        const f32x2 s2 = *reinterpret_cast<const f32x2*>(row + 8);
        seg[slot][0] = s0[0]; seg[slot][1] = s0[1]; seg[slot][2] = s0[2]; seg[slot][3] = s0[3];
        seg[slot][4] = s1[0]; seg[slot][5] = s1[1]; seg[slot][6] = s1[2]; seg[slot][7] = s1[3];
        seg[slot][8] = s2[0]; seg[slot][9] = s2[1];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) wrow[slot][dx] = *reinterpret_cast<const f32x4*>(wc + (r * 3 + dx) * 4);  // uniform
      };
      request(0, 0);
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        if (r + 1 < 9) request(r + 1, (r + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const f32x2 w01 = (f32x2){wrow[r & 1][dx][0], wrow[r & 1][dx][1]};
          const f32x2 w23 = (f32x2){wrow[r & 1][dx][2], wrow[r & 1][dx][3]};
#pragma unroll
          for (int v = 0; v < 8; ++v) {
            const f32x2 xv = (f32x2){seg[r & 1][v + dx], seg[r & 1][v + dx]};
            acc[v][0] = __builtin_elementwise_fma(xv, w01, acc[v][0]);
            acc[v][1] = __builtin_elementwise_fma(xv, w23, acc[v][1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  const int gz = z0 + tz, gy = y0 + ty, gx0 = x0 + tx * 8;
  if (gz >= D || gy >= H || gx0 >= W) return;
  float* yn = y + (int64_t)n * ybs;
  const float* an = add ? add + (int64_t)n * ybs : nullptr;
  const bool vec = gx0 + 8 <= W && ((((uintptr_t)yn) | ((uintptr_t)an)) & 15) == 0 && (W & 3) == 0;
  float val[4][8];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const float bv = (bias && o < Cout) ? bias[o] : 0.f;
    const int64_t idx = (int64_t)o * iDHW + (int64_t)gz * iHW + (int64_t)gy * W + gx0;
#pragma unroll
    for (int k = 0; k < 8; ++k) val[o][k] = acc[k][o >> 1][o & 1] + bv;
    if (an && o < Cout) {
      if (vec) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(an + idx), a1 = *reinterpret_cast<const f32x4*>(an + idx + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { val[o][k] += a0[k]; val[o][4 + k] += a1[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (gx0 + k < W) val[o][k] += an[idx + k];
      }
    }
  }
  if (softmax) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float mx = -INFINITY;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < Cout) mx = fmaxf(mx, val[o][k]);
      float sum = 0.f;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < Cout) sum += expf(val[o][k] - mx);
      const float inv = 1.f / sum;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < Cout) val[o][k] = expf(val[o][k] - mx) * inv;
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    if (o >= Cout) break;
    const int64_t idx = (int64_t)o * iDHW + (int64_t)gz * iHW + (int64_t)gy * W + gx0;
    if (vec) {
      *reinterpret_cast<f32x4*>(yn + idx) = (f32x4){val[o][0], val[o][1], val[o][2], val[o][3]};
      *reinterpret_cast<f32x4*>(yn + idx + 4) = (f32x4){val[o][4], val[o][5], val[o][6], val[o][7]};
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (gx0 + k < W) yn[idx + k] = val[o][k];
    }
  }
}

// y[n,o,s] = bias[o] + add[n,o,s] + sum_ks slab[ks][n,o,s]   (fixed order)
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                     const float* __restrict__ add, float* __restrict__ y, int N,
                                     int Cout, int64_t S, int ksplit, int64_t slab_stride,
                                     int64_t ybs) {
  const int64_t total = (int64_t)N * Cout * S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = i % S;
    const int64_t no = i / S;
    const int o = (int)(no % Cout);
    const int n = (int)(no / Cout);
    float v = slab[i];
    for (int k = 1; k < ksplit; ++k) v += slab[(int64_t)k * slab_stride + i];
    if (bias) v += bias[o];
    const int64_t yi = (int64_t)n * ybs + (int64_t)o * S + s;
    if (add) v += add[yi];
    y[yi] = v;
  }
}

// The same with the statistics partials of the following normalisation (what the unsplit kernels emit from their
// epilogue): grid (voxel chunks, Cout, N), one (sum, sum of squares) slot per block:
// stat[((n * gridDim.x + blockIdx.x) * Cout + o) * 2 + {0,1}] -- the norm then never re-reads y for its statistics.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ slab,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ add, float* __restrict__ y,
                                                                  int Cout, int64_t S, int ksplit, int64_t slab_stride,
                                                                  int64_t ybs, float* __restrict__ stat) {
  __shared__ float scratch[4];
  const int o = blockIdx.y, n = blockIdx.z;
  const float* sp = slab + ((int64_t)n * Cout + o) * S;
  const float b = bias ? bias[o] : 0.f;
  float* yp = y + (int64_t)n * ybs + (int64_t)o * S;
  const float* ap = add ? add + (int64_t)n * ybs + (int64_t)o * S : nullptr;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < S; i += gridDim.x * 256ll) {
    float v = sp[i];
    int k = 1;
    for (; k + 8 <= ksplit; k += 8) {   // eight splits' loads in flight together; summed in split order all the same
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = sp[(int64_t)(k + u) * slab_stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; k < ksplit; ++k) v += sp[(int64_t)k * slab_stride + i];
    v += b;
    if (ap) v += ap[i];
    yp[i] = v;
    s1 += v;
    s2 = fmaf(v, v, s2);
  }
  s1 = block_sum<float, 256>(s1, scratch);
  s2 = block_sum<float, 256>(s2, scratch);
  if (threadIdx.x == 0) {
    float* q = stat + (((int64_t)n * gridDim.x + blockIdx.x) * Cout + o) * 2;
    q[0] = s1;
    q[1] = s2;
  }
}

// ------------------------------------------------------- MFMA bwd-weight kernel
// dW[o, c, tap] = sum_v dy[o, v] * x[c, v + off(tap)]
// MFMA view: i = o (A = dy tile [32 o][256 voxels]), j = c (B = x halo tile
// [32 c][halo]), k = a pair of x-adjacent voxels.  Each wave owns 7 of the 27 taps
// (112 accumulator registers); the A fragment is reused across its 7 MFMAs.
// Channel strides are == 1 (mod 32) so the 32 lanes of a half hit 32 banks.
template <int GX>
struct BwTile {
  static constexpr int TX = GX;
  static constexpr int TY = (GX == 32) ? 4 : 8;
  static constexpr int TZ = (GX == 8) ? 4 : 2;
  static constexpr int NV = TZ * TY * TX;  // 256
  static constexpr int RS = TX + 2;
  static constexpr int PS = (TY + 2) * RS;
  static constexpr int HV = (TZ + 2) * PS;
  static constexpr int CSW = ((HV + 30) / 32) * 32 + 1;  // >= HV, == 1 mod 32
  static constexpr int DSW = NV + 1;
  static constexpr int XROWS = 32 * (TZ + 2) * (TY + 2);
};

template <int GX, bool VEC>
__global__ __launch_bounds__(256, 1) void conv3_mfma_bww_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, int N,
    int Cin, int Cout, int D, int H, int W, int tz_tiles, int ty_tiles, int tx_tiles, int nsplit,
    int64_t xbs, int64_t ybs) {
  using T = BwTile<GX>;
  constexpr int TX = T::TX, TY = T::TY, TZ = T::TZ, RS = T::RS, PS = T::PS,
                CSW = T::CSW, DSW = T::DSW, NV = T::NV;
  static_assert(NV == 256, "one dy voxel per thread");
  __shared__ float xs[32 * CSW];
  __shared__ float ds[32 * DSW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int ctile = blockIdx.x, otile = blockIdx.y, split = blockIdx.z;
  const int c0 = ctile * 32, o0 = otile * 32;
  const int64_t HW = (int64_t)H * W;
  const int64_t DHW = HW * D;

  int toff[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = min(wave * 7 + t, 26);
    toff[t] = (tap / 9) * PS + ((tap / 3) % 3) * RS + (tap % 3);
  }

  f32x16 acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* xb = xs + l32 * CSW + half;
  const float* db = ds + l32 * DSW + half;

  const int tiles_per_n = tz_tiles * ty_tiles * tx_tiles;
  const int ntiles = N * tiles_per_n;
  // this thread's dy voxel inside a tile
  const int vz = tid / (TY * TX), vy = (tid / TX) % TY, vx = tid % TX;

  // The next tile is fetched into registers while the MFMAs of the current one run
  // (1 workgroup per CU: nothing else would hide the global-memory latency).  The halo
  // tile is fetched as aligned float4 interior rows + two scalar halo columns, with 32-bit
  // offsets from a uniform base, to keep the address arithmetic per tile small.
  constexpr int RPC = (TZ + 2) * (TY + 2);   // rows per channel
  constexpr int ROWS = 32 * RPC;
  constexpr int Q = TX / 4;                  // float4 per interior row
  constexpr int NI = ROWS * Q;
  constexpr int IPER = (NI + 255) / 256;
  constexpr int NH = ROWS * 2;
  constexpr int HPER = (NH + 255) / 256;
  f32x4 xi[IPER];
  float xh[HPER];
  float dr[32];
  // validity bits of the prefetched registers (zero padding is applied at commit time; the
  // loads themselves are unconditional from clamped addresses -- see the forward kernel)
  unsigned mi[4], mh, md;
  const int iDHW = (int)DHW, iHW = (int)HW;

  // Tile-invariant part of every prefetch item, computed once: global offset relative to the
  // tile origin, LDS address, and a one-hot code (bit zz | bit 8+yy | bit 20+q) that a
  // per-tile uniform mask of valid rows / columns turns into a 2-instruction bounds check.
  int relI[IPER], ldsI[IPER], relH[HPER], ldsH[HPER];
  unsigned codeI[IPER], codeH[HPER];
#pragma unroll
  for (int k = 0; k < IPER; ++k) {
    const int m = tid + 256 * k;
    const int q = m % Q, row = m / Q;
    const int c = row / RPC, rem = row - c * RPC;
    const int zz = rem / (TY + 2), yy = rem - zz * (TY + 2);
    relI[k] = c * iDHW + zz * iHW + yy * W + 4 * q;
    ldsI[k] = m < NI ? c * CSW + zz * PS + yy * RS + 1 + 4 * q : -1;
    codeI[k] = (m < NI && c0 + c < Cin) ? ((1u << zz) | (1u << (8 + yy)) | (1u << (20 + q))) : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int j = 0; j < HPER; ++j) {
    const int h = tid + 256 * j;
    const int side = h & 1, row = h >> 1;
    const int c = row / RPC, rem = row - c * RPC;
    const int zz = rem / (TY + 2), yy = rem - zz * (TY + 2);
    relH[j] = c * iDHW + zz * iHW + yy * W + (side ? TX : -1);
    ldsH[j] = h < NH ? c * CSW + zz * PS + yy * RS + (side ? TX + 1 : 0) : -1;
    codeH[j] = (h < NH && c0 + c < Cin) ? ((1u << zz) | (1u << (8 + yy)) | (1u << (20 + side))) : 0xFFFFFFFFu;
  }

  auto fetch = [&](int tile) {
    int t = tile;
    const int n = t / tiles_per_n;
    t -= n * tiles_per_n;
    const int txt = t % tx_tiles;
    t /= tx_tiles;
    const int tyt = t % ty_tiles;
    const int tzt = t / ty_tiles;
    const int z0 = tzt * TZ, y0 = tyt * TY, x0 = txt * TX;
    const float* xn = x + (int64_t)n * xbs;
    const float* dn = dy + (int64_t)n * ybs;
    // uniform masks of the halo rows / interior float4s / halo sides that fall inside the volume
    unsigned zy = 0u;
    for (int zz = 0; zz < TZ + 2; ++zz)
      if (z0 + zz - 1 >= 0 && z0 + zz - 1 < D) zy |= 1u << zz;
    for (int yy = 0; yy < TY + 2; ++yy)
      if (y0 + yy - 1 >= 0 && y0 + yy - 1 < H) zy |= 1u << (8 + yy);
    unsigned qm[4] = {0u, 0u, 0u, 0u};
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (x0 + 4 * q + e < W) qm[e] |= 1u << (20 + q);
    const unsigned sm = (x0 - 1 >= 0 ? (1u << 20) : 0u) | (x0 + TX < W ? (1u << 21) : 0u);
    const int base = c0 * iDHW + (z0 - 1) * iHW + (y0 - 1) * W + x0;
    mi[0] = mi[1] = mi[2] = mi[3] = 0u;
    mh = 0u;
#pragma unroll
    for (int k = 0; k < IPER; ++k) {
      if constexpr (VEC) {
        const bool ok = (codeI[k] & ~(zy | qm[0])) == 0u;  // W % 4 == 0: a float4 is in or out as a whole
        xi[k] = *reinterpret_cast<const f32x4*>(xn + (ok ? base + relI[k] : 0));
        const unsigned bit = ok ? (1u << k) : 0u;
        mi[0] |= bit; mi[1] |= bit; mi[2] |= bit; mi[3] |= bit;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (codeI[k] & ~(zy | qm[e])) == 0u;
          xi[k][e] = xn[ok ? base + relI[k] + e : 0];
          mi[e] |= ok ? (1u << k) : 0u;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < HPER; ++j) {
      const bool ok = (codeH[j] & ~(zy | sm)) == 0u;
      xh[j] = xn[ok ? base + relH[j] : 0];
      mh |= ok ? (1u << j) : 0u;
    }
    const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
    const bool vok = gz < D && gy < H && gx < W;
    const int sp = vok ? gz * iHW + gy * W + gx : 0;
    md = 0u;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const bool ok = vok && o0 + j < Cout;
      dr[j] = dn[ok ? (o0 + j) * iDHW + sp : 0];
      md |= ok ? (1u << j) : 0u;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < IPER; ++k) {
      if (ldsI[k] >= 0) {
        float* p = xs + ldsI[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = ((mi[e] >> k) & 1u) ? xi[k][e] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < HPER; ++j)
      if (ldsH[j] >= 0) xs[ldsH[j]] = ((mh >> j) & 1u) ? xh[j] : 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) ds[j * DSW + tid] = ((md >> j) & 1u) ? dr[j] : 0.f;
  };

  if (split < ntiles) {
    fetch(split);
    commit();
  }
  __syncthreads();

  for (int tile = split; tile < ntiles; tile += nsplit) {
    const bool more = tile + nsplit < ntiles;
    if (more) fetch(tile + nsplit);
    for (int z = 0; z < TZ; ++z) {
      for (int yy = 0; yy < TY; ++yy) {
        const float* xr_ = xb + z * PS + yy * RS;
        const float* dr_ = db + (z * TY + yy) * TX;
#pragma unroll
        for (int xp = 0; xp < TX / 2; ++xp) {
          const float a = dr_[2 * xp];
#pragma unroll
          for (int t = 0; t < 7; ++t) {
            const float b = xr_[2 * xp + toff[t]];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();  // every wave is done reading this tile
    if (more) commit();
    __syncthreads();
  }

  // partial dW -> slab[split][Cout][Cin][27]
  float* sl = slab + (int64_t)split * Cout * Cin * 27;
  const int c = c0 + l32;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = wave * 7 + t;
    if (tap < 27 && c < Cin) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o < Cout) sl[((int64_t)o * Cin + c) * 27 + tap] = acc[t][r];
      }
    }
  }
}

// ---- bwd-weight, second generation (used whenever W % 4 == 0 and a sample fits 2 GiB) ----
// Same tiling and MFMA mapping as conv3_mfma_bww_kernel.  What changed is everything AROUND the
// MFMAs: with one wave per SIMD nothing else covers the matrix core while a wave issues the
// prefetch of the next tile (~1000 VALU/VMEM instructions) and commits it to LDS (~1000 more),
// which cost ~20 % of the kernel.  Here
//  * the prefetch uses raw buffer loads: out-of-volume halo elements and channels past Cin/Cout
//    get an out-of-range offset and the hardware returns 0, so there are no validity masks and the
//    commit is a plain sequence of LDS writes;
//  * the row loop is fully unrolled and the prefetch is cut into one slice per row, so its
//    instructions are issued in the shadow of that row's MFMAs (the sched_barrier at the end of
//    each row keeps the slices from being hoisted back into one block).  The prefetch is
//    unconditional (the last iteration re-fetches its own tile): a branch would split the row into
//    basic blocks and undo the interleaving.
// Channel counts that are no multiple of 32 (the reference's real widths are 40 / 80 / 120): a remainder of 1..16
// channels on the o and / or the c side runs on v_mfma_f32_16x16x4_f32 over 16-channel sub-tiles (MT x NT of them;
// the full 32 x 32 pair keeps the 32x32x2 form): a (32 o, 16 c) pair costs half a full pair, (16, 16) a quarter --
// 40 x 40 channels are 2.25 pair-units of MFMA time instead of 4.  K is a quad of x-adjacent voxels; the fragments
// are (channel = lane & 15, voxel = lane >> 4), so the LDS channel strides are == 2 (mod 32) there.
// C/D layout 16x16x4: col (c) = lane & 15, row (o) = 4 * (lane >> 4) + reg.

template <int GX, int MT, int NT>
__device__ __forceinline__ void bww2_body(const float* __restrict__ x, const float* __restrict__ dy,
                                          float* __restrict__ sl, int N, int Cin, int Cout, int D, int H, int W,
                                          int tz_tiles, int ty_tiles, int tx_tiles, int nsplit, int split, int c0,
                                          int o0, int64_t xbs, int64_t ybs, float* __restrict__ xs,
                                          float* __restrict__ ds) {
  using T = BwTile<GX>;
  constexpr bool F = MT == 2 && NT == 2;     // full pair: one 32x32x2 MFMA per (tap, voxel pair)
  constexpr int NCX = 16 * NT, NCO = 16 * MT;
  constexpr int TX = T::TX, TY = T::TY, TZ = T::TZ, RS = T::RS, PS = T::PS, NV = T::NV;
  constexpr int CSW = F ? T::CSW : T::CSW + 1, DSW = F ? T::DSW : T::DSW + 1;
  static_assert(NV == 256, "one dy voxel per thread");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int kq = lane >> 4, l16 = lane & 15;
  const int iHW = H * W, iDHW = D * H * W;

  f32x16 acc[F ? 7 : 1];
  f32x4v acc16[F ? 1 : 7][MT][NT];
  if constexpr (F) {
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  } else {
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc16[t][mt][nt][r] = 0.f;
  }

  // per-wave tap bases into the x halo tile; everything else in an LDS read address is an immediate
  const float* xt[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = min(wave * 7 + t, 26);
    xt[t] = xs + (F ? l32 * CSW + half : l16 * CSW + kq) + (tap / 9) * PS + ((tap / 3) % 3) * RS + (tap % 3);
  }
  const float* db = ds + (F ? l32 * DSW + half : l16 * DSW + kq);

  const int tiles_per_n = tz_tiles * ty_tiles * tx_tiles;
  const int ntiles = N * tiles_per_n;
  const int vz = tid / (TY * TX), vy = (tid / TX) % TY, vx = tid % TX;

  constexpr int RPC = (TZ + 2) * (TY + 2);   // rows per channel
  constexpr int ROWS = NCX * RPC;
  constexpr int Q = TX / 4;                  // float4 per interior row
  constexpr int NI = ROWS * Q, IPER = (NI + 255) / 256;
  constexpr int NH = ROWS * 2, HPER = (NH + 255) / 256;
  constexpr bool IFULL = NI % 256 == 0, HFULL = NH % 256 == 0;   // every thread owns whole prefetch items
  constexpr int NR = TZ * TY;                // MFMA rows per tile == prefetch slices
  constexpr unsigned OOB = 0x80000000u;      // >= num_records of any sample we accept: load returns 0
  f32x4 xi[IPER];
  float xh[HPER];
  float dr[NCO];

  // tile-invariant descriptors (see conv3_mfma_bww_kernel)
  int relI[IPER], ldsI[IPER], relH[HPER], ldsH[HPER];
  unsigned codeI[IPER], codeH[HPER];
#pragma unroll
  for (int k = 0; k < IPER; ++k) {
    const int m = tid + 256 * k;
    const int q = m % Q, row = m / Q;
    const int c = row / RPC, rem = row - c * RPC;
    const int zz = rem / (TY + 2), yy = rem - zz * (TY + 2);
    relI[k] = c * iDHW + zz * iHW + yy * W + 4 * q;
    ldsI[k] = c * CSW + zz * PS + yy * RS + 1 + 4 * q;
    codeI[k] = (IFULL || m < NI) ? (1u << zz) | (1u << (8 + yy)) | (1u << (20 + q)) : 0xffffffffu;  // past the end: never valid
  }
#pragma unroll
  for (int j = 0; j < HPER; ++j) {
    const int h = tid + 256 * j;
    const int side = h & 1, row = h >> 1;
    const int c = row / RPC, rem = row - c * RPC;
    const int zz = rem / (TY + 2), yy = rem - zz * (TY + 2);
    relH[j] = c * iDHW + zz * iHW + yy * W + (side ? TX : -1);
    ldsH[j] = c * CSW + zz * PS + yy * RS + (side ? TX + 1 : 0);
    codeH[j] = (HFULL || h < NH) ? (1u << zz) | (1u << (8 + yy)) | (1u << (20 + side)) : 0xffffffffu;
  }

  // per-tile uniform state of the prefetch
  __amdgpu_buffer_rsrc_t rx, rd;
  unsigned inI = 0u, inH = 0u;  // valid halo rows | interior float4s (resp. halo sides)
  int base = 0, dsp = 0;
  bool vok = false;
  auto tile_setup = [&](int tile, bool live) {  // !live: zero-sized descriptors, no memory traffic
    int t = tile;
    const int n = t / tiles_per_n;
    t -= n * tiles_per_n;
    const int txt = t % tx_tiles;
    t /= tx_tiles;
    const int tyt = t % ty_tiles;
    const int tzt = t / ty_tiles;
    const int z0 = tzt * TZ, y0 = tyt * TY, x0 = txt * TX;
    // channels past Cin / Cout are past num_records: zero-filled by the hardware as well
    rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)n * xbs), 0, live ? Cin * iDHW * 4 : 0, 0x00020000);
    rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (int64_t)n * ybs), 0, live ? Cout * iDHW * 4 : 0, 0x00020000);
    unsigned zy = 0u;
    for (int zz = 0; zz < TZ + 2; ++zz)
      if (z0 + zz - 1 >= 0 && z0 + zz - 1 < D) zy |= 1u << zz;
    for (int yy = 0; yy < TY + 2; ++yy)
      if (y0 + yy - 1 >= 0 && y0 + yy - 1 < H) zy |= 1u << (8 + yy);
    unsigned qm = 0u;
    for (int q = 0; q < Q; ++q)
      if (x0 + 4 * q < W) qm |= 1u << (20 + q);  // W % 4 == 0: a float4 is in or out as a whole
    inI = zy | qm;
    inH = zy | (x0 - 1 >= 0 ? (1u << 20) : 0u) | (x0 + TX < W ? (1u << 21) : 0u);
    base = c0 * iDHW + (z0 - 1) * iHW + (y0 - 1) * W + x0;
    const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
    vok = gz < D && gy < H && gx < W;
    dsp = o0 * iDHW + gz * iHW + gy * W + gx;
  };
  // slice r of the prefetch (r is a compile-time constant wherever this is called)
  auto fetch_slice = [&](int r) {
#pragma unroll
    for (int k = 0; k < IPER; ++k)
      if (k % NR == r) {
        const bool ok = (codeI[k] & ~inI) == 0u;
        xi[k] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(base + relI[k]) * 4u : OOB, 0, 0));
      }
#pragma unroll
    for (int j = 0; j < HPER; ++j)
      if (j % NR == r) {
        const bool ok = (codeH[j] & ~inH) == 0u;
        xh[j] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (unsigned)(base + relH[j]) * 4u : OOB, 0, 0));
      }
#pragma unroll
    for (int j = 0; j < NCO; ++j)
      if (j % NR == r)
        dr[j] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rd, vok ? (unsigned)(dsp + j * iDHW) * 4u : OOB, 0, 0));
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < IPER; ++k)
      if (IFULL || tid + 256 * k < NI) {
        float* p = xs + ldsI[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = xi[k][e];
      }
#pragma unroll
    for (int j = 0; j < HPER; ++j)
      if (HFULL || tid + 256 * j < NH) xs[ldsH[j]] = xh[j];
#pragma unroll
    for (int j = 0; j < NCO; ++j) ds[j * DSW + tid] = dr[j];
  };

  if (split < ntiles) {
    tile_setup(split, true);
#pragma unroll
    for (int r = 0; r < NR; ++r) fetch_slice(r);
    commit();
  }
  __syncthreads();

  for (int tile = split; tile < ntiles; tile += nsplit) {
    const bool more = tile + nsplit < ntiles;
    tile_setup(more ? tile + nsplit : tile, more);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      fetch_slice(r);
      const int ro = (r / TY) * PS + (r % TY) * RS;  // immediate
      if constexpr (F) {
#pragma unroll
        for (int xp = 0; xp < TX / 2; ++xp) {
          const float a = db[r * TX + 2 * xp];
#pragma unroll
          for (int t = 0; t < 7; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xt[t][ro + 2 * xp], acc[t], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int xq = 0; xq < TX / 4; ++xq) {
          float a[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[mt] = db[mt * 16 * DSW + r * TX + 4 * xq];
#pragma unroll
          for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const float b = xt[t][nt * 16 * CSW + ro + 4 * xq];
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                acc16[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b, acc16[t][mt][nt], 0, 0, 0);
            }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // every wave is done reading this tile
    if (more) commit();
    __syncthreads();
  }

  // partial dW -> slab[split][27][Cout][Cin]: the lane index is the input channel, so every store
  // writes 32 (16) consecutive floats (the [o][c][27] order of dW would scatter each lane to its own
  // cache line: 28k line requests per workgroup instead of ~900)
  if constexpr (F) {
    const int c = c0 + l32;
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int tap = wave * 7 + t;
      if (tap < 27 && c < Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (o < Cout) sl[((int64_t)tap * Cout + o) * Cin + c] = acc[t][r];
        }
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int tap = wave * 7 + t;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int c = c0 + 16 * nt + l16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = o0 + 16 * mt + 4 * kq + r;
            if (tap < 27 && c < Cin && o < Cout) sl[((int64_t)tap * Cout + o) * Cin + c] = acc16[t][mt][nt][r];
          }
        }
    }
  }
}

template <int GX>
__global__ __launch_bounds__(256, 1) void conv3_mfma_bww2_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, int N,
    int Cin, int Cout, int D, int H, int W, int tz_tiles, int ty_tiles, int tx_tiles, int nsplit,
    int ctiles, int otiles, int64_t xbs, int64_t ybs) {
  using T = BwTile<GX>;
  __shared__ float xs[32 * T::CSW];
  __shared__ float ds[32 * T::DSW];
  // XCD-aware placement (speed only): blocks b and b + 8 are observed to share an XCD and its L2, so
  // the block id is swizzled to give each XCD a contiguous run of virtual ids, and the virtual id is
  // decoded with the (c-tile, o-tile) pair fastest: the workgroups on one L2 work on the same and on
  // neighbouring voxel tiles, sharing the dy tile, the x tile and the halos.  A bijection of the
  // grid -- which partial sums land in which slab does not change, so results stay bit-identical.
  int vid;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int pairs = ctiles * otiles;
  const int pair = vid % pairs, split = vid / pairs;
  const int ctile = pair % ctiles, otile = pair / ctiles;
  bww2_body<GX, 2, 2>(x, dy, slab + (int64_t)split * 27 * Cout * Cin, N, Cin, Cout, D, H, W, tz_tiles, ty_tiles, tx_tiles,
                      nsplit, split, ctile * 32, otile * 32, xbs, ybs, xs, ds);
}

// The same, with the pairs of a channel remainder on their 16-row sub-tiles: ONE launch holds all four pair
// classes, each class cut into a number of voxel-range splits proportional to its MFMA cost, so that every
// workgroup of the launch lasts about equally long (the grid still fills the chip in whole residencies).
template <int GX>
__global__ __launch_bounds__(256, 1) void conv3_mfma_bww2c_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, int N,
    int Cin, int Cout, int D, int H, int W, int tz_tiles, int ty_tiles, int tx_tiles, BwwClasses k,
    int64_t xbs, int64_t ybs) {
  using T = BwTile<GX>;
  __shared__ float xs[32 * (T::CSW + 1)];
  __shared__ float ds[32 * (T::DSW + 1)];
  int vid;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int cls = vid >= k.start[3] ? 3 : (vid >= k.start[2] ? 2 : (vid >= k.start[1] ? 1 : 0));
  const int idx = vid - k.start[cls];
  const int npairs = cls == 0 ? k.of * k.cf : (cls == 1 ? k.of : (cls == 2 ? k.cf : 1));
  const int pair = idx % npairs, split = idx / npairs;
  float* sl = slab + (int64_t)split * 27 * Cout * Cin;
  const int ns = k.ns[cls];
  if (cls == 0) {
    bww2_body<GX, 2, 2>(x, dy, sl, N, Cin, Cout, D, H, W, tz_tiles, ty_tiles, tx_tiles, ns, split, (pair % k.cf) * 32,
                        (pair / k.cf) * 32, xbs, ybs, xs, ds);
  } else if (cls == 1) {
    bww2_body<GX, 2, 1>(x, dy, sl, N, Cin, Cout, D, H, W, tz_tiles, ty_tiles, tx_tiles, ns, split, k.cf * 32, pair * 32,
                        xbs, ybs, xs, ds);
  } else if (cls == 2) {
    bww2_body<GX, 1, 2>(x, dy, sl, N, Cin, Cout, D, H, W, tz_tiles, ty_tiles, tx_tiles, ns, split, pair * 32, k.of * 32,
                        xbs, ybs, xs, ds);
  } else {
    bww2_body<GX, 1, 1>(x, dy, sl, N, Cin, Cout, D, H, W, tz_tiles, ty_tiles, tx_tiles, ns, split, k.cf * 32, k.of * 32,
                        xbs, ybs, xs, ds);
  }
}

// dW[o][c][27] = sum over splits of slab[split][27][o][c] (fixed order): one block per (o, 32 input
// channels); coalesced reads along c, transposed through LDS, one contiguous 864-float write.
__global__ __launch_bounds__(256) void slab_reduce_t_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                            int Cin, int Cout, BwwClasses k, float scale, int* __restrict__ oflag) {
  __shared__ float tr[32 * 27 + 32];
  const int o = blockIdx.x, c0 = blockIdx.y * 32;
  const int nsplit = k.ns[(o >= k.of * 32 ? 2 : 0) + ((int)blockIdx.y >= k.cf ? 1 : 0)];  // splits of this pair's class
  const int64_t plane = (int64_t)Cout * Cin, split_stride = 27 * plane;
  const int cw = min(32, Cin - c0);
  for (int e = threadIdx.x; e < 27 * 32; e += 256) {
    const int tap = e >> 5, cl = e & 31;
    float v = 0.f;
    if (cl < cw) {
      const float* p = slab + (int64_t)tap * plane + (int64_t)o * Cin + c0 + cl;
      int s = 0;
      for (; s + 16 <= nsplit; s += 16) {  // 16 independent loads in flight, summed in split order
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = p[(int64_t)(s + u) * split_stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) v += t[u];
      }
      for (; s + 4 <= nsplit; s += 4) {
        const float t0 = p[(int64_t)s * split_stride], t1 = p[(int64_t)(s + 1) * split_stride],
                    t2 = p[(int64_t)(s + 2) * split_stride], t3 = p[(int64_t)(s + 3) * split_stride];
        v += t0; v += t1; v += t2; v += t3;
      }
      for (; s < nsplit; ++s) v += p[(int64_t)s * split_stride];
    }
    tr[cl * 27 + tap] = v * scale;   // (scale != 1 only in the fp16 training flow: the loss scale leaves here)
    report_nonfinite(v * scale, oflag);
  }
  __syncthreads();
  float* dst = out + ((int64_t)o * Cin + c0) * 27;
  for (int e = threadIdx.x; e < cw * 27; e += 256) dst[e] = tr[e];
}

// The same reduction for FEW (o, c-tile) pairs with MANY splits (32 -> 32 at 128^3: 32 blocks each walking 256 slabs,
// 22 us on 32 of 256 CUs): one block per (o, c-tile, tap); its 256 threads are 32 channels x 8 split lanes, lane j sums
// the splits s = j, j + 8, ... and the eight partial sums are added in lane order -- fixed order, bit-reproducible.
__global__ __launch_bounds__(256) void slab_reduce_tap_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                              int Cin, int Cout, BwwClasses k, float scale, int* __restrict__ oflag) {
  __shared__ float part[8][32];
  const int o = blockIdx.x, ct = blockIdx.y, tap = blockIdx.z;
  const int nsplit = k.ns[(o >= k.of * 32 ? 2 : 0) + (ct >= k.cf ? 1 : 0)];
  const int64_t plane = (int64_t)Cout * Cin, split_stride = 27 * plane;
  const int cl = threadIdx.x & 31, j = threadIdx.x >> 5;
  const int c = ct * 32 + cl;
  float v = 0.f;
  if (c < Cin) {
    const float* p = slab + (int64_t)tap * plane + (int64_t)o * Cin + c;
    int s = j;
    for (; s + 24 < nsplit; s += 32) {   // four loads in flight
      const float t0 = p[(int64_t)s * split_stride], t1 = p[(int64_t)(s + 8) * split_stride],
                  t2 = p[(int64_t)(s + 16) * split_stride], t3 = p[(int64_t)(s + 24) * split_stride];
      v += t0; v += t1; v += t2; v += t3;
    }
    for (; s < nsplit; s += 8) v += p[(int64_t)s * split_stride];
  }
  part[j][cl] = v;
  __syncthreads();
  if (j == 0 && c < Cin) {
    float t = part[0][cl];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += part[q][cl];
    out[((int64_t)o * Cin + c) * 27 + tap] = t * scale;
    report_nonfinite(t * scale, oflag);
  }
}

// picks the reduction by shape: per-tap blocks where the (o, c-tile) grid alone cannot fill the chip
static void launch_slab_reduce_t(const float* slab, float* dw, int Cin, int Cout, int ctiles, const BwwClasses& k,
                                 float scale, hipStream_t st) {
  int max_ns = 1;
  for (int c = 0; c < 4; ++c) max_ns = std::max(max_ns, k.ns[c]);
  if ((int64_t)Cout * ctiles < 2 * (int64_t)num_cus() && max_ns >= 16)
    hipLaunchKernelGGL(slab_reduce_tap_kernel, dim3((unsigned)Cout, (unsigned)ctiles, 27u), dim3(256), 0, st, slab, dw, Cin,
                       Cout, k, scale, scale != 1.f ? overflow_flag() : nullptr);
  else
    hipLaunchKernelGGL(slab_reduce_t_kernel, dim3((unsigned)Cout, (unsigned)ctiles), dim3(256), 0, st, slab, dw, Cin, Cout,
                       k, scale, scale != 1.f ? overflow_flag() : nullptr);
}

// ------------------------------------------- bwd-weight, tiny channel count on one side
// dW[o,c,tap] when Cin <= 4 (first conv) or Cout <= 4 (out conv).  The generic kernel would
// pad the narrow side to 32 MFMA rows/cols (8-10x waste).  Here the narrow channel AND the
// tap share the lane index: j = (s = l32>>3, t8 = l32&7), tap = wave*8 + t8 (27 of 32 used),
// so ONE MFMA per k-step per wave covers all taps:
//    G[i, (s,tap)] = sum_v P[i, v] * Q[s, v + off(tap)]
//  swap=0 (Cin small):  P = dy (wide = Cout), Q = x,  dW[o=i][c=s][tap]
//  swap=1 (Cout small): P = x (wide = Cin),  Q = dy read at v - off(tap) = v + off(26-tap),
//                       dW[o=s][c=i][tap]
// 16 accumulators and 46 KB of LDS -> three workgroups per CU; the next tile is prefetched into
// registers through buffer descriptors (out-of-volume halo and channels past CP / CQ read as 0)
// while the current one is multiplied.
template <int GX>
__global__ __launch_bounds__(256) void conv3_mfma_bww_small_kernel(
    const float* __restrict__ P, const float* __restrict__ Q, float* __restrict__ slab, int N,
    int CP, int CQ, int D, int H, int W, int tz_tiles, int ty_tiles, int tx_tiles, int nsplit,
    int64_t pbs, int64_t qbs, int swap, int Cin, int Cout) {
  using T = BwTile<GX>;
  constexpr int TX = T::TX, TY = T::TY, TZ = T::TZ, RS = T::RS, PS = T::PS, HV = T::HV,
                DSW = T::DSW;
  static_assert(T::NV == 256, "one wide-tile voxel per thread");
  constexpr int QS = HV + 1;
  constexpr int QE = 4 * HV, QPER = (QE + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ float ps[32 * DSW];
  __shared__ float qs[4 * QS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int p0 = blockIdx.x * 32, split = blockIdx.y;
  const int iHW = H * W, iDHW = D * H * W;

  const int s_l = l32 >> 3, t8 = l32 & 7;
  const int tap_raw = wave * 8 + t8;
  const int tap_eff = min(swap ? 26 - tap_raw : tap_raw, 26);
  const int toff = (max(tap_eff, 0) / 9) * PS + ((max(tap_eff, 0) / 3) % 3) * RS + (max(tap_eff, 0) % 3);
  const float* qb = qs + s_l * QS + toff + half;
  const float* pb = ps + l32 * DSW + half;

  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;

  const int vz = tid / (TY * TX), vy = (tid / TX) % TY, vx = tid % TX;
  const int tiles_per_n = tz_tiles * ty_tiles * tx_tiles;
  const int ntiles = N * tiles_per_n;

  // tile-invariant description of this thread's narrow-halo elements: offset from the halo origin,
  // LDS slot, halo coordinates (zz | yy << 8 | xx << 16; 0xFFFFFF past the tile: never valid)
  int qrel[QPER], qlds[QPER];
  unsigned qcode[QPER];
#pragma unroll
  for (int k = 0; k < QPER; ++k) {
    const int e = tid + 256 * k;
    const int c = e / HV, r = e - c * HV;
    const int zz = r / PS, r2 = r - zz * PS;
    const int yy = r2 / RS, xx = r2 - yy * RS;
    qrel[k] = c * iDHW + zz * iHW + yy * W + xx;
    qlds[k] = e < QE ? c * QS + r : -1;
    qcode[k] = e < QE ? ((unsigned)zz | ((unsigned)yy << 8) | ((unsigned)xx << 16)) : 0x00FFFFFFu;
  }

  float pr[32], qr[QPER];
  auto fetch = [&](int tile, bool live) {  // !live: zero-sized descriptors, no memory traffic
    int t = tile;
    const int n = t / tiles_per_n;
    t -= n * tiles_per_n;
    const int txt = t % tx_tiles;
    t /= tx_tiles;
    const int tyt = t % ty_tiles;
    const int tzt = t / ty_tiles;
    const int z0 = tzt * TZ, y0 = tyt * TY, x0 = txt * TX;
    const __amdgpu_buffer_rsrc_t rp =
        __builtin_amdgcn_make_buffer_rsrc((void*)(P + (int64_t)n * pbs), 0, live ? CP * iDHW * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq =
        __builtin_amdgcn_make_buffer_rsrc((void*)(Q + (int64_t)n * qbs), 0, live ? CQ * iDHW * 4 : 0, 0x00020000);
    const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
    const bool vok = gz < D && gy < H && gx < W;
    const int sp = p0 * iDHW + gz * iHW + gy * W + gx;
#pragma unroll
    for (int j = 0; j < 32; ++j)
      pr[j] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rp, vok ? (unsigned)(sp + j * iDHW) * 4u : OOB, 0, 0));
    const int qbase = (z0 - 1) * iHW + (y0 - 1) * W + (x0 - 1);
#pragma unroll
    for (int k = 0; k < QPER; ++k) {
      const unsigned cd = qcode[k];
      const bool ok = (unsigned)(z0 - 1 + (int)(cd & 0xFFu)) < (unsigned)D &&
                      (unsigned)(y0 - 1 + (int)((cd >> 8) & 0xFFu)) < (unsigned)H &&
                      (unsigned)(x0 - 1 + (int)(cd >> 16)) < (unsigned)W;
      qr[k] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rq, ok ? (unsigned)(qbase + qrel[k]) * 4u : OOB, 0, 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < 32; ++j) ps[j * DSW + tid] = pr[j];
#pragma unroll
    for (int k = 0; k < QPER; ++k)
      if (qlds[k] >= 0) qs[qlds[k]] = qr[k];
  };

  if (split < ntiles) fetch(split, true);
  for (int tile = split; tile < ntiles; tile += nsplit) {
    __syncthreads();  // every wave is done reading the previous tile
    commit();
    __syncthreads();
    const bool more = tile + nsplit < ntiles;
    fetch(more ? tile + nsplit : tile, more);  // in flight during the MFMAs below
    for (int z = 0; z < TZ; ++z)
      for (int yy = 0; yy < TY; ++yy) {
        const float* prow = pb + (z * TY + yy) * TX;
        const float* qrow = qb + z * PS + yy * RS;
        // two independent accumulation chains (even / odd voxel pairs): a single chain makes every MFMA
        // wait for the previous one's result
#pragma unroll
        for (int xp = 0; xp < TX / 2; xp += 2) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(prow[2 * xp], qrow[2 * xp], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(prow[2 * xp + 2], qrow[2 * xp + 2], acc2, 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
  float* sl = slab + (int64_t)split * Cout * Cin * 27;
  if (tap_raw < 27 && s_l < CQ) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = p0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i < CP) {
        const int o = swap ? s_l : i, c = swap ? i : s_l;
        sl[((int64_t)o * Cin + c) * 27 + tap_raw] = acc[r];
      }
    }
  }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                   int64_t total, int nsplit) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    // fixed summation order; the loads of 8 splits are independent and stay in flight together
    float v = 0.f;
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = slab[(int64_t)(s + u) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < nsplit; ++s) v += slab[(int64_t)s * total + i];
    out[i] = v;
  }
}

// dbias[o] = sum_{n,s} dy[n,o,s]: grid (chunks, Cout) of partial sums, then a fixed-order
// finalize (deterministic).  Short fp32 runs per thread, double across threads / blocks.  Chunks of 8192 voxels
// (float4 loads, 8 per thread and sample): the 32768-voxel chunks this started with gave the small volumes of the
// reference's dmri_hippo net (40 channels x 48x88x24) 160 workgroups of scalar loads -- 303 us for a 130 MB tensor.
template <bool VEC>
__global__ __launch_bounds__(256) void dbias_partial_kernel(const float* __restrict__ dy,
                                                            double* __restrict__ partial, int N,
                                                            int64_t S, int64_t ybs, int nblk) {
  __shared__ double scratch[4];
  const int b = blockIdx.x, o = blockIdx.y;
  const int64_t begin = (int64_t)b * DBIAS_CHUNK, end = min(S, begin + DBIAS_CHUNK);
  double acc = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* p = dy + (int64_t)n * ybs + (int64_t)o * S;
    float part = 0.f;
    if (VEC) {   // S, ybs multiples of 4 and dy 16-byte aligned: every chunk starts on a float4
      for (int64_t s = begin + threadIdx.x * 4; s < end; s += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(p + s);
        part += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int64_t s = begin + threadIdx.x; s < end; s += 256) part += p[s];
    }
    acc += part;
  }
  const double tot = block_sum<double, 256>(acc, scratch);
  if (threadIdx.x == 0) partial[(int64_t)o * nblk + b] = tot;
}
__global__ void dbias_finalize_kernel(const double* __restrict__ partial, float* __restrict__ dbias,
                                      int Cout, int nblk) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= Cout) return;
  double acc = 0.0;
  for (int b = 0; b < nblk; ++b) acc += partial[(int64_t)o * nblk + b];
  dbias[o] = (float)acc;
}

// ------------------------------------------------------ generic direct kernels
// Any cubic k / stride / pad; one thread per output element.  Fallback path.
__global__ void conv3d_direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                         const float* __restrict__ bias,
                                         const float* __restrict__ add, float* __restrict__ y,
                                         int N, int Cin, int Cout, int D, int H, int W, int OD,
                                         int OH, int OW, int k, int stride, int pad, int64_t xbs,
                                         int64_t ybs) {
  const int64_t OS = (int64_t)OD * OH * OW;
  const int64_t total = (int64_t)N * Cout * OS;
  const int k3 = k * k * k;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    int64_t r = i / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int oz = (int)(r % OD);
    r /= OD;
    const int o = (int)(r % Cout);
    const int n = (int)(r / Cout);
    const float* xn = x + (int64_t)n * xbs;
    float acc = 0.f;
    for (int c = 0; c < Cin; ++c) {
      const float* wc = w + ((int64_t)o * Cin + c) * k3;
      const float* xc = xn + (int64_t)c * D * H * W;
      for (int dz = 0; dz < k; ++dz) {
        const int iz = oz * stride + dz - pad;
        if (iz < 0 || iz >= D) continue;
        for (int dy = 0; dy < k; ++dy) {
          const int iy = oy * stride + dy - pad;
          if (iy < 0 || iy >= H) continue;
          for (int dx = 0; dx < k; ++dx) {
            const int ix = ox * stride + dx - pad;
            if (ix < 0 || ix >= W) continue;
            acc = fmaf(wc[(dz * k + dy) * k + dx], xc[((int64_t)iz * H + iy) * W + ix], acc);
          }
        }
      }
    }
    if (bias) acc += bias[o];
    const int64_t yi = (int64_t)n * ybs + (int64_t)o * OS + ((int64_t)oz * OH + oy) * OW + ox;
    if (add) acc += add[yi];
    y[yi] = acc;
  }
}

// dx[n,c,iz,iy,ix] = sum_{o,taps: (i + pad - d) % stride == 0} w[o,c,d] * dy[n,o,(i+pad-d)/stride]
__global__ void conv3d_direct_bwd_data_kernel(const float* __restrict__ dy,
                                              const float* __restrict__ w, float* __restrict__ dx,
                                              int N, int Cin, int Cout, int D, int H, int W,
                                              int OD, int OH, int OW, int k, int stride, int pad,
                                              int64_t xbs, int64_t ybs) {
  const int64_t S = (int64_t)D * H * W;
  const int64_t OS = (int64_t)OD * OH * OW;
  const int64_t total = (int64_t)N * Cin * S;
  const int k3 = k * k * k;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W);
    int64_t r = i / W;
    const int iy = (int)(r % H);
    r /= H;
    const int iz = (int)(r % D);
    r /= D;
    const int c = (int)(r % Cin);
    const int n = (int)(r / Cin);
    const float* dyn = dy + (int64_t)n * ybs;
    float acc = 0.f;
    for (int o = 0; o < Cout; ++o) {
      const float* wc = w + ((int64_t)o * Cin + c) * k3;
      const float* dyo = dyn + (int64_t)o * OS;
      for (int dz = 0; dz < k; ++dz) {
        const int tz = iz + pad - dz;
        if (tz < 0 || tz % stride) continue;
        const int oz = tz / stride;
        if (oz >= OD) continue;
        for (int dyy = 0; dyy < k; ++dyy) {
          const int ty = iy + pad - dyy;
          if (ty < 0 || ty % stride) continue;
          const int oy = ty / stride;
          if (oy >= OH) continue;
          for (int dxx = 0; dxx < k; ++dxx) {
            const int tx = ix + pad - dxx;
            if (tx < 0 || tx % stride) continue;
            const int ox = tx / stride;
            if (ox >= OW) continue;
            acc = fmaf(wc[(dz * k + dyy) * k + dxx], dyo[((int64_t)oz * OH + oy) * OW + ox], acc);
          }
        }
      }
    }
    dx[(int64_t)n * xbs + (int64_t)c * S + ((int64_t)iz * H + iy) * W + ix] = acc;
  }
}

// dw[o,c,d] = sum_{n,ov} dy[n,o,ov] * x[n,c,ov*stride + d - pad]; one block per (o,c,tap).
__global__ __launch_bounds__(256) void conv3d_direct_bwd_weight_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N,
    int Cin, int Cout, int D, int H, int W, int OD, int OH, int OW, int k, int stride, int pad,
    int64_t xbs, int64_t ybs) {
  __shared__ double scratch[4];
  const int k3 = k * k * k;
  int b = blockIdx.x;
  const int tap = b % k3;
  b /= k3;
  const int c = b % Cin;
  const int o = b / Cin;
  const int dz = tap / (k * k), dyy = (tap / k) % k, dxx = tap % k;
  const int64_t OS = (int64_t)OD * OH * OW;
  double acc = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* xc = x + (int64_t)n * xbs + (int64_t)c * D * H * W;
    const float* dyo = dy + (int64_t)n * ybs + (int64_t)o * OS;
    float part = 0.f;
    int cnt = 0;
    for (int64_t s = threadIdx.x; s < OS; s += 256) {
      const int ox = (int)(s % OW);
      const int oy = (int)((s / OW) % OH);
      const int oz = (int)(s / ((int64_t)OW * OH));
      const int iz = oz * stride + dz - pad, iy = oy * stride + dyy - pad,
                ix = ox * stride + dxx - pad;
      if (iz >= 0 && iz < D && iy >= 0 && iy < H && ix >= 0 && ix < W)
        part = fmaf(dyo[s], xc[((int64_t)iz * H + iy) * W + ix], part);
      if (++cnt == 64) { acc += part; part = 0.f; cnt = 0; }
    }
    acc += part;
  }
  const double tot = block_sum<double, 256>(acc, scratch);
  if (threadIdx.x == 0) dw[((int64_t)o * Cin + c) * k3 + tap] = (float)tot;
}

// ------------------------------------------------------------------ planning
// Plan for a 3x3x3/s1/p1 conv with K-channels `kin` and M-channels `mout`.
// Lanes along x per 32-voxel group: the widest of {32, 16, 8} unless a narrower one wastes noticeably
// fewer padded voxels (W = 24: 16 -> 2 tiles = 32 columns, 8 -> 3 tiles = 24 columns).
int pick_gx(int W) {
  int best = 8;
  int64_t best_pad = round_up(W, 8);
  for (int gx : {16, 32}) {
    const int64_t pad = round_up(W, gx);
    if (W >= gx && pad * 100 <= best_pad * 108) {  // prefer the wider tile unless it pads > 8 % more
      best = gx;
      best_pad = std::min(best_pad, pad);
    }
  }
  return best;
}

// M355_COMPUTE_F32X3 (conv3d_f32x3.hip): a 32-row tile must carry real rows, and the 8-channel slab of a sample must fit
// the 31-bit byte offsets its loads add up.  Layers with 3..7 K-channels (4 -> 32 forward, 3 -> 32 data gradient @128^3:
// one chunk, 4 / 3 of its 8 channels real) run 0.183 / 0.174 ms on the split kernel against 0.21 / 0.19 on the fp32 MFMA
// (0.06 ms of that is the 268 MB they write, the rest the half-empty K of their MFMAs) -- behind M355_F32X3_EDGE=1, off
// by default: with the FIRST layer of the net on the split kernel one voxel of the 2.1 M of the bench volume (a near-tie
// of two class probabilities) takes the other side of the CPU reference's argmax; with it on the fp32 MFMA none does.
static bool x3_layer(int kin, int mout, int D, int H, int W) {
  return tuning().f32x3 && kin >= (tuning().f32x3_edge ? 3 : 8) && mout > 4 && (int64_t)D * H * W < (1ll << 26);
}

FwdPlan plan_mfma(int N, int kin, int mout, int D, int H, int W, int compute) {
  FwdPlan p{};
  p.mfma = true;
  p.gx = pick_gx(W);
  const int gy = 32 / p.gx;
  const bool h16 = is16(compute);  // bf16 / fp16 operand modes share one plan
  if (compute == M355_COMPUTE_F32 && tuning().f32x3 == 2) compute = M355_COMPUTE_F32X3;   // M355_F32X3=2: test hook
  const bool x3 = compute == M355_COMPUTE_F32X3 && x3_layer(kin, mout, D, H, W);
  if (!h16 && !x3) compute = M355_COMPUTE_F32;
  p.x3 = x3 ? 1 : 0;
  const int cc = h16 ? 16 : (x3 ? 8 : 4);  // input channels per LDS chunk
  p.kin_pad = (int)round_up(kin, cc);
  p.mout_pad = (int)round_up(mout, 32);
  p.otiles = p.mout_pad / 32;
  // fp32: a remainder of 1..16 channels runs as ONE 16-row tile on v_mfma_f32_16x16x4_f32 (half the MFMA time of a
  // padded 32-row tile): 40 channels = 32 + 16 rows instead of 64, 80 = 64 + 16 instead of 96
  p.tile16 = (!h16 && tuning().tile16 && mout % 32 >= 1 && mout % 32 <= 16) ? 1 : 0;   // (split kernels: conv3_f32x3_m16_kernel)
  if (p.tile16) p.otiles -= 1;
  const int wtiles = p.otiles + p.tile16;   // workgroup items per spatial tile
  p.nchunks = p.kin_pad / cc;
  p.nw = 4;
  p.tz_tiles = (int)ceil_div(D, 4);
  p.tx_tiles = (int)ceil_div(W, p.gx);
  // Pick (voxel-tile height NTW, split-K) by a cost model instead of "fill the chip once":
  // workgroups of one launch do equal work, so the time is rounds x (workgroups sharing a CU) x
  // time of one workgroup, and a launch that needs 1.1 rounds costs as much as one that needs 2.
  //   slots     NTW <= 4: 66.8 KB LDS -> two workgroups per CU (512); NTW = 8: one (256)
  //   one chunk 54 x NTW MFMAs of 64 cycles per wave at ~2.04 GHz; + fill/epilogue (see `fixed`)
  //   split-K   ks x out bytes written + read again by the reduce kernel (~4 TB/s) + a launch
  const int64_t out_bytes = (int64_t)N * mout * D * H * W * 4;
  const int force_ntw = tuning().conv_ntw;
  const int force_ks = tuning().conv_ksplit;
  const int cands[4] = {4, 8, 2, 1};
  int chosen = 1, chosen_ks = 1;
  double best = 1e30;
  const int cus = num_cus();
  const bool h16_one = h16 && tuning().h16_oneshot && tuning().h16_oneshot != 3;
  if (h16_one) {
    // 16-bit kernels, one item per workgroup (conv3_h16_kernel, ONE): cost model over (tile height, split-K).
    //   time ~ residencies x (chunks per item x chunk time(NTW) x share + fixed(NTW)) + split-K reduction
    // chunk time per workgroup with two resident per CU (measured: ~44 % of the MFMA rate at NTW = 4; narrower tiles
    // re-read the weights more often), `share` < 1 when the launch leaves CUs with a single workgroup, the reduction
    // pass ~12 us + its slab traffic.  Constants fitted on the cfg2 layers (tools/plan_sweep_h16.py).
    double best_h = 1e30;
    for (int ntw : {4, 2, 1}) {
      if (p.gx == 8 && ntw == 4) continue;                 // not instantiated
      if (force_ntw && ntw != force_ntw && force_ntw != 8 && !(p.gx == 8 && force_ntw == 4)) continue;
      const int ty = ntw * gy;
      if (ty > H && ntw > 1 && !force_ntw) continue;
      const double chunk_us = ntw == 4 ? 5.8 : (ntw == 2 ? 3.5 : 2.8), fixed_us = ntw == 4 ? 6.0 : (ntw == 2 ? 3.5 : 2.5);
      const int64_t nwg1 = (int64_t)p.tz_tiles * ceil_div(H, ty) * p.tx_tiles * p.otiles * N;
      for (int ks = 1; ks <= std::min(p.nchunks, 8); ++ks) {
        if (ks > 1 && (ks - 1) * ceil_div(p.nchunks, ks) >= p.nchunks) continue;   // an empty split
        if (ks > 1 && ks * out_bytes > (128ll << 20)) break;
        const int64_t nwg = nwg1 * ks;
        const double per_cu = (double)nwg / cus;
        const double share = 0.58 + 0.42 * std::min(1.0, std::max(0.0, per_cu - 1.0));
        const double rounds = std::max(1.0, (double)ceil_div(nwg, 2 * (int64_t)cus));
        double cost = rounds * ((double)ceil_div(p.nchunks, ks) * chunk_us * share + fixed_us);
        if (ks > 1) cost += 14.0 + (double)(ks + 1) * (double)out_bytes / 2.5e6;
        if (cost < best_h * 0.97) {
          best_h = cost;
          chosen = ntw;
          chosen_ks = ks;
        }
      }
    }
  }
  for (int i = 0; i < 4 && h16 && !h16_one; ++i) {
    // 16-bit operand modes, queue-driven kernels (M355_H16_ONESHOT=0 / 3): fill the chip once, largest tile
    // first; the instantiated tiles are NTW <= 4 (<= 2 for 8 lanes along x)
    const int ntw = cands[i];
    if (ntw == 8 || (p.gx == 8 && ntw == 4)) continue;
    if (force_ntw && ntw != force_ntw && force_ntw != 8 && !(p.gx == 8 && force_ntw == 4)) continue;
    const int ty = ntw * gy;
    if (ty > H && ntw > 1 && !force_ntw) continue;
    const int64_t nwg = (int64_t)p.tz_tiles * ceil_div(H, ty) * p.tx_tiles * p.otiles * N;
    int64_t ks = std::max<int64_t>(1, std::min<int64_t>(ceil_div(512, nwg), std::min<int64_t>(p.nchunks, 8)));
    while (ks > 1 && ks * out_bytes > (128ll << 20)) --ks;
    while (ks > 1 && (ks - 1) * ceil_div(p.nchunks, ks) >= p.nchunks) --ks;
    chosen = ntw;
    chosen_ks = (int)ks;
    if (nwg * ks * 4 >= 512 * 3) break;
  }
  if (x3) {
    // conv3_f32x3_kernel: one item per workgroup, two workgroups per CU, NTW <= 4.  A chunk (8 channels) is 14 x 6 x NTW
    // MFMAs of 32 cycles per wave at the ~1.6 GHz the bf16 pipe holds; split-K as for the fp32 kernels
    double best3 = 1e30;
    for (int ntw : {4, 2, 1}) {
      if (force_ntw && ntw != force_ntw && force_ntw != 8) continue;
      const int ty = ntw * gy;
      if (ty > H && ntw > 1 && !force_ntw) continue;
      const int64_t base_wg = (int64_t)p.tz_tiles * ceil_div(H, ty) * p.tx_tiles * (p.otiles + p.tile16) * N;   // (a 16-row item: half the time)
      const double chunk_us = 14.0 * 6.0 * ntw * 32.0 / 1600.0 / (ntw >= 4 ? 1.0 : ntw == 2 ? 0.9 : 0.75);
      for (int ks = 1; ks <= std::min(p.nchunks, 8); ++ks) {
        if (ks > 1 && (ks - 1) * ceil_div(p.nchunks, ks) >= p.nchunks) continue;
        if (ks > 1 && ks * out_bytes > (128ll << 20)) break;
        const int64_t nwg = base_wg * ks;
        const double rounds = (double)ceil_div(nwg, (int64_t)cus * 2);
        const double share = nwg <= cus ? 1.0 / 0.8 : 2.0;
        double cost = rounds * share * ((double)ceil_div(p.nchunks, ks) + 1.0) * chunk_us;
        if (ks > 1) cost += (2.0 * ks + 1.0) * (double)out_bytes / 4.0e6 + 4.0;
        if (cost < best3 * 0.98) {
          best3 = cost;
          chosen = ntw;
          chosen_ks = ks;
        }
      }
    }
  }
  for (int i = 0; i < 4 && !h16 && !x3; ++i) {
    const int ntw = cands[i];
    if (p.tile16 && ntw == 8) continue;              // the 16-row kernel is instantiated for NTW <= 4
    if (force_ntw && ntw != force_ntw && !(p.tile16 && force_ntw == 8)) continue;
    const int ty = ntw * gy;
    if (ty > H && ntw > 1 && !force_ntw) continue;  // do not overhang H by a whole factor
    const int64_t base_wg = (int64_t)p.tz_tiles * ceil_div(H, ty) * p.tx_tiles * wtiles * N;
    const int per_cu = ntw <= 4 ? 2 : 1;
    // narrow tiles re-read the weights from LDS more often per MFMA ((1 + NTW) / NTW reads each)
    const double chunk_us = 54.0 * ntw * 64.0 / 2040.0 / (ntw >= 4 ? 1.0 : ntw == 2 ? 0.96 : 0.8);
    for (int ks = 1; ks <= std::min(p.nchunks, 8); ++ks) {
      if (ks > 1 && (ks - 1) * ceil_div(p.nchunks, ks) >= p.nchunks) continue;  // an empty split
      if (ks > 1 && ks * out_bytes > (128ll << 20)) break;
      const int64_t nwg = base_wg * ks;
      const double rounds = (double)ceil_div(nwg, (int64_t)cus * per_cu);
      // a lone workgroup on a CU has nothing to cover its barriers and LDS commits: measured ~0.8 of
      // the paired rate for NTW <= 4 (u0.c0 pinned to one per CU: 111 vs 126 TFLOP/s), ~0.93 for NTW = 8
      const bool lone = per_cu == 1 || nwg <= cus;
      const double share = lone ? 1.0 / (per_cu == 1 ? 0.93 : 0.8) : (double)per_cu;
      // fixed cost of an item: ~1 chunk for a one-shot workgroup, ~0.5 when the persistent kernel
      // (more items than resident workgroups) prefetches across the item boundary
      const double fixed = nwg > (int64_t)cus * per_cu ? 0.5 : 1.0;
      double cost = rounds * share * ((double)ceil_div(p.nchunks, ks) + fixed) * chunk_us;
      if (ks > 1) cost += (2.0 * ks + 1.0) * (double)out_bytes / 4.0e6 + 4.0;
      if (cost < best * 0.98) {  // candidates come in order of preference: switch only for a real gain
        best = cost;
        chosen = ntw;
        chosen_ks = ks;
      }
    }
  }
  if (h16 && !h16_one && p.gx == 32 && tuning().h16_w8 && D >= 8 && H >= 2) {
    // 8-wave double-buffered variant (tile 8 x 2 x 32, one workgroup per CU) for SHORT items (<= 4 chunks = 64
    // input channels) whose tiles fill the chip without split-K: there the single-buffered kernel spends as long
    // on chunk boundaries and item switches as on MFMAs (32->32 @128^3: 0.194 -> 0.167 ms, 32->64 @64^3: 0.088 ->
    // 0.058).  Long items stay on the 4-wave kernel: its 4-row wave tile needs 0.75 LDS fragment reads per MFMA,
    // the 2-row tile of this variant 1.17, and at 6+ chunks that LDS traffic costs more than the boundaries
    // (96->32 @128^3: 0.33 vs 0.41 ms).
    const int64_t items8 = (int64_t)ceil_div(D, 8) * ceil_div(H, 2) * p.tx_tiles * p.otiles * N;
    if (((items8 >= 2 * (int64_t)cus && p.nchunks <= 4) || tuning().h16_w8 == 2) && (!force_ntw || force_ntw == 2)) {   // 2: always (tests)
      p.nw = 8;
      p.tz_tiles = (int)ceil_div(D, 8);
      chosen = 2;
      chosen_ks = 1;
    }
  }
  if (h16_one) p.oneshot = 1;
  if (force_ks) {
    chosen_ks = std::min(force_ks, p.nchunks);
    while (chosen_ks > 1 && (chosen_ks - 1) * (int)ceil_div(p.nchunks, chosen_ks) >= p.nchunks)
      --chosen_ks;
  }
  p.ntw = chosen;
  p.ty_tiles = (int)ceil_div(H, p.ntw * gy);
  p.ksplit = chosen_ks;
  const int ksplit = chosen_ks;
  {
    // resident workgroups (LDS + registers: 2 per CU up to NTW = 4); the override exists for the tests
    const int64_t slots = (tuning().conv_slots ? tuning().conv_slots : (p.nw == 8 ? 1 : (p.ntw <= 4 ? 2 : 1)) * num_cus());
    const int64_t items = (int64_t)p.tz_tiles * p.ty_tiles * p.tx_tiles * p.otiles * N * p.ksplit;
    // single-chunk items (Cin <= 4: the first conv of the network, the data gradient of the output conv) have no
    // second chunk to hide the queue ticket's round trip or the next item's prefetch behind: the one-shot grid is
    // faster there (4->32 @128^3: 0.187 vs 0.248 ms)
    // ... and the queue only pays beyond two residencies of items: up to there the one-shot grid, whose workgroups
    // the hardware hands out as CUs free up, is 3-7 % faster (192->64 @64^3, 2.0 residencies: 1.237 -> 1.195 ms;
    // 128->384 @32^3, 1.5: 0.672 -> 0.628); from 3.4 residencies (40->40 @96^3) the queue wins by 7-9 %
    p.persistent = compute == M355_COMPUTE_F32 && !x3 && items < (1ll << 31) && tuning().conv_persistent &&
                   (tuning().conv_persistent > 1 ? items > slots
                                                 : (items > 2 * slots && ceil_div(p.nchunks, p.ksplit) > 1));
    (void)wtiles;
  }
  // packed weights + 256 B for the work counter of the persistent kernel
  p.wp_bytes = (size_t)round_up((int64_t)p.kin_pad * 27 * p.mout_pad * (h16 ? 2 : 4), 256) + 256;
  if (x3)   // [tile][chunk][pair][plane][lane] x 16 B, then the 16-row tile's [chunk][quad][plane][lane] x 16 B
    p.wp_bytes = (size_t)p.otiles * p.nchunks * (14 * 3 * 1024) + (size_t)p.tile16 * p.nchunks * (7 * 3 * 1024) + 256;
  p.slab_bytes = ksplit > 1 ? (size_t)ksplit * N * mout * D * H * W * 4 : 0;
  return p;
}

// The MFMA kernels: 3x3x3, stride 1, padding 1, and a volume whose 4-channel slab fits the 32-bit byte
// offsets of a buffer descriptor (< 2^27 voxels, i.e. below 512^3); anything else takes the generic
// direct kernels (64-bit indexing).
static bool is_k3s1p1(const m355_conv3d_desc* d) {
  return d->k == 3 && d->stride == 1 && d->pad == 1 && (int64_t)d->D * d->H * d->W < (1ll << 27);
}

static int out_dim(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

// Cout <= 4 forward in exact fp32: z-Toeplitz packed rows instead of a mostly-empty 32-row tile
static bool small_cout_fwd(const m355_conv3d_desc* d) {
  return d->Cout <= 4 && !is16(d->compute) && d->W >= 32 && d->D >= 8 && d->Cin >= 8 &&
         !tuning().no_small && (int64_t)std::max(d->Cin, d->Cout) * d->D * d->H * d->W < (1ll << 31);
}
static size_t small_cout_ws(const m355_conv3d_desc* d) {
  return (size_t)round_up((int64_t)round_up(d->Cin, 2) * TZ_K * 32 * 4, 256);
}
static bool small_bww(const m355_conv3d_desc* d) {
  // tap-on-lane kernel; a sample must fit the 32-bit byte offsets of a buffer descriptor
  return (d->Cin <= 4 || d->Cout <= 4) && !tuning().no_small &&
         (int64_t)std::max(d->Cin, d->Cout) * d->D * d->H * d->W < (1ll << 29);
}

template <int NTW, int GX>
static void launch_fwd(const FwdPlan& p, const float* x, const float* wp, const float* bias,
                       const float* add, float* y, float* slab, int N, int kin, int mout, int D,
                       int H, int W, int64_t xbs, int64_t ybs, hipStream_t st, float* stat = nullptr,
                       int* work_counter = nullptr) {
  dim3 grid((unsigned)(p.tz_tiles * p.ty_tiles * p.tx_tiles * std::max(1, p.otiles)), 1u, (unsigned)(N * p.ksplit));
  const int64_t slab_stride = (int64_t)N * mout * D * H * W;
  const int64_t slots = (tuning().conv_slots ? tuning().conv_slots : (NTW <= 4 ? 2 : 1) * num_cus());
  if (p.otiles > 0) {
    if (p.persistent) {
      hipLaunchKernelGGL((conv3_mfma_fwd_p_kernel<NTW, GX, false>), dim3((unsigned)slots), dim3(256), 0, st, x, wp, bias,
                         add, y, slab, kin, mout, D, H, W, p.mout_pad, p.tz_tiles, p.ty_tiles, p.tx_tiles, p.otiles,
                         p.nchunks, p.ksplit, N, xbs, ybs, slab_stride, stat, work_counter, 0, tuning().conv_cube);
    } else {
      hipLaunchKernelGGL((conv3_mfma_fwd_kernel<NTW, GX>), grid, dim3(256), 0, st, x, wp, bias, add,
                         y, slab, kin, mout, D, H, W, p.mout_pad, p.ty_tiles, p.tx_tiles, p.nchunks,
                         p.ksplit, xbs, ybs, slab_stride, stat, p.otiles, tuning().conv_cube);
    }
  }
  if constexpr (NTW <= 4) {
    if (p.tile16) {  // the 16-row remainder tile: queue-driven kernel over (spatial tile x sample x split) items
      const int64_t items = (int64_t)p.tz_tiles * p.ty_tiles * p.tx_tiles * N * p.ksplit;
      // (up to two residencies one workgroup per item: see plan_mfma)
      const int64_t g16 = (items <= 2 * slots && tuning().conv_persistent < 2) ? items : std::min<int64_t>(items, slots);
      hipLaunchKernelGGL((conv3_mfma_fwd_p_kernel<NTW, GX, true>), dim3((unsigned)g16),
                         dim3(256), 0, st, x, wp, bias, add, y, slab, kin, mout, D, H, W, p.mout_pad, p.tz_tiles,
                         p.ty_tiles, p.tx_tiles, 1, p.nchunks, p.ksplit, N, xbs, ybs, slab_stride, stat, work_counter,
                         32 * p.otiles, tuning().conv_cube);
    }
  }
}

// bytes of the c8 staging copy the fp32-input entry points make in 16-bit operand modes
static size_t act16_staging_bytes(int N, int C, int D, int H, int W) {
  return (size_t)round_up((int64_t)N * c8_blocks(C) * D * H * W * 16, 256);
}

static void launch_pack_w3(const FwdPlan& p, const float* w, float* wp, int Cout_w, int Cin_w, bool transpose,
                           hipStream_t st) {
  const int64_t total = (int64_t)p.kin_pad * 27 * p.mout_pad;
  const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 2048);
  hipLaunchKernelGGL(pack_w3_kernel, dim3(blocks), dim3(256), 0, st, w, wp, Cout_w, Cin_w, p.kin_pad, p.mout_pad,
                     transpose ? 1 : 0, (int*)((char*)wp + p.wp_bytes - 256));
}

static int run_mfma_conv(const float* in, const float* w, bool transpose, int Cout_w, int Cin_w,
                         const float* bias, const float* add, float* out, int N, int kin,
                         int mout, int D, int H, int W, int64_t in_bs, int64_t out_bs, void* ws,
                         size_t ws_bytes, hipStream_t st, int compute = M355_COMPUTE_F32, float* stat = nullptr,
                         const void* in16 = nullptr, int64_t in16_bs = 0, const void* prepacked = nullptr,
                         bool out16 = false, bool softmax = false) {
  // prepacked: weights already packed for this plan by m355_conv3d_pack (M355_CONV_W_PACKED); `w` is then unused
  const FwdPlan p = plan_mfma(N, kin, mout, D, H, W, compute);
  M355_REQUIRE(!stat || p.ksplit == 1 || !is16(compute) || out16, M355_EINVALID_ARG,
               "conv3d_fwd_stats: no fused statistics for this plan (m355_conv3d_stats_slots() == 0)");
  if (is16(compute)) {
    if (!in16) {
      // fp32 NCDHW input: one conversion pass into the c8 layout (the model path hands over c8 tensors that its
      // normalisation / pooling passes wrote, m355_conv3d_fwd_h16)
      const size_t base = p.wp_bytes + p.slab_bytes;
      M355_REQUIRE(ws_bytes >= base + act16_staging_bytes(N, kin, D, H, W), M355_EWORKSPACE,
                   "conv3d(16-bit operands): workspace too small (%zu < %zu)", ws_bytes,
                   base + act16_staging_bytes(N, kin, D, H, W));
      void* stage = (char*)ws + base;
      in16_bs = c8_blocks(kin) * (int64_t)D * H * W * 8;
      if (int rc = launch_pack_act16(in, stage, N, kin, (int64_t)D * H * W, in_bs, in16_bs, compute, st)) return rc;
      in16 = stage;
    }
    return run_h16_conv(p, compute, in16, in16_bs, w, transpose, Cout_w, Cin_w, bias, add, out, N, kin, mout, D, H, W,
                        out_bs, ws, ws_bytes, st, stat, prepacked, out16, softmax);
  }
  M355_REQUIRE(!softmax, M355_EUNSUPPORTED, "conv3d: no fused softmax in the fp32 MFMA kernels");
  M355_REQUIRE(ws_bytes >= p.wp_bytes + p.slab_bytes, M355_EWORKSPACE,
               "conv3d: workspace too small (%zu < %zu)", ws_bytes, p.wp_bytes + p.slab_bytes);
  M355_REQUIRE(((uintptr_t)ws & 15) == 0, M355_EINVALID_ARG, "conv3d: workspace not 16B aligned");
  float* wp = prepacked ? (float*)prepacked : (float*)ws;
  float* slab = (float*)((char*)ws + p.wp_bytes);
  int* work_counter = queue_state(st);   // per (device, stream): concurrent launches over one model never share it
  M355_REQUIRE(work_counter, M355_ELAUNCH, "conv3d: could not allocate the work-queue state");
  const float* kb = p.ksplit == 1 ? bias : nullptr;
  const float* ka = p.ksplit == 1 ? add : nullptr;
  if (p.x3) {
    if (!prepacked) launch_pack_w3_x3(p, w, wp, Cout_w, Cin_w, transpose, st);
    if (int rc = launch_x3_conv(p, in, wp, kb, ka, out, slab, N, kin, mout, D, H, W, in_bs, out_bs, st,
                                p.ksplit == 1 ? stat : nullptr))
      return rc;
  } else {
  if (!prepacked) launch_pack_w3(p, w, wp, Cout_w, Cin_w, transpose, st);
#define M355_FWD_CASE(NTW, GX)                                                              \
  if (p.ntw == NTW && p.gx == GX) {                                                         \
    launch_fwd<NTW, GX>(p, in, wp, kb, ka, out, slab, N, kin, mout, D, H, W, in_bs, out_bs, \
                        st, p.ksplit == 1 ? stat : nullptr, work_counter);                  \
  } else
  M355_FWD_CASE(8, 32)
  M355_FWD_CASE(4, 32)
  M355_FWD_CASE(2, 32)
  M355_FWD_CASE(1, 32)
  M355_FWD_CASE(8, 16)
  M355_FWD_CASE(4, 16)
  M355_FWD_CASE(2, 16)
  M355_FWD_CASE(1, 16)
  M355_FWD_CASE(8, 8)
  M355_FWD_CASE(4, 8)
  M355_FWD_CASE(2, 8)
  M355_FWD_CASE(1, 8) {
    set_error("conv3d: no kernel for ntw=%d gx=%d", p.ntw, p.gx);
    return M355_EUNSUPPORTED;
  }
#undef M355_FWD_CASE
  }
  if (p.ksplit > 1 && stat) {   // split plan + fused statistics: the reduction pass emits the partials
    const int64_t S = (int64_t)D * H * W;
    dim3 grid((unsigned)splitk_c8_slots(S), (unsigned)mout, (unsigned)N);
    hipLaunchKernelGGL(splitk_reduce_stats_kernel, grid, dim3(256), 0, st, slab, bias, add, out, mout, S, p.ksplit,
                       (int64_t)N * mout * S, out_bs, stat);
  } else if (p.ksplit > 1) {
    const int64_t S = (int64_t)D * H * W;
    const int64_t total = (int64_t)N * mout * S;
    const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, bias, add, out,
                       N, mout, S, p.ksplit, total, out_bs);
  }
  return check_launch("conv3d_mfma");
}

struct BwwPlan {
  int gx, tz_tiles, ty_tiles, tx_tiles, otiles, ctiles, nsplit;
  size_t slab_bytes;
  bool classes;     // a 1..16 channel remainder on either side: conv3_mfma_bww2c_kernel (needs the gen-2 conditions)
  BwwClasses k;     // always filled: without remainders one class with ns[*] = nsplit
  int class_wgs;    // grid of the class kernel
};

static BwwPlan plan_bww(int N, int Cin, int Cout, int D, int H, int W) {
  BwwPlan p{};
  p.gx = pick_gx(W);
  const int tz = p.gx == 8 ? 4 : 2, ty = p.gx == 32 ? 4 : 8;
  p.tz_tiles = (int)ceil_div(D, tz);
  p.ty_tiles = (int)ceil_div(H, ty);
  p.tx_tiles = (int)ceil_div(W, p.gx);
  p.otiles = (int)ceil_div(Cout, 32);
  p.ctiles = (int)ceil_div(Cin, 32);
  const int64_t ntiles = (int64_t)N * p.tz_tiles * p.ty_tiles * p.tx_tiles;
  const int64_t pairs = (int64_t)p.otiles * p.ctiles;
  const auto rem16 = [](int c) { return c % 32 >= 1 && c % 32 <= 16 ? 1 : 0; };
  p.k.orem = tuning().tile16 ? rem16(Cout) : 0;
  p.k.crem = tuning().tile16 ? rem16(Cin) : 0;
  p.k.of = p.otiles - p.k.orem;
  p.k.cf = p.ctiles - p.k.crem;
  p.classes = (p.k.orem || p.k.crem) && Cin > 4 && Cout > 4;
  // One workgroup per CU; workgroups have equal work, so time ~ rounds x (tiles per split + fixed
  // cost of a workgroup: pipeline fill + the 110 KB slab write, ~half a tile).  Pick the split that
  // minimises it (a power of two up to the tile count) instead of just filling 256 CUs once.
  int64_t nsplit = 1;
  const int cus = num_cus();
  {
    int64_t cand[24];
    int nc = 0;
    for (int64_t ns = 1; ns < ntiles; ns *= 2) cand[nc++] = ns;
    for (int r = 1; r <= 8; ++r) cand[nc++] = std::max<int64_t>(1, (int64_t)cus * r / pairs);  // exactly r rounds
    cand[nc++] = std::max<int64_t>(1, ntiles);
    std::sort(cand, cand + nc);
    double best = 1e30;
    for (int i = 0; i < nc; ++i) {
      const int64_t ns = std::min<int64_t>(cand[i], std::max<int64_t>(1, ntiles));
      const double rounds = (double)ceil_div(pairs * ns, cus);
      const double cost = rounds * ((double)ceil_div(ntiles, ns) + 0.5);
      if (cost < best * 0.97) {  // prefer fewer splits (less slab traffic) unless clearly better
        best = cost;
        nsplit = ns;
      }
    }
  }
  // Queue-driven: the plan above fills the chip in ONE residency (one workgroup per CU), so a CU that another
  // kernel still holds when this one starts -- an RCCL gradient bucket overlapping the backward pass -- delays
  // exactly the workgroup mapped there, and the launch takes up to twice as long.  Splitting the voxel range 2-3x
  // finer makes 2-3 units per CU that the hardware dispatcher hands to whichever CU is free (a held CU simply
  // takes fewer); every unit still sums a FIXED tile set into its own slab, so the result does not depend on who
  // ran what and stays bit-reproducible.  Each unit pays a pipeline fill and a slab write (and the reduce reads
  // one more slab), so this is only done where a unit keeps >= 32 tiles: measured +0.8 % on 96->32 @128^3 at 3
  // units per CU, but +9 % / +18 % on 32->32 @128^3 / 64->64 @64^3 (11 / 5 tiles per unit), which stay static.
  if (tuning().bww_queue && Cin > 4 && Cout > 4 && pairs * nsplit <= cus) {
    const int64_t per_unit = ceil_div(ntiles, nsplit);
    const int m = per_unit >= 96 ? 3 : (per_unit >= 64 ? 2 : 1);
    if (m * nsplit * (int64_t)Cout * Cin * 27 * 4 <= (96ll << 20)) nsplit *= m;
  }
  if (const int force = tuning().bww_nsplit) nsplit = std::min<int64_t>(force, std::max<int64_t>(1, ntiles));
  if (Cin <= 4 || Cout <= 4)  // tap-on-lane kernel: small LDS footprint, ~3 workgroups per CU
    nsplit = std::max<int64_t>(1, 768 / std::max<int64_t>(1, ceil_div(Cin <= 4 ? Cout : Cin, 32)));
  nsplit = std::min<int64_t>(nsplit, ntiles);
  p.nsplit = (int)nsplit;
  int64_t max_ns = nsplit;
  for (int c = 0; c < 4; ++c) p.k.ns[c] = p.nsplit;
  if (p.classes) {
    // pair classes of the remainder kernel: MFMA cost of a pair in units of a full 32 x 32 pair; the split count of
    // a class is proportional to it, scaled so that the whole launch is `rounds` residencies of equal workgroups
    const double cost[4] = {1.0, 0.5, 0.5, 0.25};
    const int64_t npairs[4] = {(int64_t)p.k.of * p.k.cf, (int64_t)p.k.of * p.k.crem, (int64_t)p.k.orem * p.k.cf,
                               (int64_t)p.k.orem * p.k.crem};
    double units = 0;
    for (int c = 0; c < 4; ++c) units += cost[c] * (double)npairs[c];
    // splits of a full pair: one residency of the chip (one workgroup per CU), never more splits than tiles; the
    // rounding of the per-class counts must not spill a workgroup into a second residency
    double base = std::min((double)ntiles, (double)cus / units);
    if (const int force = tuning().bww_nsplit) base = (double)std::min<int64_t>(force, std::max<int64_t>(1, ntiles));
    int wg = 0;
    for (;;) {
      wg = 0;
      max_ns = 1;
      for (int c = 0; c < 4; ++c) {
        const int64_t ns = std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)(base * cost[c] + 0.5)));
        p.k.ns[c] = npairs[c] ? (int)ns : 1;
        p.k.start[c] = wg;
        wg += (int)(npairs[c] * p.k.ns[c]);
        if (npairs[c]) max_ns = std::max<int64_t>(max_ns, ns);
      }
      if (wg <= cus || base <= 1.0 || tuning().bww_nsplit) break;
      base *= 0.99;
    }
    p.class_wgs = wg;
    max_ns = std::max<int64_t>(max_ns, nsplit);   // the uniform plan stays usable (generic kernel when W % 4 != 0)
  }
  p.slab_bytes = (size_t)round_up(max_ns * Cout * Cin * 27 * 4, 256);
  return p;
}

}  // namespace m355

using namespace m355;

// ---------------------------------------------------------------------- ABI
extern "C" size_t m355_conv3d_fwd_workspace(const m355_conv3d_desc* d) {
  if (!d || !is_k3s1p1(d)) return 0;
  if (small_cout_fwd(d)) return small_cout_ws(d);
  const FwdPlan p = plan_mfma(d->N, d->Cin, d->Cout, d->D, d->H, d->W, d->compute);
  return p.wp_bytes + p.slab_bytes +
         (is16(d->compute) ? act16_staging_bytes(d->N, d->Cin, d->D, d->H, d->W) : 0);
}

static int validate_conv(const m355_conv3d_desc* d, const char* who) {
  M355_REQUIRE(d != nullptr, M355_EINVALID_ARG, "%s: null descriptor", who);
  M355_REQUIRE(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->D > 0 && d->H > 0 && d->W > 0,
               M355_EINVALID_ARG, "%s: non-positive dimension", who);
  M355_REQUIRE(d->k >= 1 && d->k <= 7 && d->stride >= 1 && d->pad >= 0, M355_EINVALID_ARG,
               "%s: bad k/stride/pad (%d/%d/%d)", who, d->k, d->stride, d->pad);
  M355_REQUIRE(d->compute == M355_COMPUTE_F32 || d->compute == M355_COMPUTE_BF16 || d->compute == M355_COMPUTE_F16 ||
                   d->compute == M355_COMPUTE_F32X3,
               M355_EINVALID_ARG,
               "%s: unknown compute mode %d", who, d->compute);
  return M355_OK;
}

// Per (sample, output channel): how many (sum, sum of squares) partials the forward kernel writes
// when statistics are fused (4 waves x spatial tiles); 0 = this descriptor has no fused statistics
// (not 3x3x3 s1 p1, small-Cout kernel, bf16 operand mode, or a split-K plan).
static int64_t conv_stats_slots(const m355_conv3d_desc* d) {
  if (!is_k3s1p1(d) || small_cout_fwd(d)) return 0;
  const FwdPlan p = plan_mfma(d->N, d->Cin, d->Cout, d->D, d->H, d->W, d->compute);
  if (p.ksplit != 1)   // split-K: the fp32 reduction pass emits the partials (one slot per block of it); the 16-bit
    return !is16(d->compute) && d->N <= 65535 && d->Cout <= 65535   // kernels only with a c8 output
               ? splitk_c8_slots((int64_t)d->D * d->H * d->W) : 0;
  return (int64_t)p.tz_tiles * p.ty_tiles * p.tx_tiles * p.nw;
}
extern "C" int64_t m355_conv3d_stats_slots(const m355_conv3d_desc* d) { return d ? conv_stats_slots(d) : 0; }
// c8-output forward of the 16-bit modes: split-K plans emit the partials from their reduction pass
static int64_t conv_stats_slots_c8(const m355_conv3d_desc* d) {
  if (!is_k3s1p1(d) || !is16(d->compute)) return 0;
  FwdPlan p = plan_mfma(d->N, d->Cin, d->Cout, d->D, d->H, d->W, d->compute);
  if (p.ksplit != 1) return splitk_c8_slots((int64_t)d->D * d->H * d->W);
  return (int64_t)p.tz_tiles * p.ty_tiles * p.tx_tiles * p.nw;
}
extern "C" int64_t m355_conv3d_stats_slots_c8(const m355_conv3d_desc* d) { return d ? conv_stats_slots_c8(d) : 0; }

static void launch_pack_smallcout(const m355_conv3d_desc* d, const float* w, float* wpz, hipStream_t st) {
  if (tuning().smallcout_valu && (int64_t)d->D * d->H * d->W < (1ll << 27)) {
    hipLaunchKernelGGL(pack_w3_valu_kernel, dim3((unsigned)ceil_div(d->Cin * 27 * 4, 256)), dim3(256), 0, st, w, wpz,
                       d->Cout, d->Cin);
  } else {
    const int kin_pad = (int)round_up(d->Cin, 2);
    const int64_t total = (int64_t)kin_pad * TZ_K * 32;
    hipLaunchKernelGGL(pack_w3_toeplitz_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 2048)), dim3(256),
                       0, st, w, wpz, d->Cout, d->Cin, kin_pad);
  }
}

// softmax over the output channels in the epilogue: the fp32 packed-FMA kernel for Cout <= 4
static bool fuses_softmax(const m355_conv3d_desc* d) {
  if (!is_k3s1p1(d) || !tuning().fuse_softmax) return false;
  if (is16(d->compute))   // 16-bit kernels (c8 input, m355_conv3d_fwd_h16): in-register epilogue, unsplit plans
    return d->Cout <= 4 && plan_mfma(d->N, d->Cin, d->Cout, d->D, d->H, d->W, d->compute).ksplit == 1;
  return small_cout_fwd(d) && tuning().smallcout_valu && (int64_t)d->D * d->H * d->W < (1ll << 27);
}
extern "C" int32_t m355_conv3d_fuses_softmax(const m355_conv3d_desc* d) { return d && fuses_softmax(d) ? 1 : 0; }

static int conv3d_fwd_impl(const m355_conv3d_desc* d, const float* x, const float* w, const float* bias,
                           const float* add, float* y, float* stat, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (int rc = validate_conv(d, "conv3d_fwd")) return rc;
  M355_REQUIRE(!(d->flags & M355_CONV_SOFTMAX) || fuses_softmax(d), M355_EUNSUPPORTED,
               "conv3d_fwd: M355_CONV_SOFTMAX needs m355_conv3d_fuses_softmax(desc) != 0");
  M355_REQUIRE(!stat || conv_stats_slots(d) > 0, M355_EINVALID_ARG,
               "conv3d_fwd_stats: this descriptor has no fused statistics (m355_conv3d_stats_slots() == 0)");
  M355_REQUIRE(x && w && y, M355_EINVALID_ARG, "conv3d_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int OD = out_dim(d->D, d->k, d->stride, d->pad), OH = out_dim(d->H, d->k, d->stride, d->pad),
            OW = out_dim(d->W, d->k, d->stride, d->pad);
  M355_REQUIRE(OD > 0 && OH > 0 && OW > 0, M355_EINVALID_ARG, "conv3d_fwd: empty output");
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->Cin * d->D * d->H * d->W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->Cout * OD * OH * OW);
  if (is_k3s1p1(d) && small_cout_fwd(d)) {
    M355_REQUIRE(workspace && workspace_bytes >= small_cout_ws(d), M355_EWORKSPACE,
                 "conv3d_fwd: workspace too small (%zu < %zu)", workspace_bytes, small_cout_ws(d));
    M355_REQUIRE(((uintptr_t)workspace & 15) == 0, M355_EINVALID_ARG, "conv3d: workspace not 16B aligned");
    const bool packed = (d->flags & M355_CONV_W_PACKED) != 0;
    float* wpz = packed ? (float*)w : (float*)workspace;
    if (tuning().smallcout_valu && (int64_t)d->D * d->H * d->W < (1ll << 27)) {
      // packed-FMA kernel (see conv3_valu_smallcout_kernel); the workspace of the MFMA variant is larger
      if (!packed) launch_pack_smallcout(d, w, wpz, st);
      const int tyv = (int)ceil_div(d->H, VS_TY), txv = (int)ceil_div(d->W, VS_TX);
      dim3 gv((unsigned)(ceil_div(d->D, VS_TZ) * tyv * txv), (unsigned)d->N);
      hipLaunchKernelGGL(conv3_valu_smallcout_kernel, gv, dim3(256), 0, st, x, wpz, bias, add, y, d->Cin, d->Cout,
                         d->D, d->H, d->W, tyv, txv, xbs, ybs, (d->flags & M355_CONV_SOFTMAX) ? 1 : 0);
      return check_launch("conv3_valu_smallcout");
    }
    const int kin_pad = (int)round_up(d->Cin, 2);
    if (!packed) launch_pack_smallcout(d, w, wpz, st);
    const int tyt = (int)ceil_div(d->H, 8), txt = (int)ceil_div(d->W, 32);
    dim3 grid((unsigned)(ceil_div(d->D, 8) * tyt * txt), (unsigned)d->N);
    hipLaunchKernelGGL(conv3_mfma_fwd_smallcout_kernel, grid, dim3(256), 0, st, x, wpz, bias, add, y, d->Cin,
                       d->Cout, d->D, d->H, d->W, tyt, txt, kin_pad / 2, xbs, ybs);
    return check_launch("conv3_mfma_fwd_smallcout");
  }
  if (is_k3s1p1(d)) {
    const bool packed = (d->flags & M355_CONV_W_PACKED) != 0;
    return run_mfma_conv(x, packed ? nullptr : w, false, d->Cout, d->Cin, bias, add, y, d->N, d->Cin, d->Cout, d->D,
                         d->H, d->W, xbs, ybs, workspace, workspace_bytes, st, d->compute, stat, nullptr, 0,
                         packed ? w : nullptr, false, (d->flags & M355_CONV_SOFTMAX) != 0);
  }
  M355_REQUIRE(!(d->flags & M355_CONV_W_PACKED), M355_EINVALID_ARG, "conv3d_fwd: this descriptor has no packed weights");
  const int64_t total = (int64_t)d->N * d->Cout * OD * OH * OW;
  const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 65535);
  hipLaunchKernelGGL(conv3d_direct_fwd_kernel, dim3(blocks), dim3(256), 0, st, x, w, bias, add, y,
                     d->N, d->Cin, d->Cout, d->D, d->H, d->W, OD, OH, OW, d->k, d->stride, d->pad,
                     xbs, ybs);
  return check_launch("conv3d_direct_fwd");
}

extern "C" int m355_conv3d_fwd(const m355_conv3d_desc* d, const float* x, const float* w,
                               const float* bias, const float* add, float* y, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return conv3d_fwd_impl(d, x, w, bias, add, y, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int m355_conv3d_fwd_stats(const m355_conv3d_desc* d, const float* x, const float* w,
                                     const float* bias, const float* add, float* y, float* stat_partials,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  M355_REQUIRE(stat_partials, M355_EINVALID_ARG, "conv3d_fwd_stats: null statistics buffer");
  return conv3d_fwd_impl(d, x, w, bias, add, y, stat_partials, workspace, workspace_bytes, stream);
}

// ---- packed weights (M355_CONV_W_PACKED) ----
extern "C" size_t m355_conv3d_packed_bytes(const m355_conv3d_desc* d, int32_t which) {
  if (!d || !is_k3s1p1(d) || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return 0;
  if (which == 0 && small_cout_fwd(d)) return small_cout_ws(d);
  const FwdPlan p = which == 0 ? plan_mfma(d->N, d->Cin, d->Cout, d->D, d->H, d->W, d->compute)
                               : plan_mfma(d->N, d->Cout, d->Cin, d->D, d->H, d->W, d->compute);
  return p.wp_bytes;
}

extern "C" int m355_conv3d_pack(const m355_conv3d_desc* d, int32_t which, const float* w, void* packed, void* stream) {
  if (int rc = validate_conv(d, "conv3d_pack")) return rc;
  M355_REQUIRE(w && packed && ((uintptr_t)packed & 15) == 0, M355_EINVALID_ARG, "conv3d_pack: null / unaligned pointer");
  M355_REQUIRE(is_k3s1p1(d) && (which == 0 || which == 1), M355_EUNSUPPORTED,
               "conv3d_pack: only the 3x3x3 / stride 1 / pad 1 kernels have packed weights");
  hipStream_t st = (hipStream_t)stream;
  if (which == 0 && small_cout_fwd(d)) {
    launch_pack_smallcout(d, w, (float*)packed, st);
    return check_launch("conv3d_pack");
  }
  const FwdPlan p = which == 0 ? plan_mfma(d->N, d->Cin, d->Cout, d->D, d->H, d->W, d->compute)
                               : plan_mfma(d->N, d->Cout, d->Cin, d->D, d->H, d->W, d->compute);
  if (p.x3)
    launch_pack_w3_x3(p, w, packed, d->Cout, d->Cin, which == 1, st);
  else if (!is16(d->compute))
    launch_pack_w3(p, w, (float*)packed, d->Cout, d->Cin, which == 1, st);
  else
    launch_pack_w3_h16(p, d->compute, w, packed, d->Cout, d->Cin, which == 1, st);
  return check_launch("conv3d_pack");
}

extern "C" int m355_conv3d_pack_batch(const m355_pack_item* items, int32_t n, void* stream) {
  M355_REQUIRE(items || n == 0, M355_EINVALID_ARG, "conv3d_pack_batch: null items");
  hipStream_t st = (hipStream_t)stream;
  PackBatch b;
  X3PackBatch b3;
  int nb = 0, nb3 = 0;
  for (int i = 0; i < n; ++i) {
    const m355_pack_item& it = items[i];
    const m355_conv3d_desc* d = &it.desc;
    if (int rc = validate_conv(d, "conv3d_pack_batch")) return rc;
    M355_REQUIRE(it.w && it.packed && ((uintptr_t)it.packed & 15) == 0, M355_EINVALID_ARG,
                 "conv3d_pack_batch: item %d: null / unaligned pointer", i);
    M355_REQUIRE(is_k3s1p1(d) && (it.which == 0 || it.which == 1), M355_EUNSUPPORTED,
                 "conv3d_pack_batch: item %d: only the 3x3x3 / stride 1 / pad 1 kernels have packed weights", i);
    if (it.which == 0 && small_cout_fwd(d)) {   // (the Cout <= 4 forward layouts: one per model, launched on its own)
      launch_pack_smallcout(d, it.w, (float*)it.packed, st);
      continue;
    }
    const FwdPlan p = it.which == 0 ? plan_mfma(d->N, d->Cin, d->Cout, d->D, d->H, d->W, d->compute)
                                    : plan_mfma(d->N, d->Cout, d->Cin, d->D, d->H, d->W, d->compute);
    if (p.x3) {   // split + fragment-ordered weights: a batch of their own
      X3PackEntry& e = b3.e[nb3++];
      e.w = it.w;
      e.wq = it.packed;
      e.Cout = d->Cout;
      e.Cin = d->Cin;
      e.nchunks = p.nchunks;
      e.otiles = p.otiles;
      e.tile16 = p.tile16;
      e.transpose = it.which == 1;
      if (nb3 == PACK_BATCH) {
        launch_pack_x3_batch(b3, nb3, st);
        nb3 = 0;
      }
      continue;
    }
    const int kind = !is16(d->compute) ? 0 : (d->compute == M355_COMPUTE_BF16 ? 1 : 2);
    if (nb && (kind == 0) != (b.e[0].kind == 0)) {   // a launch holds fp32 entries or 16-bit entries, not both
      launch_pack_batch(b, nb, st);
      nb = 0;
    }
    PackEntry& e = b.e[nb++];
    e.w = it.w;
    e.wp = it.packed;
    e.counter = (int*)((char*)it.packed + p.wp_bytes - 256);
    e.Cout = d->Cout;
    e.Cin = d->Cin;
    e.kdim = !is16(d->compute) ? p.kin_pad : p.nchunks;
    e.mout_pad = p.mout_pad;
    e.transpose = it.which == 1;
    e.kind = kind;
    if (nb == PACK_BATCH) {
      launch_pack_batch(b, nb, st);
      nb = 0;
    }
  }
  if (nb) launch_pack_batch(b, nb, st);
  if (nb3) launch_pack_x3_batch(b3, nb3, st);
  return check_launch("conv3d_pack_batch");
}

// ---- 16-bit operand modes with c8 tensors handed over by the caller (h16.hpp) ----
extern "C" size_t m355_act16_bytes(int32_t N, int32_t C, int64_t S) {
  if (N <= 0 || C <= 0 || S <= 0) return 0;
  return (size_t)N * (size_t)c8_blocks(C) * (size_t)S * 16;
}

static int validate_act16(const char* who, const void* a, const void* b, int N, int C, int64_t S, int compute) {
  M355_REQUIRE(a && b, M355_EINVALID_ARG, "%s: null pointer", who);
  M355_REQUIRE(N > 0 && C > 0 && S > 0 && N <= 65535 && c8_blocks(C) <= 65535, M355_EINVALID_ARG, "%s: bad shape", who);
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "%s: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16", who);
  return M355_OK;
}

extern "C" int m355_act16_pack(const float* x, void* x16, int32_t N, int32_t C, int64_t S, int64_t x_batch_stride,
                               int64_t x16_batch_stride, int32_t compute, void* stream) {
  if (int rc = validate_act16("act16_pack", x, x16, N, C, S, compute)) return rc;
  M355_REQUIRE(((uintptr_t)x16 & 15) == 0 && x16_batch_stride % 8 == 0, M355_EINVALID_ARG, "act16_pack: c8 tensor not 16B aligned");
  return launch_pack_act16(x, x16, N, C, S, dense_or(x_batch_stride, (int64_t)C * S),
                           dense_or(x16_batch_stride, c8_blocks(C) * S * 8), compute, (hipStream_t)stream);
}

extern "C" int m355_act16_unpack(const void* x16, float* x, int32_t N, int32_t C, int64_t S, int64_t x16_batch_stride,
                                 int64_t x_batch_stride, int32_t compute, void* stream) {
  if (int rc = validate_act16("act16_unpack", x16, x, N, C, S, compute)) return rc;
  M355_REQUIRE(((uintptr_t)x16 & 15) == 0 && x16_batch_stride % 8 == 0, M355_EINVALID_ARG, "act16_unpack: c8 tensor not 16B aligned");
  return launch_unpack_act16(x16, x, N, C, S, dense_or(x16_batch_stride, c8_blocks(C) * S * 8),
                             dense_or(x_batch_stride, (int64_t)C * S), compute, (hipStream_t)stream);
}

extern "C" size_t m355_conv3d_h16_workspace(const m355_conv3d_desc* d, int32_t which) {
  if (!d || !is_k3s1p1(d) || !is16(d->compute)) return 0;
  const FwdPlan p = which == 0 ? plan_mfma(d->N, d->Cin, d->Cout, d->D, d->H, d->W, d->compute)
                               : plan_mfma(d->N, d->Cout, d->Cin, d->D, d->H, d->W, d->compute);
  return p.wp_bytes + p.slab_bytes;
}

static int validate_h16(const m355_conv3d_desc* d, const char* who) {
  if (int rc = validate_conv(d, who)) return rc;
  M355_REQUIRE(is_k3s1p1(d) && is16(d->compute), M355_EUNSUPPORTED,
               "%s: c8 input is only defined for the 3x3x3 / stride 1 / pad 1 kernels in a 16-bit compute mode", who);
  return M355_OK;
}

extern "C" int m355_conv3d_fwd_h16(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride, const float* w,
                                   const float* bias, const float* add, float* y, float* stat_partials, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (int rc = validate_h16(d, "conv3d_fwd_h16")) return rc;
  M355_REQUIRE(x16 && w && y && workspace, M355_EINVALID_ARG, "conv3d_fwd_h16: null pointer");
  M355_REQUIRE(!stat_partials || conv_stats_slots(d) > 0, M355_EINVALID_ARG,
               "conv3d_fwd_h16: this descriptor has no fused statistics (m355_conv3d_stats_slots() == 0)");
  const bool softmax = (d->flags & M355_CONV_SOFTMAX) != 0;
  M355_REQUIRE(!softmax || fuses_softmax(d), M355_EUNSUPPORTED,
               "conv3d_fwd_h16: M355_CONV_SOFTMAX needs m355_conv3d_fuses_softmax(desc) != 0");
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const bool packed = (d->flags & M355_CONV_W_PACKED) != 0;
  return run_mfma_conv(nullptr, packed ? nullptr : w, false, d->Cout, d->Cin, bias, add, y, d->N, d->Cin, d->Cout, d->D,
                       d->H, d->W, 0, dense_or(d->y_batch_stride, (int64_t)d->Cout * S), workspace, workspace_bytes,
                       (hipStream_t)stream, d->compute, stat_partials, x16,
                       dense_or(x16_batch_stride, c8_blocks(d->Cin) * S * 8), packed ? w : nullptr, false, softmax);
}

extern "C" int m355_conv3d_fwd_h16_c8(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride,
                                      const float* w, const float* bias, void* y16, int64_t y16_batch_stride,
                                      float* stat_partials, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = validate_h16(d, "conv3d_fwd_h16_c8")) return rc;
  M355_REQUIRE(x16 && w && y16 && workspace, M355_EINVALID_ARG, "conv3d_fwd_h16_c8: null pointer");
  M355_REQUIRE(!stat_partials || conv_stats_slots_c8(d) > 0, M355_EINVALID_ARG,
               "conv3d_fwd_h16_c8: this descriptor has no fused statistics (m355_conv3d_stats_slots_c8() == 0)");
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const bool packed = (d->flags & M355_CONV_W_PACKED) != 0;
  return run_mfma_conv(nullptr, packed ? nullptr : w, false, d->Cout, d->Cin, bias, nullptr, (float*)y16, d->N, d->Cin,
                       d->Cout, d->D, d->H, d->W, 0, dense_or(y16_batch_stride, c8_blocks(d->Cout) * S * 8), workspace,
                       workspace_bytes, (hipStream_t)stream, d->compute, stat_partials, x16,
                       dense_or(x16_batch_stride, c8_blocks(d->Cin) * S * 8), packed ? w : nullptr, true);
}

extern "C" int m355_conv3d_bwd_data_h16(const m355_conv3d_desc* d, const void* dy16, int64_t dy16_batch_stride,
                                        const float* w, float* dx, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  if (int rc = validate_h16(d, "conv3d_bwd_data_h16")) return rc;
  M355_REQUIRE(dy16 && w && dx && workspace, M355_EINVALID_ARG, "conv3d_bwd_data_h16: null pointer");
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const bool packed = (d->flags & M355_CONV_W_PACKED) != 0;
  return run_mfma_conv(nullptr, packed ? nullptr : w, true, d->Cout, d->Cin, nullptr, nullptr, dx, d->N, d->Cout, d->Cin,
                       d->D, d->H, d->W, 0, dense_or(d->x_batch_stride, (int64_t)d->Cin * S), workspace, workspace_bytes,
                       (hipStream_t)stream, d->compute, nullptr, dy16,
                       dense_or(dy16_batch_stride, c8_blocks(d->Cout) * S * 8), packed ? w : nullptr);
}

// ---- weight gradient with both operands in c8 (the 16-bit training flow keeps the packed conv input of the forward
// pass and packs dy once for the data and the weight gradient) ----
static size_t dbias_ws_bytes(int Cout, int64_t S);
int launch_dbias(const float* dy, float* dbias, int N, int Cout, int64_t S, int64_t ybs, void* ws, hipStream_t st);
static int bww_c8_nsplit(const m355_conv3d_desc* d) {
  const int64_t ntiles = (int64_t)d->N * ceil_div(d->D, 2) * ceil_div(d->H, 4) * ceil_div(d->W, 32);
  const int64_t pairs = ceil_div(d->Cin, 32) * ceil_div(d->Cout, 32);
  const int64_t slots = 2 * (int64_t)num_cus();
  if (const int force = tuning().bww_nsplit) return (int)std::min<int64_t>(force, ntiles);
  // time ~ residencies x (tiles per split x tile time + ~4 us pipeline fill and slab write) + the slab traffic (written
  // by the kernel, read by the reduction).  Tile time ~1.8 us with two workgroups sharing a CU, ~1.1 us alone: for few
  // pairs one workgroup per CU with half the slabs wins (32->32 @128^3: 256 splits 170 us, 512 splits 184 us), for many
  // tiles per pair two per CU do (tools/plan_sweep_bww_c8.py).
  const double slab_us = 2.0 * (double)d->Cout * d->Cin * 27 * 4 / 4.0e6;
  double best = 1e30;
  int64_t best_ns = 1;
  for (int h = pairs <= 2 ? 1 : 2; h <= 8; ++h) {   // h half-residencies: 256, 512, 768, ... workgroups (one per CU
                                                    // only pays for one or two pairs: more pairs share tiles in L2)
    const int64_t ns = std::max<int64_t>(1, std::min<int64_t>(ntiles, slots * h / (2 * pairs)));
    const int64_t wgs = pairs * ns;
    const double rounds = (double)ceil_div(wgs, slots);
    const double tile_us = wgs * 2 <= slots ? 1.1 : (wgs >= slots ? 1.8 : 1.1 + 0.7 * (double)(wgs * 2 - slots) / (double)slots);
    const double cost = rounds * ((double)ceil_div(ntiles, ns) * tile_us + 4.0) + (double)ns * slab_us;
    if (cost < best * 0.97) {
      best = cost;
      best_ns = ns;
    }
  }
  return (int)best_ns;
}

static bool bww_c8_ok(const m355_conv3d_desc* d) {
  return is_k3s1p1(d) && is16(d->compute) && (int64_t)d->D * d->H * d->W * 64 < (1ll << 31);
}

extern "C" size_t m355_conv3d_bwd_weight_h16_workspace(const m355_conv3d_desc* d) {
  if (!d || !bww_c8_ok(d)) return 0;
  return (size_t)round_up((int64_t)bww_c8_nsplit(d) * d->Cout * d->Cin * 27 * 4, 256) +
         dbias_ws_bytes(d->Cout, (int64_t)d->D * d->H * d->W);
}

extern "C" int m355_conv3d_bwd_weight_h16(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride,
                                          const void* dy16, int64_t dy16_batch_stride, const float* dy, float* dw,
                                          float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = validate_h16(d, "conv3d_bwd_weight_h16")) return rc;
  M355_REQUIRE(x16 && dy16 && dw && workspace, M355_EINVALID_ARG, "conv3d_bwd_weight_h16: null pointer");
  M355_REQUIRE(bww_c8_ok(d), M355_EUNSUPPORTED, "conv3d_bwd_weight_h16: volume too large for the c8 kernel (>= 2^25 voxels)");
  M355_REQUIRE(!dbias || dy, M355_EINVALID_ARG, "conv3d_bwd_weight_h16: the bias gradient needs the fp32 dy");
  M355_REQUIRE(workspace_bytes >= m355_conv3d_bwd_weight_h16_workspace(d), M355_EWORKSPACE,
               "conv3d_bwd_weight_h16: workspace too small (%zu < %zu)", workspace_bytes,
               m355_conv3d_bwd_weight_h16_workspace(d));
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const int64_t xbs = dense_or(x16_batch_stride, c8_blocks(d->Cin) * S * 8);
  const int64_t ybs = dense_or(dy16_batch_stride, c8_blocks(d->Cout) * S * 8);
  M355_REQUIRE((((uintptr_t)x16 | (uintptr_t)dy16) & 15) == 0 && xbs % 8 == 0 && ybs % 8 == 0, M355_EINVALID_ARG,
               "conv3d_bwd_weight_h16: c8 tensor not 16B aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = bww_c8_nsplit(d);
  float* slab = (float*)workspace;
  if (int rc = launch_bww_c8(d->compute, x16, dy16, slab, d->N, d->Cin, d->Cout, d->D, d->H, d->W, nsplit, xbs, ybs, st))
    return rc;
  BwwClasses kred{};
  kred.of = (int)ceil_div(d->Cout, 32);
  kred.cf = (int)ceil_div(d->Cin, 32);
  for (int c = 0; c < 4; ++c) kred.ns[c] = nsplit;
  launch_slab_reduce_t(slab, dw, d->Cin, d->Cout, kred.cf, kred, 1.f, st);
  if (dbias) {
    const size_t slab_b = (size_t)round_up((int64_t)nsplit * d->Cout * d->Cin * 27 * 4, 256);
    launch_dbias(dy, dbias, d->N, d->Cout, S, dense_or(d->y_batch_stride, (int64_t)d->Cout * S), (char*)workspace + slab_b, st);
  }
  return check_launch("conv3d_bwd_weight_h16");
}

// ---- the c8-only training flow: data gradient written as c8, weight gradient with the bias gradient reduced from
// the c8 dy and the loss scale of the fp16 mode removed in the fp32 epilogue ----
extern "C" int m355_conv3d_bwd_data_h16_c8(const m355_conv3d_desc* d, const void* dy16, int64_t dy16_batch_stride,
                                           const float* w, void* dx16, int64_t dx16_batch_stride, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (int rc = validate_h16(d, "conv3d_bwd_data_h16_c8")) return rc;
  M355_REQUIRE(dy16 && w && dx16 && workspace, M355_EINVALID_ARG, "conv3d_bwd_data_h16_c8: null pointer");
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const bool packed = (d->flags & M355_CONV_W_PACKED) != 0;
  return run_mfma_conv(nullptr, packed ? nullptr : w, true, d->Cout, d->Cin, nullptr, nullptr, (float*)dx16, d->N, d->Cout,
                       d->Cin, d->D, d->H, d->W, 0, dense_or(dx16_batch_stride, c8_blocks(d->Cin) * S * 8), workspace,
                       workspace_bytes, (hipStream_t)stream, d->compute, nullptr, dy16,
                       dense_or(dy16_batch_stride, c8_blocks(d->Cout) * S * 8), packed ? w : nullptr, true);
}

extern "C" size_t m355_conv3d_bwd_weight_c8_workspace(const m355_conv3d_desc* d) {
  if (!d || !bww_c8_ok(d)) return 0;
  return (size_t)round_up((int64_t)bww_c8_nsplit(d) * d->Cout * d->Cin * 27 * 4, 256) +
         dbias_c8_ws_bytes(d->N, d->Cout, (int64_t)d->D * d->H * d->W);
}

extern "C" int m355_conv3d_bwd_weight_c8(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride,
                                         const void* dy16, int64_t dy16_batch_stride, float* dw, float* dbias,
                                         float grad_unscale, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = validate_h16(d, "conv3d_bwd_weight_c8")) return rc;
  M355_REQUIRE(x16 && dy16 && dw && workspace, M355_EINVALID_ARG, "conv3d_bwd_weight_c8: null pointer");
  M355_REQUIRE(bww_c8_ok(d), M355_EUNSUPPORTED, "conv3d_bwd_weight_c8: volume too large for the c8 kernel (>= 2^25 voxels)");
  M355_REQUIRE(workspace_bytes >= m355_conv3d_bwd_weight_c8_workspace(d), M355_EWORKSPACE,
               "conv3d_bwd_weight_c8: workspace too small (%zu < %zu)", workspace_bytes, m355_conv3d_bwd_weight_c8_workspace(d));
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const int64_t xbs = dense_or(x16_batch_stride, c8_blocks(d->Cin) * S * 8);
  const int64_t ybs = dense_or(dy16_batch_stride, c8_blocks(d->Cout) * S * 8);
  M355_REQUIRE((((uintptr_t)x16 | (uintptr_t)dy16) & 15) == 0 && xbs % 8 == 0 && ybs % 8 == 0, M355_EINVALID_ARG,
               "conv3d_bwd_weight_c8: c8 tensor not 16B aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = bww_c8_nsplit(d);
  float* slab = (float*)workspace;
  // edge layers (first conv: Cin <= 4; output conv: Cout <= 4): tap and narrow channel share the MFMA column
  const bool edge = (d->Cin <= 4 || d->Cout <= 4) && !tuning().no_small;
  if (int rc = edge ? launch_bww_c8_small(d->compute, x16, dy16, slab, d->N, d->Cin, d->Cout, d->D, d->H, d->W, nsplit, xbs, ybs, st)
                    : launch_bww_c8(d->compute, x16, dy16, slab, d->N, d->Cin, d->Cout, d->D, d->H, d->W, nsplit, xbs, ybs, st))
    return rc;
  BwwClasses kred{};
  kred.of = (int)ceil_div(d->Cout, 32);
  kred.cf = (int)ceil_div(d->Cin, 32);
  for (int c = 0; c < 4; ++c) kred.ns[c] = nsplit;
  launch_slab_reduce_t(slab, dw, d->Cin, d->Cout, kred.cf, kred, grad_unscale, st);
  if (dbias) {
    const size_t slab_b = (size_t)round_up((int64_t)nsplit * d->Cout * d->Cin * 27 * 4, 256);
    if (int rc = launch_dbias_c8(dy16, ybs, dbias, d->N, d->Cout, S, d->compute, grad_unscale, (char*)workspace + slab_b, st))
      return rc;
  }
  return check_launch("conv3d_bwd_weight_c8");
}

static bool bww_plain_h16(const m355_conv3d_desc* d);
static bool bww_x3(const m355_conv3d_desc* d);
extern "C" int m355_conv3d_plan(const m355_conv3d_desc* d, int32_t which, int32_t* out4) {
  M355_REQUIRE(d && out4, M355_EINVALID_ARG, "conv3d_plan: null pointer");
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  if (!is_k3s1p1(d)) return M355_OK;
  if (which == 2) {   // weight gradient of the plain entry point: 8 = conv3_bww_x3_kernel, 9 = conv3_mfma_bww2(c)_kernel,
                      // 10 = conv3_mfma_bww_small_kernel, 11 = the c8 kernel behind an operand pack (16-bit modes)
    if (bww_x3(d)) {
      const BwwX3Plan p = plan_bww_x3(d->N, d->Cin, d->Cout, d->D, d->H, d->W);
      out4[0] = 8; out4[2] = p.tx; out4[3] = p.nsplit;
    } else {
      const BwwPlan p = plan_bww(d->N, d->Cin, d->Cout, d->D, d->H, d->W);
      out4[0] = bww_plain_h16(d) ? 11 : (small_bww(d) ? 10 : 9); out4[2] = p.gx; out4[3] = p.nsplit;
    }
    return M355_OK;
  }
  if (which == 0 && small_cout_fwd(d)) { out4[0] = 2; return M355_OK; }  // z-Toeplitz small-Cout kernel
  const FwdPlan p = which == 0 ? plan_mfma(d->N, d->Cin, d->Cout, d->D, d->H, d->W, d->compute)
                               : plan_mfma(d->N, d->Cout, d->Cin, d->D, d->H, d->W, d->compute);
  out4[0] = is16(d->compute) ? (p.oneshot ? 6 : (p.nw == 8 ? 5 : 4)) : (p.x3 ? 7 : (p.persistent ? 3 : 1)); out4[1] = p.ntw; out4[2] = p.gx; out4[3] = p.ksplit;
  return M355_OK;
}

extern "C" size_t m355_conv3d_bwd_data_workspace(const m355_conv3d_desc* d) {
  if (!d || !is_k3s1p1(d)) return 0;
  const FwdPlan p = plan_mfma(d->N, d->Cout, d->Cin, d->D, d->H, d->W, d->compute);
  return p.wp_bytes + p.slab_bytes +
         (is16(d->compute) ? act16_staging_bytes(d->N, d->Cout, d->D, d->H, d->W) : 0);
}

extern "C" int m355_conv3d_bwd_data(const m355_conv3d_desc* d, const float* dy, const float* w,
                                    float* dx, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  if (int rc = validate_conv(d, "conv3d_bwd_data")) return rc;
  M355_REQUIRE(dy && w && dx, M355_EINVALID_ARG, "conv3d_bwd_data: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int OD = out_dim(d->D, d->k, d->stride, d->pad), OH = out_dim(d->H, d->k, d->stride, d->pad),
            OW = out_dim(d->W, d->k, d->stride, d->pad);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->Cin * d->D * d->H * d->W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->Cout * OD * OH * OW);
  if (is_k3s1p1(d)) {
    // dx = conv(dy, flipped/transposed w): K-channels = Cout, M-channels = Cin
    const bool packed = (d->flags & M355_CONV_W_PACKED) != 0;
    return run_mfma_conv(dy, packed ? nullptr : w, true, d->Cout, d->Cin, nullptr, nullptr, dx, d->N, d->Cout, d->Cin,
                         d->D, d->H, d->W, ybs, xbs, workspace, workspace_bytes, st, d->compute, nullptr, nullptr, 0,
                         packed ? w : nullptr);
  }
  M355_REQUIRE(!(d->flags & M355_CONV_W_PACKED), M355_EINVALID_ARG, "conv3d_bwd_data: this descriptor has no packed weights");
  const int64_t total = (int64_t)d->N * d->Cin * d->D * d->H * d->W;
  const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 65535);
  hipLaunchKernelGGL(conv3d_direct_bwd_data_kernel, dim3(blocks), dim3(256), 0, st, dy, w, dx,
                     d->N, d->Cin, d->Cout, d->D, d->H, d->W, OD, OH, OW, d->k, d->stride, d->pad,
                     xbs, ybs);
  return check_launch("conv3d_direct_bwd_data");
}

static size_t dbias_ws_bytes(int Cout, int64_t S) {
  return (size_t)round_up((int64_t)Cout * ceil_div(S, DBIAS_CHUNK) * 8, 256);
}

// dbias through the shared two-stage reduction; `ws` must hold dbias_ws_bytes()
int launch_dbias(const float* dy, float* dbias, int N, int Cout, int64_t S, int64_t ybs, void* ws,
                 hipStream_t st) {
  const int nblk = (int)ceil_div(S, DBIAS_CHUNK);
  if (S % 4 == 0 && ybs % 4 == 0 && ((uintptr_t)dy & 15) == 0)
    hipLaunchKernelGGL(dbias_partial_kernel<true>, dim3((unsigned)nblk, (unsigned)Cout), dim3(256), 0, st, dy,
                       (double*)ws, N, S, ybs, nblk);
  else
    hipLaunchKernelGGL(dbias_partial_kernel<false>, dim3((unsigned)nblk, (unsigned)Cout), dim3(256), 0, st, dy,
                       (double*)ws, N, S, ybs, nblk);
  hipLaunchKernelGGL(dbias_finalize_kernel, dim3((unsigned)ceil_div(Cout, 64)), dim3(64), 0, st,
                     (const double*)ws, dbias, Cout, nblk);
  return M355_OK;
}

// fp32 NCDHW operands in a 16-bit compute mode (the plain entry point; the model path hands over c8 tensors through
// m355_conv3d_bwd_weight_h16 / _c8): both operands are rounded into c8 copies and the c8 kernel runs (round 1 had a kernel
// of its own for this case, conv3_mfma_bww_h16_kernel, three dx-shifted LDS copies at a third of the c8 kernel's rate)
static bool bww_c8_ok(const m355_conv3d_desc* d);
static bool bww_plain_h16(const m355_conv3d_desc* d) {
  return is16(d->compute) && is_k3s1p1(d) && !small_bww(d) && bww_c8_ok(d) && d->N <= 65535;
}

// M355_COMPUTE_F32X3: the weight gradient on the split kernels too (conv3_bww_x3_kernel; M355_F32X3=2 forces every
// fp32 layer there, M355_F32X3_BWW=0 keeps the weight gradient on the fp32 MFMA kernels)
static bool bww_x3(const m355_conv3d_desc* d) {
  const bool mode = d->compute == M355_COMPUTE_F32X3 || (d->compute == M355_COMPUTE_F32 && tuning().f32x3 == 2);
  return mode && tuning().f32x3 && tuning().f32x3_bww && is_k3s1p1(d) && d->Cin > 4 && d->Cout > 4 && d->D >= 2 &&
         (int64_t)d->D * d->H * d->W < (1ll << 24);
}

extern "C" size_t m355_conv3d_bwd_weight_workspace(const m355_conv3d_desc* d) {
  if (!d) return 0;
  const int OD = out_dim(d->D, d->k, d->stride, d->pad), OH = out_dim(d->H, d->k, d->stride, d->pad),
            OW = out_dim(d->W, d->k, d->stride, d->pad);
  const size_t db = dbias_ws_bytes(d->Cout, (int64_t)OD * OH * OW);
  if (!is_k3s1p1(d)) return db;
  if (bww_x3(d)) return plan_bww_x3(d->N, d->Cin, d->Cout, d->D, d->H, d->W).slab_bytes + db;
  const size_t f32 = plan_bww(d->N, d->Cin, d->Cout, d->D, d->H, d->W).slab_bytes + db;
  if (bww_plain_h16(d)) {   // 16-bit operand mode: both operands are rounded into c8 copies behind the c8 kernel's own workspace
    const int64_t S = (int64_t)d->D * d->H * d->W;
    return std::max(f32, m355_conv3d_bwd_weight_h16_workspace(d) + (size_t)round_up(d->N * c8_blocks(d->Cin) * S * 16, 256) +
                             (size_t)round_up(d->N * c8_blocks(d->Cout) * S * 16, 256));
  }
  return f32;
}

extern "C" int m355_conv3d_bwd_weight(const m355_conv3d_desc* d, const float* x, const float* dy,
                                      float* dw, float* dbias, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (int rc = validate_conv(d, "conv3d_bwd_weight")) return rc;
  M355_REQUIRE(x && dy && dw, M355_EINVALID_ARG, "conv3d_bwd_weight: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int OD = out_dim(d->D, d->k, d->stride, d->pad), OH = out_dim(d->H, d->k, d->stride, d->pad),
            OW = out_dim(d->W, d->k, d->stride, d->pad);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->Cin * d->D * d->H * d->W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->Cout * OD * OH * OW);
  size_t slab_used = 0;   // the bias gradient's scratch follows the slabs
  if (bww_x3(d)) {
    const BwwX3Plan p = plan_bww_x3(d->N, d->Cin, d->Cout, d->D, d->H, d->W);
    M355_REQUIRE(workspace && workspace_bytes >= p.slab_bytes, M355_EWORKSPACE,
                 "conv3d_bwd_weight: workspace too small (%zu < %zu)", workspace_bytes, p.slab_bytes);
    M355_REQUIRE((((uintptr_t)x | (uintptr_t)dy) & 3) == 0, M355_EINVALID_ARG, "conv3d_bwd_weight: misaligned tensor");
    float* slab = (float*)workspace;
    if (int rc = launch_bww_x3(p, x, dy, slab, d->N, d->Cin, d->Cout, d->D, d->H, d->W, xbs, ybs, st)) return rc;
    launch_slab_reduce_t(slab, dw, d->Cin, d->Cout, p.ctiles, p.k, 1.f, st);   // p.k.ns: splits of each pair class
    slab_used = p.slab_bytes;
  } else if (is_k3s1p1(d)) {
    const BwwPlan p = plan_bww(d->N, d->Cin, d->Cout, d->D, d->H, d->W);
    slab_used = p.slab_bytes;
    M355_REQUIRE(workspace_bytes >= p.slab_bytes, M355_EWORKSPACE,
                 "conv3d_bwd_weight: workspace too small (%zu < %zu)", workspace_bytes,
                 p.slab_bytes);
    float* slab = (float*)workspace;
    M355_REQUIRE((int64_t)d->Cin * d->D * d->H * d->W < (1ll << 31) &&
                     (int64_t)d->Cout * d->D * d->H * d->W < (1ll << 31),
                 M355_EUNSUPPORTED, "conv3d_bwd_weight: tensor exceeds 2^31 elements per sample");
    if (bww_plain_h16(d)) {
      const int64_t S = (int64_t)d->D * d->H * d->W;
      const size_t hws = m355_conv3d_bwd_weight_h16_workspace(d);
      const size_t xb = (size_t)round_up(d->N * c8_blocks(d->Cin) * S * 16, 256), yb = (size_t)round_up(d->N * c8_blocks(d->Cout) * S * 16, 256);
      M355_REQUIRE(workspace && workspace_bytes >= hws + xb + yb, M355_EWORKSPACE,
                   "conv3d_bwd_weight: workspace too small (%zu < %zu)", workspace_bytes, hws + xb + yb);
      char* x16 = (char*)workspace + hws;
      char* dy16 = x16 + xb;
      if (int rc = launch_pack_act16(x, x16, d->N, d->Cin, S, xbs, c8_blocks(d->Cin) * S * 8, d->compute, st)) return rc;
      if (int rc = launch_pack_act16(dy, dy16, d->N, d->Cout, S, ybs, c8_blocks(d->Cout) * S * 8, d->compute, st)) return rc;
      m355_conv3d_desc dd = *d;
      dd.y_batch_stride = ybs;
      return m355_conv3d_bwd_weight_h16(&dd, x16, 0, dy16, 0, dbias ? dy : nullptr, dw, dbias, workspace, hws, stream);
    } else if (small_bww(d)) {
      // narrow side (<= 4 channels) shares the lane index with the taps
      const int swap = d->Cin <= 4 ? 0 : 1;
      const float* P = swap ? x : dy;
      const float* Q = swap ? dy : x;
      const int CP = swap ? d->Cin : d->Cout, CQ = swap ? d->Cout : d->Cin;
      const int64_t pbs = swap ? xbs : ybs, qbs = swap ? ybs : xbs;
      dim3 g2((unsigned)ceil_div(CP, 32), (unsigned)p.nsplit);
#define M355_BWS_LAUNCH(GXV)                                                                       \
  hipLaunchKernelGGL((conv3_mfma_bww_small_kernel<GXV>), g2, dim3(256), 0, st, P, Q, slab, d->N, CP, \
                     CQ, d->D, d->H, d->W, p.tz_tiles, p.ty_tiles, p.tx_tiles, p.nsplit, pbs, qbs,  \
                     swap, d->Cin, d->Cout);
      if (p.gx == 32) { M355_BWS_LAUNCH(32) } else if (p.gx == 16) { M355_BWS_LAUNCH(16) } else { M355_BWS_LAUNCH(8) }
      const int64_t total = (int64_t)d->Cout * d->Cin * 27;
      const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 2048);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, dw, total, p.nsplit);
    } else {
    dim3 grid((unsigned)p.ctiles, (unsigned)p.otiles, (unsigned)p.nsplit);
    // float4 interior rows need 16-byte aligned rows; per-sample extents must fit int32 offsets
    const bool vec = (d->W % 4 == 0) && (xbs % 4 == 0) && (((uintptr_t)x) & 15) == 0;
#define M355_BWW_LAUNCH(GXV)                                                                      \
  {                                                                                               \
    if (vec)                                                                                      \
      hipLaunchKernelGGL((conv3_mfma_bww_kernel<GXV, true>), grid, dim3(256), 0, st, x, dy, slab, \
                         d->N, d->Cin, d->Cout, d->D, d->H, d->W, p.tz_tiles, p.ty_tiles,       \
                         p.tx_tiles, p.nsplit, xbs, ybs);                                        \
    else                                                                                          \
      hipLaunchKernelGGL((conv3_mfma_bww_kernel<GXV, false>), grid, dim3(256), 0, st, x, dy,      \
                         slab, d->N, d->Cin, d->Cout, d->D, d->H, d->W, p.tz_tiles, p.ty_tiles, \
                         p.tx_tiles, p.nsplit, xbs, ybs);                                        \
  }
    M355_REQUIRE((int64_t)d->Cin * d->D * d->H * d->W < (1ll << 31) &&
                     (int64_t)d->Cout * d->D * d->H * d->W < (1ll << 31),
                 M355_EUNSUPPORTED, "conv3d_bwd_weight: tensor exceeds 2^31 elements per sample");
    // second-generation kernel: float4 rows, and a sample must fit the 32-bit byte offsets of a
    // buffer descriptor (the hardware zero-fills what lies past it)
    const int64_t spatial = (int64_t)d->D * d->H * d->W;
    const bool gen2 = vec && ((uintptr_t)dy & 3) == 0 && (int64_t)d->Cin * spatial < (1ll << 29) &&
                      (int64_t)d->Cout * spatial < (1ll << 29) && tuning().bww_gen == 2;
#define M355_BWW2_LAUNCH(GXV)                                                                     \
  hipLaunchKernelGGL((conv3_mfma_bww2_kernel<GXV>), dim3((unsigned)(p.ctiles * p.otiles * p.nsplit)),  \
                     dim3(256), 0, st, x, dy, slab, d->N, d->Cin, d->Cout, d->D, d->H, d->W, p.tz_tiles, \
                     p.ty_tiles, p.tx_tiles, p.nsplit, p.ctiles, p.otiles, xbs, ybs);
#define M355_BWW2C_LAUNCH(GXV)                                                                    \
  hipLaunchKernelGGL((conv3_mfma_bww2c_kernel<GXV>), dim3((unsigned)p.class_wgs), dim3(256), 0, st, x, dy, slab, d->N, \
                     d->Cin, d->Cout, d->D, d->H, d->W, p.tz_tiles, p.ty_tiles, p.tx_tiles, p.k, xbs, ybs);
    BwwClasses kred = p.k;   // what the reduction sums: the class splits, or the uniform count
    if (gen2 && p.classes) {
      if (p.gx == 32) { M355_BWW2C_LAUNCH(32) } else if (p.gx == 16) { M355_BWW2C_LAUNCH(16) } else { M355_BWW2C_LAUNCH(8) }
    } else if (gen2) {
      for (int c = 0; c < 4; ++c) kred.ns[c] = p.nsplit;
      if (p.gx == 32) { M355_BWW2_LAUNCH(32) } else if (p.gx == 16) { M355_BWW2_LAUNCH(16) } else { M355_BWW2_LAUNCH(8) }
    } else if (p.gx == 32)
      M355_BWW_LAUNCH(32)
    else if (p.gx == 16)
      M355_BWW_LAUNCH(16)
    else
      M355_BWW_LAUNCH(8)
    if (gen2) {
      launch_slab_reduce_t(slab, dw, d->Cin, d->Cout, p.ctiles, kred, 1.f, st);
    } else {
      const int64_t total = (int64_t)d->Cout * d->Cin * 27;
      const int blocks = (int)std::min<int64_t>(ceil_div(total, 64), 4096);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(64), 0, st, slab, dw, total, p.nsplit);
    }
    }
  } else {
    const int k3 = d->k * d->k * d->k;
    const int64_t nblk = (int64_t)d->Cout * d->Cin * k3;
    M355_REQUIRE(nblk < (1ll << 31), M355_EUNSUPPORTED, "conv3d_bwd_weight: grid too large");
    hipLaunchKernelGGL(conv3d_direct_bwd_weight_kernel, dim3((unsigned)nblk), dim3(256), 0, st, x,
                       dy, dw, d->N, d->Cin, d->Cout, d->D, d->H, d->W, OD, OH, OW, d->k,
                       d->stride, d->pad, xbs, ybs);
  }
  if (dbias) {
    const int64_t OS = (int64_t)OD * OH * OW;
    const size_t slab_b = slab_used;
    M355_REQUIRE(workspace && workspace_bytes >= slab_b + dbias_ws_bytes(d->Cout, OS), M355_EWORKSPACE,
                 "conv3d_bwd_weight: workspace too small for the bias gradient");
    launch_dbias(dy, dbias, d->N, d->Cout, OS, ybs, (char*)workspace + slab_b, st);
  }
  return check_launch("conv3d_bwd_weight");
}
