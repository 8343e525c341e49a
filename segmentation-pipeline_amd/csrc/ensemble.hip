// Test-time-augmentation ensembles on the device.
//
// Reference ops replaced (models/ensemble.py): every member runs the model on a flipped / axis-permuted view
// of the input and maps its prediction back (`.flip(f).permute(inverse)`, :61,89-91); apply_strategy (:16-35)
// then stacks the E predictions and takes `mean(dim=0)` or `argmax -> mode -> one_hot`.  Here
//   * the member input is produced by ONE gather pass (flip + permutation as index math) instead of torch's
//     permute / flip / contiguous chain,
//   * a member's prediction is never mapped back or stacked: the accumulate kernel reads it THROUGH the inverse
//     index transform and adds it to a running sum (mean) or casts its argmax as a vote into a per-class
//     histogram (majority); a finalize pass divides, or picks the winning class (ties: the smallest class index,
//     what torch.mode returns on the CPU) and writes the int64 one-hot mask the reference returns.
// All HBM-bound; the canonical (output) side is always the coalesced one.
#include "common.hpp"

namespace m355 {

struct AxisMap {
  int size[3];      // canonical spatial size (i0, i1, i2)
  int64_t mstride[3];  // stride, in the MEMBER tensor, of canonical axis k (negative when that axis is flipped)
  int64_t moff;     // member offset of canonical voxel (0, 0, 0)
};

// perm[j] in {0,1,2}: member axis j is canonical axis perm[j]; flip bit j: member axis j is reversed
static AxisMap make_axis_map(const int32_t* size, const int32_t* perm, int flip_mask) {
  AxisMap m{};
  int msize[3];
  for (int j = 0; j < 3; ++j) msize[j] = size[perm[j]];
  int64_t mst[3] = {(int64_t)msize[1] * msize[2], msize[2], 1};
  m.moff = 0;
  for (int k = 0; k < 3; ++k) m.size[k] = size[k];
  for (int j = 0; j < 3; ++j) {
    const int k = perm[j];
    if ((flip_mask >> j) & 1) {
      m.mstride[k] = -mst[j];
      m.moff += (int64_t)(msize[j] - 1) * mst[j];
    } else {
      m.mstride[k] = mst[j];
    }
  }
  return m;
}

// member[n, c, a] = x[n, c, i(a)]: one thread per MEMBER element (coalesced writes), gathered reads
__global__ __launch_bounds__(256) void flip_permute_kernel(const float* __restrict__ x, float* __restrict__ y, int NC,
                                                           AxisMap m, int64_t S) {
  // iterate canonical voxels (coalesced reads of x), scatter into the member tensor
  const int64_t total = (int64_t)NC * S;
  const int64_t s12 = (int64_t)m.size[1] * m.size[2];
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t nc = i / S, v = i - nc * S;
    const int i0 = (int)(v / s12), r = (int)(v - i0 * s12);
    const int i1 = r / m.size[2], i2 = r - i1 * m.size[2];
    y[nc * S + m.moff + i0 * m.mstride[0] + i1 * m.mstride[1] + i2 * m.mstride[2]] = x[i];
  }
}

// mode 0: acc[n,c,v] (+)= pred[n,c,a(v)];  mode 1: votes[n, argmax_c pred[n,:,a(v)], v] += 1
__global__ __launch_bounds__(256) void ensemble_accumulate_kernel(const float* __restrict__ pred, float* __restrict__ acc,
                                                                  int32_t* __restrict__ votes, int N, int C, AxisMap m,
                                                                  int64_t S, int mode, int first) {
  const int64_t total = (int64_t)N * S;
  const int64_t s12 = (int64_t)m.size[1] * m.size[2];
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t n = i / S, v = i - n * S;
    const int i0 = (int)(v / s12), r = (int)(v - i0 * s12);
    const int i1 = r / m.size[2], i2 = r - i1 * m.size[2];
    const float* p = pred + n * C * S + m.moff + i0 * m.mstride[0] + i1 * m.mstride[1] + i2 * m.mstride[2];
    if (mode == 0) {
      float* a = acc + n * C * S + v;
      for (int c = 0; c < C; ++c) {
        const float t = p[(int64_t)c * S];
        a[(int64_t)c * S] = first ? t : a[(int64_t)c * S] + t;
      }
    } else {
      int best = 0;
      float bv = p[0];
      for (int c = 1; c < C; ++c) {
        const float t = p[(int64_t)c * S];
        if (t > bv) { bv = t; best = c; }   // first maximum wins, as torch.argmax
      }
      int32_t* q = votes + n * C * S + v;
      if (first) {
        for (int c = 0; c < C; ++c) q[(int64_t)c * S] = c == best ? 1 : 0;
      } else {
        q[(int64_t)best * S] += 1;
      }
    }
  }
}

__global__ __launch_bounds__(256) void ensemble_finalize_kernel(const float* __restrict__ acc,
                                                                const int32_t* __restrict__ votes, float* __restrict__ mean_out,
                                                                int64_t* __restrict__ onehot_out, int N, int C, int64_t S,
                                                                float inv_members, int mode) {
  if (mode == 0) {
    const int64_t total = (int64_t)N * C * S;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) mean_out[i] = acc[i] * inv_members;
    return;
  }
  const int64_t total = (int64_t)N * S;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t n = i / S, v = i - n * S;
    const int32_t* q = votes + n * C * S + v;
    int best = 0, bv = q[0];
    for (int c = 1; c < C; ++c) {
      const int t = q[(int64_t)c * S];
      if (t > bv) { bv = t; best = c; }     // ties: the smallest class index (torch.mode on the CPU)
    }
    int64_t* o = onehot_out + n * C * S + v;
    for (int c = 0; c < C; ++c) o[(int64_t)c * S] = c == best ? 1 : 0;
  }
}

static int check_map_args(const char* who, const int32_t* size3, const int32_t* perm3, int flip_mask) {
  M355_REQUIRE(size3 && perm3, M355_EINVALID_ARG, "%s: null shape / permutation", who);
  M355_REQUIRE(size3[0] > 0 && size3[1] > 0 && size3[2] > 0, M355_EINVALID_ARG, "%s: non-positive size", who);
  int seen = 0;
  for (int j = 0; j < 3; ++j) {
    M355_REQUIRE(perm3[j] >= 0 && perm3[j] <= 2, M355_EINVALID_ARG, "%s: permutation entry %d out of range", who, perm3[j]);
    seen |= 1 << perm3[j];
  }
  M355_REQUIRE(seen == 7, M355_EINVALID_ARG, "%s: not a permutation of (0, 1, 2)", who);
  M355_REQUIRE(flip_mask >= 0 && flip_mask <= 7, M355_EINVALID_ARG, "%s: flip mask out of range", who);
  return M355_OK;
}

static unsigned grid_of(int64_t total) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(total, 256), 16384)); }

}  // namespace m355

using namespace m355;

extern "C" int m355_flip_permute(const float* x, float* member, int32_t N, int32_t C, const int32_t* size3,
                                 const int32_t* perm3, int32_t flip_mask, void* stream) {
  if (int rc = check_map_args("flip_permute", size3, perm3, flip_mask)) return rc;
  M355_REQUIRE(x && member && N > 0 && C > 0, M355_EINVALID_ARG, "flip_permute: null pointer / bad shape");
  const AxisMap m = make_axis_map(size3, perm3, flip_mask);
  const int64_t S = (int64_t)size3[0] * size3[1] * size3[2];
  hipLaunchKernelGGL(flip_permute_kernel, dim3(grid_of((int64_t)N * C * S)), dim3(256), 0, (hipStream_t)stream, x, member,
                     N * C, m, S);
  return check_launch("flip_permute");
}

extern "C" int m355_ensemble_accumulate(const float* pred, float* acc, int32_t* votes, int32_t N, int32_t C,
                                        const int32_t* size3, const int32_t* perm3, int32_t flip_mask, int32_t mode,
                                        int32_t first, void* stream) {
  if (int rc = check_map_args("ensemble_accumulate", size3, perm3, flip_mask)) return rc;
  M355_REQUIRE(pred && N > 0 && C > 0 && (mode == 0 || mode == 1), M355_EINVALID_ARG, "ensemble_accumulate: bad arguments");
  M355_REQUIRE(mode == 0 ? acc != nullptr : votes != nullptr, M355_EINVALID_ARG, "ensemble_accumulate: null accumulator");
  const AxisMap m = make_axis_map(size3, perm3, flip_mask);
  const int64_t S = (int64_t)size3[0] * size3[1] * size3[2];
  hipLaunchKernelGGL(ensemble_accumulate_kernel, dim3(grid_of((int64_t)N * S)), dim3(256), 0, (hipStream_t)stream, pred,
                     acc, votes, N, C, m, S, mode, first);
  return check_launch("ensemble_accumulate");
}

extern "C" int m355_ensemble_finalize(const float* acc, const int32_t* votes, float* mean_out, int64_t* onehot_out,
                                      int32_t N, int32_t C, int64_t S, int32_t members, int32_t mode, void* stream) {
  M355_REQUIRE(N > 0 && C > 0 && S > 0 && members > 0 && (mode == 0 || mode == 1), M355_EINVALID_ARG,
               "ensemble_finalize: bad arguments");
  M355_REQUIRE(mode == 0 ? (acc && mean_out) : (votes && onehot_out), M355_EINVALID_ARG, "ensemble_finalize: null pointer");
  const int64_t total = mode == 0 ? (int64_t)N * C * S : (int64_t)N * S;
  hipLaunchKernelGGL(ensemble_finalize_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, acc, votes,
                     mean_out, onehot_out, N, C, S, 1.0f / (float)members, mode);
  return check_launch("ensemble_finalize");
}
