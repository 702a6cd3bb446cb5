// Row-wise and elementwise kernels of the encoder path: LayerNorm (reference modeling:860-865,
// 878-880, 1251, 1138), patch extraction for the conv-as-GEMM patch embedding (modeling:329-350),
// operand splitting for the bf16x3 mode, and time-embedding row selection (modeling:435-450).
// All of them are HBM-bound: one wave per row, 16-byte vector accesses, wave-shuffle reductions.
#include "sf_common.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row lives in registers (D <= 64*4*MAXV), two-pass statistics.
// ------------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void sf_layernorm_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ y_f32,
                                                           bf16_t* __restrict__ y_hi,
                                                           bf16_t* __restrict__ y_lo, int rows, int D,
                                                           float eps, const bf16_t* xp_hi, const bf16_t* xp_lo, const bf16_t* xp_lo2,
                                                           float* const* __restrict__ y_f32_ind) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (y_f32_ind) y_f32 = *y_f32_ind;
  const int nv = D >> 2;                       // float4 per row
  const f32x4_t* xr = reinterpret_cast<const f32x4_t*>(x + (size_t)row * D);
  // xp_hi / xp_lo: the row arrives as hi + lo bf16 planes (residual stream of the BASELINE-sized bf16 forward) instead of x;
  // y_hi may be xp_hi itself: the row is in registers before the first store
  const u32x2_t* xh = reinterpret_cast<const u32x2_t*>(xp_hi + (size_t)row * D);
  const u32x2_t* xl = reinterpret_cast<const u32x2_t*>(xp_lo + (size_t)row * D);
  const u32x2_t* xl2 = reinterpret_cast<const u32x2_t*>(xp_lo2 + (size_t)row * D);
  f32x4_t v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    if (c < nv) {
      if (xp_hi) {
        const u32x2_t h = xh[c], l = xl[c];
        v[i] = (f32x4_t){bf2f(h[0] & 0xffffu) + bf2f(l[0] & 0xffffu), bf2f(h[0] >> 16) + bf2f(l[0] >> 16),
                         bf2f(h[1] & 0xffffu) + bf2f(l[1] & 0xffffu), bf2f(h[1] >> 16) + bf2f(l[1] >> 16)};
        if (xp_lo2) {
          const u32x2_t l2 = xl2[c];
          v[i] += (f32x4_t){bf2f(l2[0] & 0xffffu), bf2f(l2[0] >> 16), bf2f(l2[1] & 0xffffu), bf2f(l2[1] >> 16)};
        }
      } else {
        v[i] = xr[c];
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wave_sum_dpp(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    if (c < nv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum_dpp(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    if (c < nv) {
      const f32x4_t g = reinterpret_cast<const f32x4_t*>(gamma)[c];
      const f32x4_t b = reinterpret_cast<const f32x4_t*>(beta)[c];
      f32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      const size_t off = (size_t)row * D + (size_t)c * 4;
      if (y_f32) *reinterpret_cast<f32x4_t*>(y_f32 + off) = o;
      if (y_hi) {
        unsigned int h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_bf(o[j], h[j], l[j]);
        *reinterpret_cast<u32x2_t*>(y_hi + off) = (u32x2_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        if (y_lo) *reinterpret_cast<u32x2_t*>(y_lo + off) = (u32x2_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
      }
    }
  }
}

hipError_t sf_launch_layernorm(const float* x, const float* gamma, const float* beta, float* y_f32,
                               bf16_t* y_hi, bf16_t* y_lo, int rows, int D, float eps, hipStream_t s, const bf16_t* xp_hi,
                               const bf16_t* xp_lo, const bf16_t* xp_lo2, float* const* y_f32_ind) {
  if (rows <= 0) return hipSuccess;
  if (D % 4 || D > 64 * 4 * 16) return hipErrorInvalidValue;
  const dim3 grid((rows + 3) / 4), block(256);
  const int nv = (D / 4 + 63) / 64;
  if (nv <= 1) hipLaunchKernelGGL(sf_layernorm_kernel<1>, grid, block, 0, s, x, gamma, beta, y_f32, y_hi, y_lo, rows, D, eps, xp_hi, xp_lo, xp_lo2, y_f32_ind);
  else if (nv <= 3) hipLaunchKernelGGL(sf_layernorm_kernel<3>, grid, block, 0, s, x, gamma, beta, y_f32, y_hi, y_lo, rows, D, eps, xp_hi, xp_lo, xp_lo2, y_f32_ind);
  else if (nv <= 8) hipLaunchKernelGGL(sf_layernorm_kernel<8>, grid, block, 0, s, x, gamma, beta, y_f32, y_hi, y_lo, rows, D, eps, xp_hi, xp_lo, xp_lo2, y_f32_ind);
  else hipLaunchKernelGGL(sf_layernorm_kernel<16>, grid, block, 0, s, x, gamma, beta, y_f32, y_hi, y_lo, rows, D, eps, xp_hi, xp_lo, xp_lo2, y_f32_ind);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// patchify: pixels [F,C,H,W] -> A[F*N, C*P*P] (bf16 hi/lo), column = (c*P + ph)*P + pw,
// patch n = prow*(W/P) + pcol.  One thread moves 8 consecutive pw pixels (16 B of bf16 out).
// IN = 0 fp32, 1 bf16 (already normalised frames), 2 uint8 raw frames: the image processor's
// rescale + normalize (vqa_enc:1436-1447: x / 255, then (x - mean) / std per channel) is fused here as
// one FMA per pixel, so frames cross PCIe / HBM as bytes.
// ------------------------------------------------------------------------------------------------
template <int IN>
__global__ __launch_bounds__(256) void sf_patchify_kernel(const void* __restrict__ pixels,
                                                          bf16_t* __restrict__ out_hi,
                                                          bf16_t* __restrict__ out_lo, int F, int C, int H,
                                                          int W, int P, int gh, int gw, SfPixelNorm norm,
                                                          const SfStreamParams* __restrict__ sp, SfStreamParams* sp_write,
                                                          SfStreamParams sp_value) {
  if (sp) pixels = sp->pixels;
  if (sp_write && blockIdx.x == 0 && threadIdx.x == 0) *sp_write = sp_value;
  const int Kp = C * P * P;
  const int chunks_per_row = Kp >> 3;
  const size_t total = (size_t)F * gh * gw * chunks_per_row;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ck = (int)(i % chunks_per_row);
    const size_t prow = i / chunks_per_row;          // global patch row = f*N + n
    const int n = (int)(prow % (gh * gw));
    const int f = (int)(prow / (gh * gw));
    const int col = ck << 3;
    const int c = col / (P * P), rem = col % (P * P);
    const int ph = rem / P, pw = rem % P;
    const int y = (n / gw) * P + ph, x0 = (n % gw) * P + pw;
    const size_t src = (((size_t)f * C + c) * H + y) * W + x0;
    float v[8];
    if (IN == 2) {
      const u32x2_t raw = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned char*>(pixels) + src);
      const float sc = norm.scale[c & 3], sh = norm.shift[c & 3];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf((float)((raw[j >> 2] >> ((j & 3) * 8)) & 0xffu), sc, sh);
    } else if (IN == 1) {
      const bf16_t* pp = reinterpret_cast<const bf16_t*>(pixels) + src;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = bf2f(pp[j]);
    } else {
      const float* pp = reinterpret_cast<const float*>(pixels) + src;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = pp[j];
    }
    unsigned int h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf(v[j], h[j], l[j]);
    const size_t o = prow * Kp + col;
    *reinterpret_cast<u32x4_t*>(out_hi + o) =
        (u32x4_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    if (out_lo)
      *reinterpret_cast<u32x4_t*>(out_lo + o) =
          (u32x4_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
  }
}

// Any patch size / padded patch vectors (14 x 14 patches: K = 588 -> 640): one thread per 8 output columns, element by element;
// columns past C * P * P are zeros (the patch-embedding weight is zero-padded to the same width at upload).
template <int IN>
__global__ __launch_bounds__(256) void sf_patchify_generic_kernel(const void* __restrict__ pixels, bf16_t* __restrict__ out_hi,
                                                                  bf16_t* __restrict__ out_lo, int F, int C, int H, int W, int P, int gh, int gw,
                                                                  int Kpad, SfPixelNorm norm, const SfStreamParams* __restrict__ sp,
                                                                  SfStreamParams* sp_write, SfStreamParams sp_value) {
  if (sp) pixels = sp->pixels;
  if (sp_write && blockIdx.x == 0 && threadIdx.x == 0) *sp_write = sp_value;
  const int Kr = C * P * P, chunks_per_row = Kpad >> 3;
  const size_t total = (size_t)F * gh * gw * chunks_per_row;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ck = (int)(i % chunks_per_row);
    const size_t prow = i / chunks_per_row;
    const int n = (int)(prow % (gh * gw)), f = (int)(prow / (gh * gw));
    unsigned int h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = ck * 8 + j;
      float v = 0.f;
      if (col < Kr) {
        const int c = col / (P * P), rem = col % (P * P);
        const size_t src = (((size_t)f * C + c) * H + (n / gw) * P + rem / P) * W + (n % gw) * P + rem % P;
        if (IN == 2) v = fmaf((float)reinterpret_cast<const unsigned char*>(pixels)[src], norm.scale[c & 3], norm.shift[c & 3]);
        else if (IN == 1) v = bf2f(reinterpret_cast<const bf16_t*>(pixels)[src]);
        else v = reinterpret_cast<const float*>(pixels)[src];
      }
      split_bf(v, h[j], l[j]);
    }
    const size_t o = prow * Kpad + (size_t)ck * 8;
    *reinterpret_cast<u32x4_t*>(out_hi + o) = (u32x4_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    if (out_lo) *reinterpret_cast<u32x4_t*>(out_lo + o) = (u32x4_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
  }
}

hipError_t sf_launch_patchify(const void* pixels, int pixel_kind, bf16_t* out_hi, bf16_t* out_lo,
                              int F, int C, int H, int W, int P, hipStream_t s, const SfPixelNorm* pnorm, const SfStreamParams* sp,
                              SfStreamParams* sp_write, const SfStreamParams* sp_value, int Kpad) {
  SfStreamParams spv = {};
  if (sp_write) { if (!sp_value) return hipErrorInvalidValue; spv = *sp_value; }
  SfPixelNorm norm;
  for (int i = 0; i < 4; ++i) { norm.scale[i] = 1.0f / 127.5f; norm.shift[i] = -1.0f; }    // mean = std = 0.5, rescale 1/255
  if (pnorm) norm = *pnorm;
  if (Kpad <= 0) Kpad = C * P * P;
  if (P % 8 || Kpad != C * P * P || (pixel_kind == 2 && W % 8)) {      // generic path
    if (Kpad % 8 || Kpad < C * P * P || (pixel_kind == 2 && C > 4)) return hipErrorInvalidValue;
    const int gh = H / P, gw = W / P;
    const size_t total = (size_t)F * gh * gw * (Kpad / 8);
    if (!total) return hipSuccess;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (pixel_kind == 2) hipLaunchKernelGGL(sf_patchify_generic_kernel<2>, dim3(blocks), dim3(256), 0, s, pixels, out_hi, out_lo, F, C, H, W, P, gh, gw, Kpad, norm, sp, sp_write, spv);
    else if (pixel_kind == 1) hipLaunchKernelGGL(sf_patchify_generic_kernel<1>, dim3(blocks), dim3(256), 0, s, pixels, out_hi, out_lo, F, C, H, W, P, gh, gw, Kpad, norm, sp, sp_write, spv);
    else hipLaunchKernelGGL(sf_patchify_generic_kernel<0>, dim3(blocks), dim3(256), 0, s, pixels, out_hi, out_lo, F, C, H, W, P, gh, gw, Kpad, norm, sp, sp_write, spv);
    return hipGetLastError();
  }
  if (pixel_kind == 2 && C > 4) return hipErrorInvalidValue;
  const int gh = H / P, gw = W / P;
  const size_t total = (size_t)F * gh * gw * (C * P * P / 8);
  if (!total) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (pixel_kind == 2)
    hipLaunchKernelGGL(sf_patchify_kernel<2>, dim3(blocks), dim3(256), 0, s, pixels, out_hi, out_lo, F, C, H, W, P, gh, gw, norm, sp, sp_write, spv);
  else if (pixel_kind == 1)
    hipLaunchKernelGGL(sf_patchify_kernel<1>, dim3(blocks), dim3(256), 0, s, pixels, out_hi, out_lo, F, C, H, W, P, gh, gw, norm, sp, sp_write, spv);
  else
    hipLaunchKernelGGL(sf_patchify_kernel<0>, dim3(blocks), dim3(256), 0, s, pixels, out_hi, out_lo, F, C, H, W, P, gh, gw, norm, sp, sp_write, spv);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_split_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi,
                                                       bf16_t* __restrict__ lo, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4_t v = reinterpret_cast<const f32x4_t*>(x)[i];
    unsigned int h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_bf(v[j], h[j], l[j]);
    reinterpret_cast<u32x2_t*>(hi)[i] = (u32x2_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
    if (lo) reinterpret_cast<u32x2_t*>(lo)[i] = (u32x2_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
  }
}

hipError_t sf_launch_split(const float* x, bf16_t* hi, bf16_t* lo, size_t n, hipStream_t s) {
  if (n % 4) return hipErrorInvalidValue;
  if (!n) return hipSuccess;
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(sf_split_kernel, dim3(blocks), dim3(256), 0, s, x, hi, lo, n4);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// LN-fold entry: bf16 copy of the rows + {sum x, sum x^2} (used once per forward, on the embeddings)
__global__ __launch_bounds__(256) void sf_rowstats_cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ xb,
                                                               bf16_t* __restrict__ xlo, bf16_t* __restrict__ xlo2, float* __restrict__ stats, int rows, int D, int wide) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = D >> 2;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < nv; c += 64) {
    const f32x4_t v = reinterpret_cast<const f32x4_t*>(x + (size_t)row * D)[c];
    s1 += (v[0] + v[1]) + (v[2] + v[3]);
    s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    const u32x2_t h = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *reinterpret_cast<u32x2_t*>(xb + (size_t)row * D + (size_t)c * 4) = h;
    if (xlo) {
      const float r0 = v[0] - bf2f(h[0] & 0xffffu), r1 = v[1] - bf2f(h[0] >> 16), r2 = v[2] - bf2f(h[1] & 0xffffu), r3 = v[3] - bf2f(h[1] >> 16);
      const u32x2_t l = {pack_bf2(r0, r1), pack_bf2(r2, r3)};
      *reinterpret_cast<u32x2_t*>(xlo + (size_t)row * D + (size_t)c * 4) = l;
      if (xlo2)
        *reinterpret_cast<u32x2_t*>(xlo2 + (size_t)row * D + (size_t)c * 4) =
            (u32x2_t){pack_bf2(r0 - bf2f(l[0] & 0xffffu), r1 - bf2f(l[0] >> 16)), pack_bf2(r2 - bf2f(l[1] & 0xffffu), r3 - bf2f(l[1] >> 16))};
    }
  }
  s1 = wave_sum_dpp(s1);
  s2 = wave_sum_dpp(s2);
  if (lane == 0) {          // wide rows ([8]: this pair + empty ones): the accurate mode, and the bf16 mode's four-pair layout
    *reinterpret_cast<f32x4_t*>(stats + (size_t)row * (wide ? 8 : 4)) = (f32x4_t){s1, s2, 0.f, 0.f};
    if (wide) *reinterpret_cast<f32x4_t*>(stats + (size_t)row * 8 + 4) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
}

hipError_t sf_launch_rowstats_cast(const float* x, bf16_t* xb, float* stats, int rows, int D, hipStream_t s, bf16_t* xlo, bf16_t* xlo2, int wide) {
  if (rows <= 0) return hipSuccess;
  if (D % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_rowstats_cast_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, xb, xlo, xlo2, stats, rows, D, (wide || xlo) ? 1 : 0);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_gather_rows_kernel(const float* __restrict__ table,
                                                             float* __restrict__ out, SfRowIndex idx, int D,
                                                             const int* __restrict__ base_dev) {
  const int t = blockIdx.x;
  const float* src = table + (size_t)(idx.idx[t] + (base_dev ? *base_dev : 0)) * D;
  for (int i = threadIdx.x; i < D; i += blockDim.x) out[(size_t)t * D + i] = src[i];
}

hipError_t sf_launch_gather_rows(const float* table, float* out, const SfRowIndex& idx, int D, hipStream_t s, const int* base_dev) {
  if (idx.n <= 0 || idx.n > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_gather_rows_kernel, dim3(idx.n), dim3(256), 0, s, table, out, idx, D, base_dev);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// position + time embedding table: out[(t, n), :] = pos[n, :] + time_rows[t, :]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_pos_time_table_kernel(const float* __restrict__ pos, const float* __restrict__ te,
                                                                float* __restrict__ out, int T, int N, int D4) {
  const size_t total = (size_t)T * N * D4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % D4);
    const size_t row = i / D4;
    const int n = (int)(row % N), t = (int)(row / N);
    reinterpret_cast<f32x4_t*>(out)[i] = reinterpret_cast<const f32x4_t*>(pos)[(size_t)n * D4 + c] + reinterpret_cast<const f32x4_t*>(te)[(size_t)t * D4 + c];
  }
}
hipError_t sf_launch_pos_time_table(const float* pos, const float* time_rows, float* out, int T, int N, int D, hipStream_t s) {
  if (D % 4) return hipErrorInvalidValue;
  const size_t total = (size_t)T * N * (D / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(sf_pos_time_table_kernel, dim3(blocks), dim3(256), 0, s, pos, time_rows, out, T, N, D / 4);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void sf_copy2_kernel(const float* __restrict__ a_src, float* __restrict__ a_dst, size_t na4,
                                                       const float* __restrict__ b_src, float* __restrict__ b_dst, size_t nb4,
                                                       const SfStreamParams* __restrict__ sp) {
  if (sp) { a_dst = sp->lhs; b_dst = sp->pooler; }
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < na4) reinterpret_cast<f32x4_t*>(a_dst)[i] = reinterpret_cast<const f32x4_t*>(a_src)[i];
  else if (i - na4 < nb4) reinterpret_cast<f32x4_t*>(b_dst)[i - na4] = reinterpret_cast<const f32x4_t*>(b_src)[i - na4];
}
__global__ void sf_stream_params_kernel(SfStreamParams* dst, SfStreamParams v) {
  if (threadIdx.x == 0) *dst = v;
}
hipError_t sf_launch_stream_params(SfStreamParams* dst, const SfStreamParams& v, hipStream_t s) {
  hipLaunchKernelGGL(sf_stream_params_kernel, dim3(1), dim3(64), 0, s, dst, v);
  return hipGetLastError();
}

hipError_t sf_launch_copy2(const float* a_src, float* a_dst, size_t na, const float* b_src, float* b_dst, size_t nb, hipStream_t s,
                           const SfStreamParams* sp) {
  if ((na % 4) || (nb % 4)) return hipErrorInvalidValue;
  if (!b_src || (!b_dst && !sp)) nb = 0;
  const size_t n4 = (na + nb) / 4;
  if (!n4) return hipSuccess;
  hipLaunchKernelGGL(sf_copy2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a_src, a_dst, na / 4, b_src, b_dst, nb / 4, sp);
  return hipGetLastError();
}
