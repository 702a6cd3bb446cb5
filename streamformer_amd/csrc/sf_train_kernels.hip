// Row-wise, elementwise and parameter-side kernels of the training step:
// GELU forward/backward (modeling:819-824), LayerNorm backward (nn.LayerNorm at modeling:860-865,
// 878-880, 1251, 1138), bias / embedding-table gradient reductions, fp32-master -> bf16 working
// weights (LoRA merge modeling:519-573, temporal gate modeling:954-958), and the fused AdamW update
// (torch.optim.AdamW as configured by optim_factory.py:59-104).  All HBM-bound; every reduction is
// two-stage with a fixed order, so gradients are bit-reproducible run to run.
#include "sf_train.h"
#include "sf_switches.h"

// ------------------------------------------------------------------------------------------------
// GELU
// ------------------------------------------------------------------------------------------------
SF_DEVICE void unpack8(const u32x4_t v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf2f(v[i] & 0xffffu);
    f[2 * i + 1] = bf2f(v[i] >> 16);
  }
}
SF_DEVICE u32x4_t pack8(const float* f) {
  u32x4_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return v;
}

__global__ __launch_bounds__(256) void sf_gelu_fwd_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ act,
                                                          size_t nchunks) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
    float f[8];
    unpack8(reinterpret_cast<const u32x4_t*>(pre)[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = gelu_erf(f[j]);
    reinterpret_cast<u32x4_t*>(act)[i] = pack8(f);
  }
}

__global__ __launch_bounds__(256) void sf_gelu_bwd_kernel(bf16_t* __restrict__ d, const bf16_t* __restrict__ pre,
                                                          size_t nchunks) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
    float x[8], g[8];
    unpack8(reinterpret_cast<const u32x4_t*>(pre)[i], x);
    unpack8(reinterpret_cast<const u32x4_t*>(d)[i], g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
      const float cdf = 0.5f * (1.0f + erff(x[j] * 0.70710678118654752440f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * x[j] * x[j]);
      g[j] *= cdf + x[j] * pdf;
    }
    reinterpret_cast<u32x4_t*>(d)[i] = pack8(g);
  }
}

static int ew_grid(size_t nchunks) {
  size_t b = (nchunks + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ------------------------------------------------------------------------------------------------
// drop_path (stochastic depth, modeling:460-486, 846-856): the per-"sample" keep / drop factor of a residual branch, 0 or
// 1 / keep_prob, where the reference's sample axis is dim 0 of the tensor the branch returns:
//   mode 0 temporal (B*N, T, D)  -> group = b * N + n      mode 1 spatial (B*T, N, D) -> group = b * T + t = row / N
//   mode 2 MLP      (B, N*T, D)  -> group = b = row / (T * N)                (token row = (b * T + t) * N + n)
// ------------------------------------------------------------------------------------------------
SF_DEVICE int dp_group(int row, int mode, int T, int N) {
  if (mode == 1) return row / N;
  if (mode == 2) return row / (T * N);
  return (row / (T * N)) * N + row % N;
}
__global__ __launch_bounds__(256) void sf_rowscale_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                               const float* __restrict__ scales, size_t nchunks, int D8, int mode, int T, int N, SfDrop d) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / D8);
    const float sc = scales ? scales[dp_group(row, mode, T, N)] : 1.f;
    const u32x4_t v = reinterpret_cast<const u32x4_t*>(in)[i];
    u32x4_t o;
    const unsigned e0 = (unsigned)(i * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float f0 = d.on ? sc * sf_drop_factor(d, e0 + 2 * j) : sc, f1 = d.on ? sc * sf_drop_factor(d, e0 + 2 * j + 1) : sc;
      o[j] = pack_bf2(bf2f(v[j] & 0xffffu) * f0, __uint_as_float(v[j] & 0xffff0000u) * f1);
    }
    reinterpret_cast<u32x4_t*>(out)[i] = o;
  }
}
hipError_t sf_launch_rowscale_bf16(const bf16_t* in, bf16_t* out, const float* scales, int rows, int D, int mode, int T, int N, hipStream_t s, SfDrop drop) {
  if (D % 8) return hipErrorInvalidValue;
  const size_t n = (size_t)rows * (D / 8);
  if (!n) return hipSuccess;
  if (n * 8 >= ((size_t)1 << 32)) return hipErrorInvalidValue;       // mask indices are 32-bit
  hipLaunchKernelGGL(sf_rowscale_bf16_kernel, dim3(ew_grid(n)), dim3(256), 0, s, in, out, scales, n, D / 8, mode, T, N, drop);
  return hipGetLastError();
}
// out = resid + scale[group(row)] * mask * y        (out may alias resid)
__global__ __launch_bounds__(256) void sf_resid_rowscale_kernel(float* __restrict__ out, const float* __restrict__ resid, const float* __restrict__ y,
                                                                const float* __restrict__ scales, size_t nchunks, int D4, int mode, int T, int N, SfDrop d) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / D4);
    const float sc = scales ? scales[dp_group(row, mode, T, N)] : 1.f;
    f32x4_t yv = reinterpret_cast<const f32x4_t*>(y)[i];
    if (d.on) {
      const unsigned e0 = (unsigned)(i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) yv[j] *= sf_drop_factor(d, e0 + j);
    }
    reinterpret_cast<f32x4_t*>(out)[i] = reinterpret_cast<const f32x4_t*>(resid)[i] + sc * yv;
  }
}
hipError_t sf_launch_resid_rowscale(float* out, const float* resid, const float* y, const float* scales, int rows, int D, int mode, int T, int N,
                                    hipStream_t s, SfDrop drop) {
  if (D % 4) return hipErrorInvalidValue;
  const size_t n = (size_t)rows * (D / 4);
  if (!n) return hipSuccess;
  if (n * 4 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_resid_rowscale_kernel, dim3(ew_grid(n)), dim3(256), 0, s, out, resid, y, scales, n, D / 4, mode, T, N, drop);
  return hipGetLastError();
}
// h = m_time o (m_pos o h + time[t])   (modeling:374, 378; rows (b, t, n))
__global__ __launch_bounds__(256) void sf_embed_dropout_kernel(float* __restrict__ h, const float* __restrict__ te, size_t nchunks, int D4, int T, int N,
                                                               SfDrop dpos, SfDrop dtime) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / D4), c = (int)(i % D4);
    const int t = (row / N) % T;
    f32x4_t v = reinterpret_cast<f32x4_t*>(h)[i];
    const f32x4_t tv = reinterpret_cast<const f32x4_t*>(te)[(size_t)t * D4 + c];
    const unsigned e0 = (unsigned)(i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = sf_drop_factor(dtime, e0 + j) * (sf_drop_factor(dpos, e0 + j) * v[j] + tv[j]);
    reinterpret_cast<f32x4_t*>(h)[i] = v;
  }
}
hipError_t sf_launch_embed_dropout(float* h, const float* time_rows, int M, int D, int T, int N, SfDrop pos_drop, SfDrop time_drop, hipStream_t s) {
  if (D % 4 || !pos_drop.on || !time_drop.on) return hipErrorInvalidValue;
  const size_t n = (size_t)M * (D / 4);
  if (!n) return hipSuccess;
  if (n * 4 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_embed_dropout_kernel, dim3(ew_grid(n)), dim3(256), 0, s, h, time_rows, n, D / 4, T, N, pos_drop, time_drop);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void sf_dropout_f32_kernel(float* __restrict__ g, bf16_t* __restrict__ g_bf, size_t nchunks, SfDrop d) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
    f32x4_t v = reinterpret_cast<f32x4_t*>(g)[i];
    const unsigned e0 = (unsigned)(i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= sf_drop_factor(d, e0 + j);
    reinterpret_cast<f32x4_t*>(g)[i] = v;
    if (g_bf) reinterpret_cast<u32x2_t*>(g_bf)[i] = (u32x2_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
  }
}
hipError_t sf_launch_dropout_f32(float* g, bf16_t* g_bf, size_t n, SfDrop drop, hipStream_t s) {
  if (n % 4 || !drop.on) return hipErrorInvalidValue;
  if (!n) return hipSuccess;
  if (n >= ((size_t)1 << 32)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_dropout_f32_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, s, g, g_bf, n / 4, drop);
  return hipGetLastError();
}

hipError_t sf_launch_gelu_fwd(const bf16_t* pre, bf16_t* act, size_t n, hipStream_t s) {
  if (n % 8) return hipErrorInvalidValue;
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(sf_gelu_fwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, s, pre, act, n / 8);
  return hipGetLastError();
}
hipError_t sf_launch_gelu_bwd(bf16_t* d, const bf16_t* pre, size_t n, hipStream_t s) {
  if (n % 8) return hipErrorInvalidValue;
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(sf_gelu_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, s, d, pre, n / 8);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row (row in registers), waves walk rows with a grid stride and
// keep per-lane column sums of dy*xhat / dy; one partial row pair per block, then a column reduce.
// ------------------------------------------------------------------------------------------------
#define LN_BWD_MAX_BLOCKS 2048      // partial-sum rows the workspace holds; the launch uses ln_bwd_blocks() of them

template <int MAXV, bool DYB>
__global__ __launch_bounds__(256) void sf_ln_bwd_kernel(const float* __restrict__ x, const void* __restrict__ dy,
                                                        const float* __restrict__ gamma, const float* g_in,
                                                        float* g_out, bf16_t* __restrict__ g_out_bf,
                                                        float* __restrict__ partial, int rows, int D, float eps) {
  extern __shared__ float red[];                      // [3][2][D] for waves 1..3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = D >> 2;
  f32x4_t ag[MAXV], ab[MAXV], gm[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    ag[i] = ab[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int c = i * 64 + lane;
    gm[i] = c < nv ? reinterpret_cast<const f32x4_t*>(gamma)[c] : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const float inv_d = 1.0f / (float)D;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const f32x4_t* xr = reinterpret_cast<const f32x4_t*>(x + (size_t)row * D);
    const f32x4_t* dr = reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(dy) + (size_t)row * D);
    const u32x2_t* db = reinterpret_cast<const u32x2_t*>(reinterpret_cast<const bf16_t*>(dy) + (size_t)row * D);
    f32x4_t v[MAXV], d[MAXV], gi[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = i * 64 + lane;
      gi[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (c < nv) {
        v[i] = xr[c];
        if (g_in) gi[i] = reinterpret_cast<const f32x4_t*>(g_in + (size_t)row * D)[c];      // with the row's other loads, not behind its four reductions
        if (DYB) {
          const u32x2_t t = db[c];
          d[i] = (f32x4_t){bf2f(t[0] & 0xffffu), bf2f(t[0] >> 16), bf2f(t[1] & 0xffffu), bf2f(t[1] >> 16)};
        } else {
          d[i] = dr[c];
        }
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      } else {
        v[i] = d[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mean = wave_sum_dpp(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = i * 64 + lane;
      if (c < nv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = v[i][j] - mean;
          q += t * t;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum_dpp(q) * inv_d + eps);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = i * 64 + lane;
      if (c < nv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (v[i][j] - mean) * rstd;
          const float dg = d[i][j] * gm[i][j];
          v[i][j] = xh;                  // keep xhat
          ag[i][j] += d[i][j] * xh;
          ab[i][j] += d[i][j];
          d[i][j] = dg;                  // keep dy * gamma
          c1 += dg;
          c2 += dg * xh;
        }
      }
    }
    c1 = wave_sum_dpp(c1) * inv_d;
    c2 = wave_sum_dpp(c2) * inv_d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = i * 64 + lane;
      if (c < nv) {
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (d[i][j] - c1 - v[i][j] * c2);
        const size_t off = (size_t)row * D + (size_t)c * 4;
        o += gi[i];
        *reinterpret_cast<f32x4_t*>(g_out + off) = o;
        if (g_out_bf) *reinterpret_cast<u32x2_t*>(g_out_bf + off) = (u32x2_t){pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
      }
    }
  }
  // block reduce of the column sums (waves 1..3 -> LDS -> wave 0), fixed order
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = i * 64 + lane;
      if (c < nv) {
        *reinterpret_cast<f32x4_t*>(red + ((wave - 1) * 2 + 0) * D + c * 4) = ag[i];
        *reinterpret_cast<f32x4_t*>(red + ((wave - 1) * 2 + 1) * D + c * 4) = ab[i];
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = i * 64 + lane;
      if (c < nv) {
        for (int w = 0; w < 3; ++w) {
          ag[i] += *reinterpret_cast<const f32x4_t*>(red + (w * 2 + 0) * D + c * 4);
          ab[i] += *reinterpret_cast<const f32x4_t*>(red + (w * 2 + 1) * D + c * 4);
        }
        *reinterpret_cast<f32x4_t*>(partial + ((size_t)blockIdx.x * 2 + 0) * D + c * 4) = ag[i];
        *reinterpret_cast<f32x4_t*>(partial + ((size_t)blockIdx.x * 2 + 1) * D + c * 4) = ab[i];
      }
    }
  }
}

// d_gamma[c] += sum_b partial[b][0][c], d_beta[c] += sum_b partial[b][1][c]: 64 columns per block,
// the sixteen waves split the partial rows, fixed combination order
__global__ __launch_bounds__(1024) void sf_ln_bwd_finish_kernel(const float* __restrict__ partial, int nblocks, int D,
                                                                float* d_gamma, float* d_beta) {
  __shared__ float red[2][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a = 0.f, b = 0.f;
  if (c < D) {
#pragma unroll 8
    for (int i = wave; i < nblocks; i += 16) {      // unrolled: the loads of eight partial rows are in flight together
      a += partial[((size_t)i * 2 + 0) * D + c];
      b += partial[((size_t)i * 2 + 1) * D + c];
    }
  }
  red[0][wave][lane] = a;
  red[1][wave][lane] = b;
  __syncthreads();
  if (wave == 0 && c < D) {
    a = b = 0.f;
    for (int w = 0; w < 16; ++w) { a += red[0][w][lane]; b += red[1][w][lane]; }
    if (d_gamma) d_gamma[c] += a;
    if (d_beta) d_beta[c] += b;
  }
}

size_t sf_ln_bwd_partial_floats(int D) { return (size_t)LN_BWD_MAX_BLOCKS * 2 * D; }

hipError_t sf_launch_ln_bwd(const float* x, const void* dy, int dy_is_bf16, const float* gamma, const float* g_in, float* g_out,
                            bf16_t* g_out_bf, float* d_gamma, float* d_beta, float* partial, int rows, int D, float eps,
                            hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (D % 4 || D > 64 * 4 * 8) return hipErrorInvalidValue;
  // 512 workgroups = 8 waves per CU run at the HBM rate (fp32 dy, M = 25 088: 49 us = 6.2 TB/s; 1024 .. 2048 workgroups measure 50 .. 54 us
  // and a slower finish kernel, profiles/r04_ln_bwd_lab.txt).  SF_LN_BWD_BLOCKS overrides (lab).
  const int cap = sf_sw(SW_LN_BWD_BLOCKS) ? atoi(sf_sw(SW_LN_BWD_BLOCKS)) : 512;
  int blocks = (rows + 3) / 4;
  if (blocks > cap) blocks = cap;
  if (blocks > LN_BWD_MAX_BLOCKS) blocks = LN_BWD_MAX_BLOCKS;
  const size_t lds = (size_t)3 * 2 * D * sizeof(float);
  const int nv = (D / 4 + 63) / 64;
#define SF_LNB(MV)                                                                                                                     \
  do {                                                                                                                               \
    if (dy_is_bf16) hipLaunchKernelGGL((sf_ln_bwd_kernel<MV, true>), dim3(blocks), dim3(256), lds, s, x, dy, gamma, g_in, g_out, g_out_bf, partial, rows, D, eps); \
    else hipLaunchKernelGGL((sf_ln_bwd_kernel<MV, false>), dim3(blocks), dim3(256), lds, s, x, dy, gamma, g_in, g_out, g_out_bf, partial, rows, D, eps);           \
  } while (0)
  if (nv <= 1) SF_LNB(1);
  else if (nv <= 3) SF_LNB(3);
  else SF_LNB(8);
#undef SF_LNB
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (d_gamma || d_beta) {
    hipLaunchKernelGGL(sf_ln_bwd_finish_kernel, dim3((D + 63) / 64), dim3(1024), 0, s, partial, blocks, D, d_gamma, d_beta);
    e = hipGetLastError();
  }
  return e;
}

// ------------------------------------------------------------------------------------------------
// column sums of a bf16 matrix (bias gradients)
// ------------------------------------------------------------------------------------------------
#define CS_MAX_CHUNKS 1024

__global__ __launch_bounds__(256) void sf_colsum_bf16_kernel(const bf16_t* __restrict__ x, int rows, int cols, int ld,
                                                             int rows_per_chunk, float* __restrict__ partial) {
  __shared__ float red[8][32][9];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = (blockIdx.x * 32 + tx) * 8;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c0 < cols) {
    for (int r = r0 + ty; r < r1; r += 8) {
      float f[8];
      unpack8(*reinterpret_cast<const u32x4_t*>(x + (size_t)r * ld + c0), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx][j] = acc[j];
  __syncthreads();
  if (ty == 0 && c0 < cols) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = 0.f;
      for (int y = 0; y < 8; ++y) t += red[y][tx][j];
      if (c0 + j < cols) partial[(size_t)blockIdx.y * cols + c0 + j] = t;
    }
  }
}

__global__ __launch_bounds__(256) void sf_colsum_finish_kernel(const float* __restrict__ partial, int nchunks, int cols,
                                                               float alpha, float* out, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float t = 0.f;
  if (c < cols)
    for (int i = wave; i < nchunks; i += 4) t += partial[(size_t)i * cols + c];
  red[wave][lane] = t;
  __syncthreads();
  if (wave == 0 && c < cols) {
    t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    out[c] = (accumulate ? out[c] : 0.f) + alpha * t;
  }
}

size_t sf_colsum_partial_floats(int cols) { return (size_t)CS_MAX_CHUNKS * cols; }

hipError_t sf_launch_colsum_bf16(const bf16_t* x, int rows, int cols, int ld, float alpha, float* out, int accumulate,
                                 float* partial, hipStream_t s) {
  if (rows <= 0 || cols <= 0 || (cols % 8) || (ld % 8)) return hipErrorInvalidValue;
  int rpc = 256;
  while ((rows + rpc - 1) / rpc > CS_MAX_CHUNKS) rpc *= 2;
  const int nchunks = (rows + rpc - 1) / rpc;
  hipLaunchKernelGGL(sf_colsum_bf16_kernel, dim3((cols + 255) / 256, nchunks), dim3(256), 0, s, x, rows, cols, ld, rpc, partial);
  hipLaunchKernelGGL(sf_colsum_finish_kernel, dim3((cols + 63) / 64), dim3(256), 0, s, partial, nchunks, cols, alpha, out, accumulate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fp32 row reductions (position / time embedding gradients)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_sum_rows_kernel(const float* __restrict__ in, float* out, int n_a, long stride_a,
                                                          long stride_b, int R, long stride_r, int D, int accumulate) {
  const int o = blockIdx.x;
  const long base = (long)(o % n_a) * stride_a + (long)(o / n_a) * stride_b;
  for (int c = threadIdx.x; c < (D >> 2); c += 256) {
    f32x4_t t = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) t += *reinterpret_cast<const f32x4_t*>(in + (size_t)(base + r * stride_r) * D + c * 4);
    f32x4_t* dst = reinterpret_cast<f32x4_t*>(out + (size_t)o * D + c * 4);
    if (accumulate) t += *dst;
    *dst = t;
  }
}
hipError_t sf_launch_sum_rows(const float* in, float* out, int n_out, int n_a, long stride_a, long stride_b, int R,
                              long stride_r, int D, int accumulate, hipStream_t s) {
  if (n_out <= 0) return hipSuccess;
  if (D % 4 || n_a <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_sum_rows_kernel, dim3(n_out), dim3(256), 0, s, in, out, n_a, stride_a, stride_b, R, stride_r, D, accumulate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fp32 master weights -> bf16 working copies (row-major and transposed), LoRA merge, gate scaling
// ------------------------------------------------------------------------------------------------
// [64 x 64] tile per workgroup: float4 loads, 8-byte (4 x bf16) stores for both the row-major and the transposed copy
// (2-byte stores ran this pass at 0.9 TB/s: 735 us per step for 102 M parameters)
#define PREP_T SF_PREP_TILE
SF_DEVICE void prep_tile(const float* __restrict__ w, const float* __restrict__ la, const float* __restrict__ lb, int rank,
                         const float* gate, bf16_t* w_bf, bf16_t* wT_bf, const float* bias, float* bias_out, int N, int K,
                         int kt, int nt, float (*tile)[PREP_T + 1]) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 x 16 threads, 4 consecutive k (or n) each
  const int k0 = kt * PREP_T, n0 = nt * PREP_T;
  const float scale = gate ? tanhf(*gate) : 1.0f;
  const bool kvec = (K % 4) == 0;
  __shared__ float las[32][PREP_T + 1], lbs[PREP_T][33];       // LoRA factor tiles: A[r0 .. r0+32, k0 .. k0+64], B[n0 .. n0+64, r0 .. r0+32]
  auto stage_lora = [&](int r0) {
    for (int e = threadIdx.x; e < 32 * PREP_T; e += 256) {
      const int r = e / PREP_T, kk = e % PREP_T;
      las[r][kk] = (r0 + r < rank && k0 + kk < K) ? la[(size_t)(r0 + r) * K + k0 + kk] : 0.f;
      const int nn = e / 32, rr = e % 32;
      lbs[nn][rr] = (r0 + rr < rank && n0 + nn < N) ? lb[(size_t)(n0 + nn) * rank + r0 + rr] : 0.f;
    }
  };
  if (la) { stage_lora(0); __syncthreads(); }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 16 * i, k = k0 + tx * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N && k < K) {
      if (kvec) {
        const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(w + (size_t)n * K + k);
        v[0] = wv[0]; v[1] = wv[1]; v[2] = wv[2]; v[3] = wv[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = w[(size_t)n * K + k + j];
      }
      if (la) {       // LoRA merge from the factor tiles staged below (the naive per-element loads were most of this kernel)
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        const int rn = min(32, rank);
        for (int r = 0; r < rn; ++r) {
          const float b = lbs[ty + 16 * i][r];
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] += b * las[r][tx * 4 + j];
        }
        for (int r = 32; r < rank; ++r) {          // ranks beyond the staged 32 (not used by the recipes here): straight from memory
          const float b = lb[(size_t)n * rank + r];
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k + j < K) t[j] += b * la[(size_t)r * K + k + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += t[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= scale;
      if (w_bf) {
        if (kvec) *reinterpret_cast<u32x2_t*>(w_bf + (size_t)n * K + k) = (u32x2_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k + j < K) w_bf[(size_t)n * K + k + j] = (bf16_t)f2bf(v[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[ty + 16 * i][tx * 4 + j] = v[j];
  }
  __syncthreads();
  if (wT_bf) {
    const bool nvec = (N % 4) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + ty + 16 * i, n = n0 + tx * 4;
      if (k < K && n < N) {
        const float a0 = tile[tx * 4 + 0][ty + 16 * i], a1 = tile[tx * 4 + 1][ty + 16 * i];
        const float a2 = tile[tx * 4 + 2][ty + 16 * i], a3 = tile[tx * 4 + 3][ty + 16 * i];
        if (nvec) *reinterpret_cast<u32x2_t*>(wT_bf + (size_t)k * N + n) = (u32x2_t){pack_bf2(a0, a1), pack_bf2(a2, a3)};
        else {
          const float a[4] = {a0, a1, a2, a3};
#pragma unroll
          for (int j = 0; j < 4; ++j) if (n + j < N) wT_bf[(size_t)k * N + n + j] = (bf16_t)f2bf(a[j]);
        }
      }
    }
  }
  if (bias_out && kt == 0 && threadIdx.x < PREP_T) {
    const int n = n0 + threadIdx.x;
    if (n < N) bias_out[n] = bias ? scale * bias[n] : 0.f;
  }
}

// every weight of the model in ONE launch: workgroup -> (job, tile) through the jobs' tile prefix sums
__global__ __launch_bounds__(256) void sf_prep_weights_batched_kernel(const float* __restrict__ base,
                                                                      const SfPrepJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[PREP_T][PREP_T + 1];
  int lo = 0, hi = njobs - 1;                   // last job with tile0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SfPrepJob j = jobs[lo];
  const int t = blockIdx.x - j.tile0;
  const int tiles_k = (j.K + PREP_T - 1) / PREP_T;
  prep_tile(base + j.w_off, j.la_off >= 0 ? base + j.la_off : nullptr, j.lb_off >= 0 ? base + j.lb_off : nullptr, j.rank,
            j.gate_off >= 0 ? base + j.gate_off : nullptr, j.w_bf, j.wT_bf, j.bias_off >= 0 ? base + j.bias_off : nullptr,
            j.bias_out, j.N, j.K, t % tiles_k, t / tiles_k, tile);
}
hipError_t sf_launch_prep_weights_batched(const float* base, const SfPrepJob* jobs_dev, int njobs, int total_tiles,
                                          hipStream_t s) {
  if (njobs <= 0 || total_tiles <= 0) return hipSuccess;
  hipLaunchKernelGGL(sf_prep_weights_batched_kernel, dim3(total_tiles), dim3(256), 0, s, base, jobs_dev, njobs);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// pooling-head query (probe path) forward / backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_head_query_kernel(const float* __restrict__ probe, const float* __restrict__ wq,
                                                            const float* __restrict__ bq, float scale, float* q, int D) {
  const int d = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (d >= D) return;
  float t = 0.f;
  for (int k = lane; k < D; k += 64) t += probe[k] * wq[(size_t)d * D + k];
  t = wave_sum(t);
  if (lane == 0) q[d] = (t + bq[d]) * scale;
}
hipError_t sf_launch_head_query(const float* probe, const float* wq, const float* bq, float scale, float* q, int D,
                                hipStream_t s) {
  hipLaunchKernelGGL(sf_head_query_kernel, dim3((D + 3) / 4), dim3(256), 0, s, probe, wq, bq, scale, q, D);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void sf_head_query_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ probe,
                                                                const float* __restrict__ wq, float scale, float* d_wq,
                                                                float* d_bq, float* d_probe, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;       // over D*D
  if (i < D * D) {
    const int d = i / D, k = i % D;
    d_wq[i] += scale * dq[d] * probe[k];
    if (k == 0) d_bq[d] += scale * dq[d];
  }
  if (i < D) {                                        // thread k: dprobe[k] = scale * sum_d Wq[d,k] dq[d]
    float t = 0.f;
#pragma unroll 16
    for (int d = 0; d < D; ++d) t += wq[(size_t)d * D + i] * dq[d];      // unrolled: 768 dependent round trips cost 300 us
    d_probe[i] += scale * t;
  }
}
hipError_t sf_launch_head_query_bwd(const float* dq, const float* probe, const float* wq, float scale, float* d_wq,
                                    float* d_bq, float* d_probe, int D, hipStream_t s) {
  hipLaunchKernelGGL(sf_head_query_bwd_kernel, dim3((D * D + 255) / 256), dim3(256), 0, s, dq, probe, wq, scale, d_wq, d_bq, d_probe, D);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// temporal gate: GATE_BLOCKS workgroups update dW and produce partial dots, a finish kernel combines
// them in a fixed order
// ------------------------------------------------------------------------------------------------
#define GATE_BLOCKS 128
__global__ __launch_bounds__(256) void sf_gate_grad_kernel(const float* __restrict__ G, const float* __restrict__ cs,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* gate, float* d_w, float* d_b, float* partial, int N,
                                                           int K, const float* __restrict__ r1) {
  __shared__ float red[4];
  const float t = tanhf(*gate);
  float dot = 0.f;
  const size_t nv = ((size_t)N * K) >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)GATE_BLOCKS * 256) {
    f32x4_t g = reinterpret_cast<const f32x4_t*>(G)[i];
    if (r1) {          // G_eff = G + cs (x) r1: the bias term of the fused temporal projections (g^T t_out = G1 W_o^T + cs b_o^T)
      const size_t e = i << 2;
      const int n = (int)(e / (size_t)K), k = (int)(e - (size_t)n * K);
      g += cs[n] * *reinterpret_cast<const f32x4_t*>(r1 + k);
    }
    const f32x4_t ww = reinterpret_cast<const f32x4_t*>(w)[i];
    dot += (g[0] * ww[0] + g[1] * ww[1]) + (g[2] * ww[2] + g[3] * ww[3]);
    if (d_w) {
      f32x4_t d = reinterpret_cast<f32x4_t*>(d_w)[i];
      d += g * t;
      reinterpret_cast<f32x4_t*>(d_w)[i] = d;
    }
  }
  if (blockIdx.x == 0) {
    for (int n = threadIdx.x; n < N; n += 256) {
      const float c = cs[n];
      if (b) dot += c * b[n];
      if (d_b) d_b[n] += t * c;
    }
  }
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void sf_gate_grad_finish_kernel(const float* __restrict__ partial, const float* gate,
                                                                 float* d_gate) {
  float a = 0.f;
  for (int i = threadIdx.x; i < GATE_BLOCKS; i += 64) a += partial[i];
  a = wave_sum(a);
  if (threadIdx.x == 0 && d_gate) {
    const float t = tanhf(*gate);
    *d_gate += (1.0f - t * t) * a;
  }
}
hipError_t sf_launch_gate_grad(const float* G, const float* cs, const float* w, const float* b, const float* gate,
                               float* d_w, float* d_b, float* d_gate, float* partial, int N, int K, hipStream_t s, const float* r1) {
  if (((size_t)N * K) % 4 || (K % 4)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_gate_grad_kernel, dim3(GATE_BLOCKS), dim3(256), 0, s, G, cs, w, b, gate, d_w, d_b, partial, N, K, r1);
  hipLaunchKernelGGL(sf_gate_grad_finish_kernel, dim3(1), dim3(64), 0, s, partial, gate, d_gate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// temporal_dense o temporal_attention.output.dense as ONE projection in the training step (modeling:947-958: two Linear layers with
// nothing in between at drop rates 0):  W_f = tanh(g) W_d W_o,  b_f = tanh(g) (W_d b_o + b_d).
// sf_fuse_temporal_kernel: W_f[i][j] = sum_k wd[i][k] woT[j][k] from the bf16 working copies (wd carries tanh(g) already; woT is the
// transposed copy of W_o), one 16 x 16 tile per wave, written as wf [D, D] (forward operand) and wfT [D, D] (input-gradient operand).
// grid (D/16 * D/16 / 4, layers)
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf_tk;
SF_DEVICE f32x4_t tk_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_tk, a), __builtin_bit_cast(v8bf_tk, b), c, 0, 0, 0);
}
// a wave computes a 16 (i) x 64 (j) strip: the wd fragment is shared by four MFMAs, ten loads per trip are in flight
__global__ __launch_bounds__(256) void sf_fuse_temporal_kernel(const SfFuseJob* __restrict__ jobs, int D) {
  const SfFuseJob J = jobs[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int sj = D >> 6, si = D >> 4;           // strips along j, along i
  const int strip = blockIdx.x * 4 + wave;
  if (strip >= si * sj) return;
  const int i0 = (strip / sj) * 16, j0 = (strip % sj) * 64;
  const bf16_t* br = J.wd + (size_t)(i0 + l15) * D + g * 8;       // B operand rows: i
  const bf16_t* ar = J.woT + (size_t)(j0 + l15) * D + g * 8;      // A operand rows: j (four 16-row tiles, 16 D apart)
  f32x4_t acc[4], acct[4];             // acct: the same products with the operand roles swapped = the transposed tile's lane layout
#pragma unroll
  for (int q = 0; q < 4; ++q) { acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acct[q] = acc[q]; }
  for (int k = 0; k < D; k += 64) {                                // D % 64 == 0: two k-steps per trip, loads first
    const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(br + k), b1 = *reinterpret_cast<const bf16x8_t*>(br + k + 32);
    bf16x8_t a0[4], a1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a0[q] = *reinterpret_cast<const bf16x8_t*>(ar + (size_t)q * 16 * D + k);
      a1[q] = *reinterpret_cast<const bf16x8_t*>(ar + (size_t)q * 16 * D + k + 32);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q] = tk_mfma(a0[q], b0, acc[q]);
      acc[q] = tk_mfma(a1[q], b1, acc[q]);
      acct[q] = tk_mfma(b0, a0[q], acct[q]);
      acct[q] = tk_mfma(b1, a1[q], acct[q]);
    }
  }
  // lane: W_f[i0 + l15][j0 + 16 q + 4g + jj], jj = 0..3
  const int i = i0 + l15;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = j0 + 16 * q + 4 * g;
    *reinterpret_cast<u32x2_t*>(J.wf + (size_t)i * D + j) = (u32x2_t){pack_bf2(acc[q][0], acc[q][1]), pack_bf2(acc[q][2], acc[q][3])};
    // acct lane: W_f[i0 + 4g + jj][j0 + 16 q + l15] -> wfT[j][i0 + 4g .. + 3]: 8-byte stores instead of 2-byte scatters
    *reinterpret_cast<u32x2_t*>(J.wfT + (size_t)(j0 + 16 * q + l15) * D + i0 + 4 * g) = (u32x2_t){pack_bf2(acct[q][0], acct[q][1]), pack_bf2(acct[q][2], acct[q][3])};
  }
}
// b_f[i] = tanh(g) (sum_k W_d[i][k] b_o[k] + b_d[i]) from the fp32 parameters; grid (D / 4, layers), one wave per row
__global__ __launch_bounds__(256) void sf_fuse_temporal_bias_kernel(const float* __restrict__ base, const SfFuseJob* __restrict__ jobs, int D) {
  const SfFuseJob J = jobs[blockIdx.y];
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= D) return;
  const float* wd = base + J.wd_off + (size_t)i * D;
  const float* bo = base + J.bo_off;
  float t = 0.f;
  for (int k = lane; k < D; k += 64) t = fmaf(wd[k], bo[k], t);
  t = wave_sum(t);
  if (lane == 0) J.bf[i] = tanhf(base[J.gate_off]) * (t + base[J.bd_off + i]);
}
hipError_t sf_launch_fuse_temporal(const float* base, const SfFuseJob* jobs_dev, int layers, int D, hipStream_t s) {
  if (layers <= 0) return hipSuccess;
  if (D % 64) return hipErrorInvalidValue;
  const int strips = (D / 16) * (D / 64);
  hipLaunchKernelGGL(sf_fuse_temporal_kernel, dim3((strips + 3) / 4, layers), dim3(256), 0, s, jobs_dev, D);
  hipLaunchKernelGGL(sf_fuse_temporal_bias_kernel, dim3((D + 3) / 4, layers), dim3(256), 0, s, base, jobs_dev, D);
  return hipGetLastError();
}
// out[k] += sum_i w[i * ld + k] * v[i]   (w bf16 [rows, ld]): db_o = (tanh(g) W_d)^T colsum(g) of the fused temporal projections
__global__ __launch_bounds__(1024) void sf_matvec_t_bf16_kernel(const bf16_t* __restrict__ w, int ld, const float* __restrict__ v, float* __restrict__ out,
                                                                int rows, int cols) {
  __shared__ float part[16][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;        // 64 columns x 16 row groups, partial sums met in a fixed order
  const int k = blockIdx.x * 64 + c;
  float t = 0.f;
  if (k < cols)
#pragma unroll 4
    for (int i = rg; i < rows; i += 16) t = fmaf(bf2f(w[(size_t)i * ld + k]), v[i], t);
  part[rg][c] = t;
  __syncthreads();
  if (rg == 0 && k < cols) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += part[r][c];
    out[k] += a;
  }
}
hipError_t sf_launch_matvec_t_bf16(const bf16_t* w, int ld, const float* v, float* out, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(sf_matvec_t_bf16_kernel, dim3((cols + 63) / 64), dim3(1024), 0, s, w, ld, v, out, rows, cols);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// AdamW over the flat parameter buffer (torch.optim.AdamW update order)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_adamw_kernel(SfAdamWArgs a) {
  const size_t nv = a.n >> 2;
  float gscale = a.grad_scale;
  if (a.clip_sumsq) gscale *= fminf(1.0f, a.clip_norm / (sqrtf(a.clip_sumsq[0]) * a.grad_scale + 1e-6f));
  if (a.guard_flag) {        // every thread reads the same two scalars: a uniform, deterministic decision
    bool bad = !(a.guard_sumsq[0] < __builtin_huge_valf());                       // inf or NaN
    if (a.guard_loss) bad = bad || !(fabsf(a.guard_loss[0]) < __builtin_huge_valf());
    if (bad) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { a.guard_flag[0] = 1; a.guard_flag[1] += 1; }
      if (a.zero_grads)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256)
          reinterpret_cast<f32x4_t*>(a.g)[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      return;
    }
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
    const size_t e = i << 2;                     // segments start on multiples of 64 elements
    int lo = 0, hi = a.nseg - 1;                 // first segment with seg_end > e
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((size_t)a.seg_end[mid] > e) hi = mid; else lo = mid + 1;
    }
    if (!a.seg_train[lo]) continue;
    const float wd = a.seg_decay[lo] ? a.weight_decay : 0.f;
    float bc1 = a.bias_correction1, bc2 = a.bias_correction2;
    if (a.n_extra && lo >= a.extra_seg0) {       // a head scalar: its own step count, or skipped (16 threads per slot)
      const int idx = lo - a.extra_seg0;
      int st = 0;
#pragma unroll
      for (int k = 0; k < 64; ++k) st = (k == idx) ? a.extra_steps[k] : st;      // static kernarg offsets
      if (st <= 0) {
        if (a.zero_grads) reinterpret_cast<f32x4_t*>(a.g)[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        continue;
      }
      bc1 = 1.0f - powf(a.beta1, (float)st);
      bc2 = 1.0f - powf(a.beta2, (float)st);
    }
    f32x4_t p = reinterpret_cast<f32x4_t*>(a.p)[i];
    f32x4_t g = reinterpret_cast<const f32x4_t*>(a.g)[i];
    if (a.zero_grads) reinterpret_cast<f32x4_t*>(a.g)[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t m = reinterpret_cast<f32x4_t*>(a.m)[i];
    f32x4_t v = reinterpret_cast<f32x4_t*>(a.v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = g[j] * gscale;
      p[j] *= 1.0f - a.lr * wd;
      m[j] = a.beta1 * m[j] + (1.0f - a.beta1) * gj;
      v[j] = a.beta2 * v[j] + (1.0f - a.beta2) * gj * gj;
      const float denom = sqrtf(v[j]) / sqrtf(bc2) + a.eps;
      p[j] -= (a.lr / bc1) * (m[j] / denom);
    }
    reinterpret_cast<f32x4_t*>(a.p)[i] = p;
    reinterpret_cast<f32x4_t*>(a.m)[i] = m;
    reinterpret_cast<f32x4_t*>(a.v)[i] = v;
  }
}
hipError_t sf_launch_adamw(const SfAdamWArgs& a, hipStream_t s) {
  if (a.n % 4 || a.nseg <= 0) return hipErrorInvalidValue;
  if (!a.n) return hipSuccess;
  hipLaunchKernelGGL(sf_adamw_kernel, dim3(ew_grid(a.n / 4)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// 16-byte loads, four in flight per lane, four independent accumulators (round 4: the 4-byte, one-at-a-time loop of the first version ran
// at 2.3 TB/s: 174 us for the 407 MB of gradients); fixed combination order, so the sum is reproducible for a given n
__global__ __launch_bounds__(256) void sf_sumsq_kernel(const float* __restrict__ g, size_t n, float* partial) {
  __shared__ float red[4];
  const size_t n4 = ((reinterpret_cast<size_t>(g) & 15) == 0) ? n / 4 : 0;      // vector body only on a 16-byte aligned base
  const f32x4_t* g4 = reinterpret_cast<const f32x4_t*>(g);
  const size_t stride = (size_t)gridDim.x * 256;
  f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4_t v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
    a0 += v0 * v0; a1 += v1 * v1; a2 += v2 * v2; a3 += v3 * v3;
  }
  for (; i < n4; i += stride) { const f32x4_t v = g4[i]; a0 += v * v; }
  const f32x4_t a = (a0 + a1) + (a2 + a3);
  float t = (a[0] + a[1]) + (a[2] + a[3]);
  for (size_t j = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) t += g[j] * g[j];      // tail (or everything, unaligned)
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void sf_sumsq_finish_kernel(const float* __restrict__ partial, int n, float* out) {
  float t = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) t += partial[i];
  t = wave_sum(t);
  if (threadIdx.x == 0) out[0] = t;
}
hipError_t sf_launch_sumsq(const float* g, size_t n, float* out, float* partial, hipStream_t s) {
  const int blocks = 1024;
  hipLaunchKernelGGL(sf_sumsq_kernel, dim3(blocks), dim3(256), 0, s, g, n, partial);
  hipLaunchKernelGGL(sf_sumsq_finish_kernel, dim3(1), dim3(64), 0, s, partial, blocks, out);
  return hipGetLastError();
}
