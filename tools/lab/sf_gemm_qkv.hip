// LAB LIBRARY ONLY (build.py --lab, SF_QKV_FUSED=1 / SF_SQKV_PANEL=1): built to bit-identical parity in round 4, 16 us of kernel time per layer
// cheaper under rocprof, 0.07 ms per forward SLOWER on the wall clock — the forward is bound by the power envelope, so the MFMA kernels
// that lose the low-power attention launch between them give the time back (profiles/r04_qkv_fused_ab.txt).
//
// LayerNorm-folded qkv projection on the panel tile, with the temporal attention of a full 16-frame clip as its epilogue
// (modeling_timesformer_siglip.py:937-958 — temporal LayerNorm, `temporal_attention.attention` qkv Linear, causal attention over
// the frames of one patch; round 4).
//
//   qkv = rstd * (bf16(x) W'^T - mean * s) + b'          W' = W * gamma, b' = b + W beta, s_n = sum_k bf16(W')[n,k]
//   ctx[(b, t, n), h] = softmax_t'(scale * q_t k_t'^T + causal) v_t'            per (clip b, patch n, head h), t, t' < 16
//
// Why here: a temporal sequence is 16 rows x 192 columns of the qkv tensor.  With the GEMM's rows taken in (clip, patch, frame)
// order — a row gather in the LDS-DMA source addresses, free — an MFMA row tile IS one sequence, and with W' permuted so that a
// 384-column tile holds [q | k | v] of two heads, a 208 x 384 tile holds 13 sequences x 2 heads completely.  The qkv tensor
// (115 MB written, 115 MB read back) and the attention launch (26 us) disappear; the tile's 26 attention problems (16 x 16 scores,
// seven MFMAs each) run from LDS on the waves that produced them.
//
// gfx950 structure = sf_gemm_panel.hip: 8 waves = 1(M) x 8(N), a wave owns 13 m-tiles x 3 n-tiles (39 MFMA 16x16x32 per K-tile), one
// A piece (256 rows x 64 B) + one W piece (384 x 64 B) per K-tile in a 4-slot LDS ring (160 KB), the two halves of the workgroup one
// barrier apart, refills issued in the read segment, counted vmcnt.  768 tiles of 196 rows (plain) / 726 of 208 rows (fused) on 256
// persistent workgroups = three rounds; the six column tiles of a row panel run on one XCD (its A rows are fetched once into that L2).
// Epilogue: LayerNorm fold in registers -> bf16 -> LDS as thirteen [16 x 384] images of six swizzled [16 x 128 B] head blocks (the ring
// is dead by then: 156 KB) -> either whole-row copy-out (plain: the spatial qkv projection) or the attention problems (fused).
#include "sf_common.h"
#include <cstdlib>

#define Q_THREADS 512
#define Q_MT 13
#define Q_NT 3
#define Q_SLOT_BYTES 40960
#define Q_A_BYTES 16384
#define Q_IMG_BYTES 2048              // one head block: 16 rows x 128 B
#define Q_MTILE_BYTES (6 * Q_IMG_BYTES)

typedef __attribute__((address_space(3))) void* q_lptr_t;
typedef __attribute__((ext_vector_type(4))) short q_s16x4_t;
typedef __attribute__((address_space(3))) q_s16x4_t* q_lds_s16x4_ptr;

SF_DEVICE f32x4_t q_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
template <int N>
SF_DEVICE void q_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
SF_DEVICE bf16x8_t q_rd32(const char* piece, int row, int kc) {
  return *reinterpret_cast<const bf16x8_t*>(piece + row * 64 + ((kc ^ sf_swz64(row)) << 4));
}
// head-block swizzle (same as sf_attention.hip): row fragments of 16 rows and the 8 rows of a half-wave transposed read are conflict-free
SF_DEVICE int q_bswz(int row) { return (((row >> 1) & 3) << 1) | ((row >> 3) & 1); }
SF_DEVICE int q_img_off(int row, int chunk) { return row * 128 + ((chunk ^ q_bswz(row)) << 4); }
SF_DEVICE bf16x8_t q_row_frag(const char* img, int row, int chunk) {
  return *reinterpret_cast<const bf16x8_t*>(img + q_img_off(row, chunk));
}
// token-major fragment of a 16-row block: lane (l15 = column e of head-dim tile et, g) gets rows 4g .. 4g+3; rows 16 .. 31 are zeros
SF_DEVICE bf16x8_t q_tr_frag16(const char* img, int et, int lane) {
  const int t16 = lane & 15, g = lane >> 4;
  const int row = 4 * g + (t16 >> 2);
  const int off = q_img_off(row, 2 * et + ((t16 & 3) >> 1)) + ((t16 & 1) << 3);
  const q_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q_lds_s16x4_ptr)(img + off));
  bf16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = f[5] = f[6] = f[7] = 0;
  return f;
}

template <bool FUSED>
__global__ __launch_bounds__(Q_THREADS) void sf_gemm_qkv_kernel(SfQkvArgs p, int rows_per_tile, int npanels) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  const int K = p.K;
  const int nkt = K >> 5;
  const int ncol = (3 * p.D) / 384;                   // column tiles per row panel (6 at D = 768)
  const int nseq = p.B * p.NP;                        // fused: temporal sequences (clip, patch)
  // persistent walk: XCD x (= blockIdx & 7 under round-robin dispatch) owns row panels x, x + 8, ...; its workgroups take that list's
  // (panel, column tile) pairs in order, so the column tiles of a panel run together on one L2
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nloc = gridDim.x >> 3;
  const int npx = (npanels - xcd + 7) >> 3;
  for (int it = wl; it < npx * ncol; it += nloc) {
  const int panel = (it / ncol) * 8 + xcd, ct = it % ncol;
  const int m0 = panel * rows_per_tile;               // plain: first row; fused: 13 * panel = first sequence, times 16
  const int rows_here = FUSED ? min(13, nseq - panel * 13) * 16 : min(rows_per_tile, p.M - m0);
  const int n0 = ct * 384;
  if (rows_here <= 0) continue;

  // source row of tile row r: plain m0 + r; fused: sequence q = 13 panel + r / 16 = (clip, patch), frame r % 16 -> ((clip T + frame) NP + patch)
  auto src_row = [&](int r) -> int {
    r = r < rows_here ? r : rows_here - 1;
    if (!FUSED) return m0 + r;
    const int q = panel * 13 + (r >> 4), t = r & 15;
    const int b = q / p.NP, n = q - b * p.NP;
    return (b * p.T + t) * p.NP + n;
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (unsigned)p.M * (unsigned)K * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)(3 * p.D) * (unsigned)K * 2u, 0x00020000);
  unsigned offA[2], offW[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * Q_THREADS + tid;
    const int row = c >> 2, kc = (c & 3) ^ sf_swz64(row);
    if (i < 2) offA[i] = ((unsigned)src_row(row) * (unsigned)K + kc * 8) * 2u;
    offW[i] = ((unsigned)(n0 + row) * (unsigned)K + kc * 8) * 2u;
  }
  const int dma_lds = wave * 1024;
  // A rows 208 .. 255 of a piece are never read: waves 5 .. 7 skip the second A instruction (their vmcnt counts are one lower per
  // K-tile), which leaves 3 KB at the tail of every slot's A piece free for the epilogue's operand tables below
  const bool short_a = wave >= 5;
  auto issue = [&](int t) {
    char* dst = smem + (t & 3) * Q_SLOT_BYTES + dma_lds;
    const int kof = t * 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (q_lptr_t)dst, 16, offA[0], kof, 0, 0);
    if (!short_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (q_lptr_t)(dst + 8192), 16, offA[1], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (q_lptr_t)(dst + Q_A_BYTES), 16, offW[0], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (q_lptr_t)(dst + Q_A_BYTES + 8192), 16, offW[1], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (q_lptr_t)(dst + Q_A_BYTES + 16384), 16, offW[2], kof, 0, 0);
  };

  // epilogue operands, fetched now (nothing here depends on the K loop): {mean, rstd} of the tile's 208 rows from the producer's row
  // sums -> tail of slot 0; bias' and s of the tile's 384 columns -> tail of slot 1.  Four {sum x, sum x^2} pairs per row, combined in
  // the fixed order of the 256^2 kernel's consumer (bit-identical mean / rstd).
  float* tab_mr = reinterpret_cast<float*>(smem + 208 * 64);
  float* tab_bs = reinterpret_cast<float*>(smem + Q_SLOT_BYTES + 208 * 64);
  if (tid < 208) {
    const float* sp = p.ln_stats + (size_t)src_row(tid) * 8;
    f32x4_t st = *reinterpret_cast<const f32x4_t*>(sp);
    const f32x4_t s2 = *reinterpret_cast<const f32x4_t*>(sp + 4);
    st[0] += st[2]; st[1] += st[3];
    st[2] = s2[0] + s2[2]; st[3] = s2[1] + s2[3];
    const float invd = 1.0f / (float)K;
    const float mu = (st[0] + st[2]) * invd;
    const float rs = rsqrtf((st[1] + st[3]) * invd - mu * mu + p.ln_eps);
    *reinterpret_cast<u32x2_t*>(tab_mr + 2 * tid) = (u32x2_t){__float_as_uint(mu), __float_as_uint(rs)};
  } else if (tid < 208 + 96) {
    const int i4 = (tid - 208) * 4;
    *reinterpret_cast<f32x4_t*>(tab_bs + i4) = *reinterpret_cast<const f32x4_t*>(p.bias + n0 + i4);
    *reinterpret_cast<f32x4_t*>(tab_bs + 384 + i4) = *reinterpret_cast<const f32x4_t*>(p.ln_s + n0 + i4);
  }
  f32x4_t acc[Q_MT][Q_NT];
#pragma unroll
  for (int i = 0; i < Q_MT; ++i)
#pragma unroll
    for (int j = 0; j < Q_NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[Q_MT], wf[Q_NT];
  auto reads = [&](int t) {
    const char* pa = smem + (t & 3) * Q_SLOT_BYTES;
    const char* pw = pa + Q_A_BYTES;
#pragma unroll
    for (int nt = 0; nt < Q_NT; ++nt) wf[nt] = q_rd32(pw, wave * 48 + nt * 16 + l15, g);
#pragma unroll
    for (int mt = 0; mt < Q_MT; ++mt) af[mt] = q_rd32(pa, mt * 16 + l15, g);
  };
  auto mma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < Q_MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < Q_NT; ++nt) acc[mt][nt] = q_mfma(wf[nt], af[mt], acc[mt][nt]);
    __builtin_amdgcn_s_setprio(0);
  };
  // ---- main loop: the schedule of sf_gemm_panel.hip (see the hazard argument there) ----------------------------------------------
  int t = 0;
  if (half == 1) {
    // waves 5 .. 7 issue four DMA instructions per K-tile instead of five: the same waits, counted in their own instructions
    auto wait2 = [&]() { if (short_a) q_wait_vm<8>(); else q_wait_vm<10>(); };
    auto wait1 = [&]() { if (short_a) q_wait_vm<4>(); else q_wait_vm<5>(); };
    issue(0); issue(1); issue(2);
    wait2();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    for (; t + 3 < nkt; ++t) {
      issue(t + 3);
      reads(t);
      wait2();
      __builtin_amdgcn_s_barrier();
      mma();
      __builtin_amdgcn_s_barrier();
    }
    reads(t); wait1(); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier(); ++t;
    reads(t); q_wait_vm<0>(); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier(); ++t;
    reads(t); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier();
  } else {
    issue(0); issue(1);
    q_wait_vm<5>();
    __builtin_amdgcn_s_barrier();
    for (; t + 2 < nkt; ++t) {
      issue(t + 2);
      reads(t);
      q_wait_vm<5>();
      __builtin_amdgcn_s_barrier();
      mma();
      __builtin_amdgcn_s_barrier();
    }
    reads(t); q_wait_vm<0>(); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier(); ++t;
    reads(t); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue 1: LayerNorm fold, bias, bf16, into the LDS images (every read of the ring has been consumed: last barrier above) --
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int el15 = tid_e & 15, eg = (tid_e >> 4) & 3;
  f32x4_t bias4[Q_NT], lns4[Q_NT];
#pragma unroll
  for (int nt = 0; nt < Q_NT; ++nt) {
    const int c = wave * 48 + nt * 16 + eg * 4;
    bias4[nt] = *reinterpret_cast<const f32x4_t*>(tab_bs + c);
    lns4[nt] = *reinterpret_cast<const f32x4_t*>(tab_bs + 384 + c);
  }
  u32x2_t mr[Q_MT];
#pragma unroll
  for (int mt = 0; mt < Q_MT; ++mt) mr[mt] = *reinterpret_cast<const u32x2_t*>(tab_mr + 2 * (mt * 16 + el15));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                       // the tables are in registers everywhere: the images may overwrite them
#pragma unroll
  for (int mt = 0; mt < Q_MT; ++mt) {
    const float mu = __uint_as_float(mr[mt][0]), rs = __uint_as_float(mr[mt][1]);
#pragma unroll
    for (int nt = 0; nt < Q_NT; ++nt) {
      const int c = wave * 48 + nt * 16 + eg * 4;             // column inside the tile: head block c / 64, chunk (c % 64) / 8
      const f32x4_t v = rs * (acc[mt][nt] - mu * lns4[nt]) + bias4[nt];
      const u32x2_t hv = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(smem + mt * Q_MTILE_BYTES + (c >> 6) * Q_IMG_BYTES + q_img_off(el15, (c & 63) >> 3) + (c & 7) * 2) = hv;
    }
  }
  __syncthreads();

  if (!FUSED) {
    // ---- epilogue 2 (plain): whole rows out, 16 bytes per lane, lanes 0..47 = the 384 columns of the tile -------------------------
    const int elane = tid_e & 63;
    for (int r = wave; r < rows_here; r += 8) {
      if (elane < 48) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + (r >> 4) * Q_MTILE_BYTES + (elane >> 3) * Q_IMG_BYTES + q_img_off(r & 15, elane & 7));
        *reinterpret_cast<u32x4_t*>(p.out + (size_t)(m0 + r) * (size_t)(3 * p.D) + n0 + elane * 8) = v;
      }
    }
  } else {
    // ---- epilogue 2 (fused): 13 sequences x 2 heads, one wave per problem; blocks q_h, k_h, v_h = images h, 2 + h, 4 + h --------------
    const int elane = tid_e & 63;
    const float c2 = p.scale * 1.44269504088896340736f;
    for (int task = wave; task < 2 * (rows_here >> 4); task += 8) {
      const int sq = task >> 1, h = task & 1;
      char* base = smem + sq * Q_MTILE_BYTES;
      char* q_img = base + h * Q_IMG_BYTES;
      const char* k_img = base + (2 + h) * Q_IMG_BYTES;
      const char* v_img = base + (4 + h) * Q_IMG_BYTES;
      // S^T = K Q^T: lane (query l15, g) holds keys 4 g + r
      f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) s = q_mfma(q_row_frag(k_img, el15, ks * 4 + eg), q_row_frag(q_img, el15, ks * 4 + eg), s);
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 4 * eg + r;
        const bool ok = key < p.T && (!p.causal || key <= el15);
        s[r] = ok ? s[r] : -INFINITY;
        mx = fmaxf(mx, s[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mc = mx * c2;
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -mc));
        s[r] = e;
        sum += e;
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
      const u32x4_t pu = {pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3]), 0u, 0u};
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
      f32x4_t o[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = q_mfma(q_tr_frag16(v_img, dt, elane), pf, (f32x4_t){0.f, 0.f, 0.f, 0.f});
      // the q block of this problem is consumed: it becomes the wave's output patch [16 queries][128 B]
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int off = el15 * 128 + (((dt * 2 + (eg >> 1)) ^ (el15 & 7)) << 4) + (eg & 1) * 8;
        const f32x4_t ov = o[dt] * inv;
        *reinterpret_cast<u32x2_t*>(q_img + off) = (u32x2_t){pack_bf2(ov[0], ov[1]), pack_bf2(ov[2], ov[3])};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int qg = panel * 13 + sq;                     // (clip, patch)
      const int b = qg / p.NP, n = qg - b * p.NP;
      const int head = ct * 2 + h;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const int idx = i2 * 64 + elane;
        const int r = idx >> 3, c = idx & 7;
        if (r < p.T) {
          const size_t o_off = ((size_t)(b * p.T + r) * p.NP + n) * (size_t)p.D + head * 64 + c * 8;
          *reinterpret_cast<u32x4_t*>(p.out + o_off) = *reinterpret_cast<const u32x4_t*>(q_img + r * 128 + ((c ^ (r & 7)) << 4));
        }
      }
    }
  }
  __syncthreads();                                      // the images are the next tile's ring
  }   // tiles
}

bool sf_gemm_qkv_supported(const SfQkvArgs& a, bool fused) {
  if (!a.a || !a.w || !a.bias || !a.ln_s || !a.ln_stats || !a.out) return false;
  if (a.D <= 0 || (3 * a.D) % 384 || a.K % 32 || a.K < 128) return false;
  if ((size_t)a.M * a.K * 2 >= ((size_t)1 << 32) || (size_t)3 * a.D * a.K * 2 >= ((size_t)1 << 32)) return false;
  if (fused) {
    if (a.T != 16 || a.B <= 0 || a.NP <= 0 || a.M != a.B * a.T * a.NP) return false;
    if (a.B * a.NP < 13 * 8 * 4) return false;           // fewer than four row panels per XCD: the large-tile schedule does not pay
  } else if (a.M < 196 * 8 * 4) return false;
  return true;
}

hipError_t sf_launch_gemm_qkv(const SfQkvArgs& a, bool fused, hipStream_t s) {
  if (!sf_gemm_qkv_supported(a, fused)) return hipErrorInvalidValue;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  static int cached_cus = 0;
  if (!cached_cus) {
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached_cus = prop.multiProcessorCount & ~7;
    if (cached_cus < 8) cached_cus = 256;
  }
  cus = cached_cus;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_qkv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * Q_SLOT_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_qkv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * Q_SLOT_BYTES);
  }
  const size_t lds = 4 * Q_SLOT_BYTES;
  if (fused) {
    const int npanels = (a.B * a.NP + 12) / 13;
    hipLaunchKernelGGL(sf_gemm_qkv_kernel<true>, dim3(cus), dim3(Q_THREADS), lds, s, a, 208, npanels);
  } else {
    // rows per tile: the panel plan of sf_gemm_panel.hip — P a multiple of CUs / 2 ... here simply ceil(M / panels) <= 208 with panels a multiple of 8
    int panels = ((a.M + 207) / 208 + 7) & ~7;
    int rows = (a.M + panels - 1) / panels;
    // prefer exactly 196-row panels when they divide M (one frame per panel at 224^2)
    if (a.M % 196 == 0 && ((a.M / 196) & 7) == 0) { panels = a.M / 196; rows = 196; }
    hipLaunchKernelGGL(sf_gemm_qkv_kernel<false>, dim3(cus), dim3(Q_THREADS), lds, s, a, rows, panels);
  }
  return hipGetLastError();
}
