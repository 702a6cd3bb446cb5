// "Pipe" panel GEMM for the N = 768 residual projections at BASELINE-sized M (round 4):
//     out = resid + alpha * (A[M,K] * W[N,K]^T + bias)        residual stream as hi + lo bf16 planes
//
// sf_gemm_panel.hip gives every CU ONE 196 x 384 tile: its compute waves also issue the LDS-DMA refills (100-185 issue
// cycles each in that phase: "a phase is two read segments long") and the residual read-modify-write of the
// epilogue (154 MB per launch) runs after the K loop with no MFMA anywhere on the chip.  This kernel splits the ROLES over the
// eight waves of one persistent workgroup per CU instead, and lets them run decoupled:
//   waves 0-3   MFMA: one per SIMD, each owns all 7 m-tiles x 3 n-tiles of a 98 x 192 tile (21 MFMA 16x16x32 per 32-deep K-step);
//               two fragment sets: the ds_reads of step s + 1 return under the MFMAs of step s; at the end of a tile they dump
//               acc + bias into an fp32 staging image
//   wave 4      A loader: LDS-DMA of the tile's A piece (98 rows x 64 B) of step s + 7 into an 8-slot ring (HBM latency)
//   wave 5      W loader: LDS-DMA of the W piece (192 x 64 B) of step s + 2 into a 3-slot ring (L2 latency)
//   waves 6-7   storers: epilogue of tile k WHILE the MFMA waves run tile k + 1: residual planes in (requested a whole tile
//               ahead), staged C from LDS, re-split, both planes out, LayerNorm row sums of the next Linear
// (two loader waves are enough: the 19 LDS-DMA pieces of a step are paced by the CU's texture-address path, ~17 cycles per
// 1 KB piece whoever issues them — profiles/r04_pipe_trace.txt)
// A workgroup walks row panels of <= 98 rows; a panel is four consecutive tiles (column quarters), so its A rows come from HBM
// once and from L2 three times, and the step sequence runs across tiles and panels without draining.
// Synchronisation: no s_barrier after the prologue.  Every hand-off is a single-writer step counter in LDS (one dword per
// writing wave; readers take the minimum over the writers with one ds_read_b128): ready_a / ready_w (loaders -> MFMA waves,
// written behind the loader's counted vmcnt wait), consumed (MFMA waves -> loaders, written behind the lgkmcnt wait of the
// fragment reads), staged (MFMA waves -> storers) and done (storers -> MFMA waves: the staging image is free again).  Counter
// accesses are inline asm (the compiler neither reorders them nor drains vmcnt for them) and every spin is bounded.
// Slot reuse by construction: a loader overwrites the slot of step s - 1 only after every MFMA wave published consumed >= s,
// which each does after its reads of step s - 1 have returned.
#include "sf_common.h"
#include <cstdlib>

#define PI_THREADS 512
#define PI_MT 7
#define PI_NT 3
#define PI_BN 192
#define PI_ROWS 98                       // rows of a panel (staging image and A slots are sized for it)
#define PI_W_SLOT 12288                  // 192 rows x 64 B
#define PI_A_SLOT 6272                   // 98 rows x 64 B
#define PI_RW 3
#define PI_RA 8
#define PI_W_OFF 0
#define PI_A_OFF (PI_RW * PI_W_SLOT)                     // 36864
#define PI_ST_OFF (PI_A_OFF + PI_RA * PI_A_SLOT)         // 87040
#define PI_FLAG_OFF (PI_ST_OFF + PI_ROWS * 768)          // 162304
#define PI_LDS_BYTES (PI_FLAG_OFF + 128)                 // 162432 <= 163840
#define PI_SPIN_LIMIT (1 << 18)      // ~25 ms of polling; after one failed wait a wave stops waiting altogether (garbage out, no hang)

typedef __attribute__((address_space(3))) void* lptr_t;

namespace {
SF_DEVICE f32x4_t mfma16i(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
template <int N>
SF_DEVICE void wait_vmi() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
SF_DEVICE bf16x8_t rd32i(const char* piece, int row, int kc) {
  return *reinterpret_cast<const bf16x8_t*>(piece + row * 64 + ((kc ^ sf_swz64(row)) << 4));
}
SF_DEVICE float half_sum_dpp_i(float v) {      // sums over lanes 0..31 / 32..63, totals in lanes 31 and 63; all lanes active
  v = dpp_add<0x111, 0xf>(v);
  v = dpp_add<0x112, 0xf>(v);
  v = dpp_add<0x114, 0xf>(v);
  v = dpp_add<0x118, 0xf>(v);
  v = dpp_add<0x142, 0xa>(v);
  return v;
}
// staging image: [98 rows][48 chunks of 16 B], chunk XOR (row & 7) inside its 8-chunk group
SF_DEVICE int stage_off_i(int r, int chunk) { return r * 768 + (((chunk & ~7) | ((chunk ^ r) & 7)) << 4); }

// ---- LDS step counters (byte offsets from the start of the workgroup's LDS) ---------------------------------------------
SF_DEVICE void flag_store(unsigned off, int v) {      // every earlier LDS access of this wave has completed (callers wait first)
  asm volatile("ds_write_b32 %0, %1" ::"v"(off), "v"(v) : "memory");
}
SF_DEVICE int flag_min4(unsigned off) {               // minimum of four adjacent counters
  u32x4_t f;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(off) : "memory");
  const int a = min((int)f[0], (int)f[1]), b = min((int)f[2], (int)f[3]);
  return min(a, b);
}
SF_DEVICE int flag_min2(unsigned off) {
  u32x2_t f;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(off) : "memory");
  return min((int)f[0], (int)f[1]);
}
// bounded spin until the minimum over the `n` (2 or 4) counters at `off` reaches `want`; false = gave up (results invalid, no hang)
template <int N>
SF_DEVICE bool wait_flag(unsigned off, int want) {
  for (int it = 0; it < PI_SPIN_LIMIT; ++it) {
    const int v = N == 4 ? flag_min4(off) : flag_min2(off);
    if (__builtin_amdgcn_readfirstlane(v) >= want) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
}  // namespace

// SF_PIPE_TRACE (tools/pipe_trace_lab.hip only): shader-clock stamps per role, summed over the workgroups into pipe_trace[]
#ifdef SF_PIPE_TRACE
__device__ unsigned long long pipe_trace[32];
#define TR_DECL unsigned long long tr_a = 0, tr_b = 0, tr_c = 0, tr_d = 0, tr_t = 0, tr_0 = __builtin_amdgcn_s_memtime()
#define TR_MARK(x) do { const unsigned long long tr_n = __builtin_amdgcn_s_memtime(); (x) += tr_n - tr_t; tr_t = tr_n; } while (0)
#define TR_START() do { tr_t = __builtin_amdgcn_s_memtime(); } while (0)
#define TR_FLUSH(base) do { if (lane == 0) { atomicAdd(&pipe_trace[(base) + 0], tr_a); atomicAdd(&pipe_trace[(base) + 1], tr_b); atomicAdd(&pipe_trace[(base) + 2], tr_c); \
    atomicAdd(&pipe_trace[(base) + 3], tr_d); atomicAdd(&pipe_trace[(base) + 4], __builtin_amdgcn_s_memtime() - tr_0); atomicAdd(&pipe_trace[(base) + 5], 1ull); } } while (0)
#else
#define TR_DECL
#define TR_MARK(x)
#define TR_START()
#define TR_FLUSH(base)
#endif

// flags (dwords): [0..3] consumed (MFMA waves), [4..5] ready_a (both written by the A loader), [6..7] ready_w, [8..11] staged (MFMA waves), [12..13] done (storers)
#define PI_F_CONSUMED (PI_FLAG_OFF + 0)
#define PI_F_READY_A (PI_FLAG_OFF + 16)
#define PI_F_READY_W (PI_FLAG_OFF + 24)
#define PI_F_STAGED (PI_FLAG_OFF + 32)
#define PI_F_DONE (PI_FLAG_OFF + 48)

__global__ __launch_bounds__(PI_THREADS) void sf_gemm_pipe_kernel(SfGemmArgs p, int rows_per_panel, int panels, int* fail_flag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.K;
  const int nkt = K >> 5;
  // panels of this workgroup: blockIdx.x, + gridDim.x, ...; a panel = 4 tiles (column quarters), a tile = nkt steps
  const int my_panels = (panels - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ntiles = my_panels * 4;
  const int S = ntiles * nkt;
  if (tid < 16) reinterpret_cast<int*>(smem + PI_FLAG_OFF)[tid] = 0;
  __syncthreads();                       // the only barrier: counters zeroed before anyone polls
  if (my_panels <= 0) return;
  bool ok = true;

  if (wave < 4) {
    // ================================================= MFMA waves ======================================================
    // Two fragment sets: the reads of step s + 1 are issued BEFORE the 21 MFMAs of step s and return underneath them; the
    // readiness of step s + 2 is polled by a read issued at the same point and looked at one step later.
    const int l15 = lane & 15, g = lane >> 4;
    f32x4_t acc[PI_MT][PI_NT];
    bf16x8_t af0[PI_MT], wf0[PI_NT], af1[PI_MT], wf1[PI_NT];
    TR_DECL;
#pragma unroll
    for (int i = 0; i < PI_MT; ++i)
#pragma unroll
      for (int jn = 0; jn < PI_NT; ++jn) acc[i][jn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto reads = [&](int s, bf16x8_t (&af)[PI_MT], bf16x8_t (&wf)[PI_NT]) {
      const char* pa = smem + PI_A_OFF + (s & (PI_RA - 1)) * PI_A_SLOT;
      const char* pw = smem + PI_W_OFF + (s % PI_RW) * PI_W_SLOT;
#pragma unroll
      for (int nt = 0; nt < PI_NT; ++nt) wf[nt] = rd32i(pw, wave * 48 + nt * 16 + l15, g);
#pragma unroll
      for (int mt = 0; mt < PI_MT; ++mt) af[mt] = rd32i(pa, mt * 16 + l15, g);
    };
    int seen = 0;                        // ready steps as of the last poll (min over the two loaders)
    u32x4_t pfr = {0u, 0u, 0u, 0u};
    auto step = [&](int s, bf16x8_t (&ca)[PI_MT], bf16x8_t (&cw)[PI_NT], bf16x8_t (&na)[PI_MT], bf16x8_t (&nw)[PI_NT]) {
      TR_START();
      const bool more = s + 1 < S;
      if (more) {
        if (seen < s + 2 && ok) ok = wait_flag<4>(PI_F_READY_A, s + 2);     // normally known from the poll of the previous step
        reads(s + 1, na, nw);
        asm volatile("ds_read_b128 %0, %1" : "=v"(pfr) : "v"((unsigned)PI_F_READY_A) : "memory");
      }
      TR_MARK(tr_a);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mt = 0; mt < PI_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < PI_NT; ++nt) acc[mt][nt] = mfma16i(cw[nt], ca[mt], acc[mt][nt]);
      __builtin_amdgcn_s_setprio(0);
      TR_MARK(tr_c);
      if (more) {
        // next fragments and the poll through the lgkmcnt wait as tied operands; then the slot of step s + 1 is free
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(nw[0]), "+v"(nw[1]), "+v"(nw[2]), "+v"(na[0]), "+v"(na[1]), "+v"(na[2]), "+v"(na[3]), "+v"(na[4]), "+v"(na[5]), "+v"(na[6]), "+v"(pfr)
                     :
                     : "memory");
        seen = __builtin_amdgcn_readfirstlane((int)min(min(pfr[0], pfr[1]), min(pfr[2], pfr[3])));
        flag_store(PI_F_CONSUMED + wave * 4, s + 2);
      }
      TR_MARK(tr_b);
      if ((s + 1) % nkt == 0) {
        // ---- tile done: acc + bias into the staging image once the storers have left the previous one ----------------
        const int tile = s / nkt;
        const int q = tile & 3;
        f32x4_t bias4[PI_NT];
#pragma unroll
        for (int nt = 0; nt < PI_NT; ++nt)
          bias4[nt] = p.bias ? *reinterpret_cast<const f32x4_t*>(p.bias + q * PI_BN + wave * 48 + nt * 16 + g * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (ok) ok = wait_flag<2>(PI_F_DONE, tile);
#pragma unroll
        for (int mt = 0; mt < PI_MT; ++mt) {
          const int r = mt * 16 + l15;
          if (r < PI_ROWS) {
#pragma unroll
            for (int nt = 0; nt < PI_NT; ++nt)
              *reinterpret_cast<f32x4_t*>(smem + PI_ST_OFF + stage_off_i(r, wave * 12 + nt * 4 + g)) = acc[mt][nt] + bias4[nt];
          }
#pragma unroll
          for (int nt = 0; nt < PI_NT; ++nt) acc[mt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flag_store(PI_F_STAGED + wave * 4, tile + 1);
        TR_MARK(tr_d);
      }
    };
    if (ok) ok = wait_flag<4>(PI_F_READY_A, 1);
    reads(0, af0, wf0);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(wf0[0]), "+v"(wf0[1]), "+v"(wf0[2]), "+v"(af0[0]), "+v"(af0[1]), "+v"(af0[2]), "+v"(af0[3]), "+v"(af0[4]), "+v"(af0[5]), "+v"(af0[6])
                 :
                 : "memory");
    flag_store(PI_F_CONSUMED + wave * 4, 1);
    for (int s = 0; s < S; s += 2) {       // S is even (nkt is)
      step(s, af0, wf0, af1, wf1);
      step(s + 1, af1, wf1, af0, wf0);
    }
    if (wave == 0) TR_FLUSH(0);
  } else if (wave == 4) {
    // ================================================= A loader ========================================================
    // 98 rows x 4 chunks of 16 B = 392 chunks = 6 whole 1 KB pieces + 8 lanes of a seventh; chunk c -> row c >> 2, 16-byte slot
    // (c & 3) ^ ((row >> 2) & 3) on the SOURCE side (the LDS image is lane-linear)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a_hi, 0, (unsigned)p.M * (unsigned)K * 2u, 0x00020000);
    constexpr int kPer = 7;
    int cur_panel = -1;
    unsigned off[kPer] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
    auto issue = [&](int u) {
      const int tile = u / nkt, t = u - tile * nkt;
      const int pi = tile >> 2;
      if (pi != cur_panel) {            // row offsets of this panel
        cur_panel = pi;
        const int panel = (int)blockIdx.x + pi * (int)gridDim.x;
        const int m0 = panel * rows_per_panel;
        const int m_end = min(m0 + rows_per_panel, p.M);
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
          const int c = i * 64 + lane;
          const int row = c >> 2, kc = (c & 3) ^ sf_swz64(row);
          int ar = m0 + row;
          ar = ar < m_end ? ar : m_end - 1;
          off[i] = ((unsigned)ar * (unsigned)K + kc * 8) * 2u;
        }
      }
      char* dst = smem + PI_A_OFF + (u & (PI_RA - 1)) * PI_A_SLOT;
      const int kof = t * 64;
#pragma unroll
      for (int i = 0; i < kPer - 1; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(dst + i * 1024), 16, (int)off[i], kof, 0, 0);
      if (lane < 8)                     // chunks 384..391: the last two rows
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(dst + 6144), 16, (int)off[6], kof, 0, 0);
    };
    const int lead = PI_RA - 1;
    TR_DECL;
    for (int u = 0; u < lead && u < S; ++u) issue(u);
    for (int i = 0; i < S; ++i) {
      TR_START();
      // step i landed: everything but the pieces of the steps issued after it (i + 1 .. i + lead - 1 at this point) has returned
      const int after = min(lead - 1, S - 1 - i);
      if (after == lead - 1) wait_vmi<kPer * (PI_RA - 2)>(); else wait_vmi<0>();
      flag_store(PI_F_READY_A, i + 1);
      flag_store(PI_F_READY_A + 4, i + 1);
      TR_MARK(tr_a);
      if (i + lead < S) {
        if (ok) ok = wait_flag<4>(PI_F_CONSUMED, i);          // slot of step i - 1: every MFMA wave has read it
        TR_MARK(tr_b);
        issue(i + lead);
        TR_MARK(tr_c);
      }
    }
    TR_FLUSH(8);
  } else if (wave == 5) {
    // ================================================= W loader ========================================================
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, (unsigned)p.N * (unsigned)K * 2u, 0x00020000);
    constexpr int kPer = 12;
    unsigned off[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int c = i * 64 + lane;
      const int row = c >> 2, kc = (c & 3) ^ sf_swz64(row);
      off[i] = ((unsigned)row * (unsigned)K + kc * 8) * 2u;
    }
    auto issue = [&](int u) {
      const int tile = u / nkt, t = u - tile * nkt;
      const int q = tile & 3;
      char* dst = smem + PI_W_OFF + (u % PI_RW) * PI_W_SLOT;
      const int sof = q * PI_BN * K * 2 + t * 64;
#pragma unroll
      for (int i = 0; i < kPer; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(dst + i * 1024), 16, (int)off[i], sof, 0, 0);
    };
    const int lead = PI_RW - 1;
    TR_DECL;
    for (int u = 0; u < lead && u < S; ++u) issue(u);
    for (int i = 0; i < S; ++i) {
      TR_START();
      const int after = min(lead - 1, S - 1 - i);
      if (after == lead - 1) wait_vmi<kPer * (PI_RW - 2)>(); else wait_vmi<0>();
      flag_store(PI_F_READY_W, i + 1);
      flag_store(PI_F_READY_W + 4, i + 1);
      TR_MARK(tr_a);
      if (i + lead < S) {
        if (ok) ok = wait_flag<4>(PI_F_CONSUMED, i);
        TR_MARK(tr_b);
        issue(i + lead);
        TR_MARK(tr_c);
      }
    }
    TR_FLUSH(16);
  } else {
    // ================================================= storers =========================================================
    // A wave-instruction covers FOUR rows: 16 lanes per row, 12 columns per lane (24 bytes of each plane = one 16-byte + one
    // 8-byte access), so every lane works and a row's LayerNorm sums are one 16-lane DPP reduction.  Storer v takes the row
    // groups v, v + 2, ... (13 / 12 of the 25 groups of a 98-row panel).  The residual planes of tile k + 1 are requested as
    // soon as tile k is stored and waited for ONCE (vmcnt(0)) when tile k + 1 is staged, a K loop later: loads are inline asm
    // so that the compiler's own vmcnt bookkeeping (which drains the counter whenever loads and stores are both pending) stays
    // out of the loop; the stores never wait.
    const int v = wave - 6;
    const int rsub = lane >> 4, c16 = lane & 15;
    constexpr int kIter = ((PI_ROWS + 3) / 4 + 1) / 2;       // 13
    const int lane_off = ((v * 4 + rsub) * p.ldc + c16 * 12) * 2;          // row (v * 4 + rsub) of the panel, this lane's 12 columns
    const int iter_stride = 8 * p.ldc * 2;                                 // rows advance by 8 per iteration
    u32x4_t rh4[kIter], rl4[kIter];
    u32x2_t rh2[kIter], rl2[kIter];
    auto tile_base = [&](int tile, int* m0_out) -> size_t {
      const int panel = (int)blockIdx.x + (tile >> 2) * (int)gridDim.x;
      *m0_out = panel * rows_per_panel;
      return ((size_t)*m0_out * (size_t)p.ldc + (size_t)(tile & 3) * PI_BN) * 2;
    };
    auto request = [&](int tile) {
      int m0;
      const size_t base = tile_base(tile, &m0);
      const char* bh = reinterpret_cast<const char*>(p.resid_hi) + base;
      const char* bl = reinterpret_cast<const char*>(p.resid_lo) + base;
      const int m_end = min(m0 + rows_per_panel, p.M);
      int vo = lane_off;                 // advanced per iteration behind an opaque copy: thirteen hoisted offsets would spill
      asm volatile("" : "+v"(vo));
#pragma unroll
      for (int j = 0; j < kIter; ++j, vo += iter_stride) {
        const int r = (j * 2 + v) * 4 + rsub;
        if (r < rows_per_panel && m0 + r < m_end) {
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rh4[j]) : "v"(vo), "s"(bh) : "memory");
          asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "=v"(rh2[j]) : "v"(vo), "s"(bh) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rl4[j]) : "v"(vo), "s"(bl) : "memory");
          asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "=v"(rl2[j]) : "v"(vo), "s"(bl) : "memory");
        }
      }
    };
    TR_DECL;
#pragma unroll
    for (int j = 0; j < kIter; ++j) { rh4[j] = rl4[j] = (u32x4_t){0u, 0u, 0u, 0u}; rh2[j] = rl2[j] = (u32x2_t){0u, 0u}; }
    request(0);
    for (int tile = 0; tile < ntiles; ++tile) {
      TR_START();
      const int q = tile & 3;
      int m0;
      const size_t base = tile_base(tile, &m0);
      char* oh = reinterpret_cast<char*>(p.out_hi) + base;
      char* ol = reinterpret_cast<char*>(p.out_lo) + base;
      const int m_end = min(m0 + rows_per_panel, p.M);
      if (ok) ok = wait_flag<4>(PI_F_STAGED, tile + 1);
      // this tile's residual rows (requested a K loop ago) and the previous tile's stores: everything of this wave has returned
#pragma unroll
      for (int j = 0; j < kIter; ++j)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rh4[j]), "+v"(rh2[j]), "+v"(rl4[j]), "+v"(rl2[j])::"memory");
      TR_MARK(tr_a);
      int vo = lane_off;
      asm volatile("" : "+v"(vo));
#pragma unroll
      for (int j = 0; j < kIter; ++j, vo += iter_stride) {
        const int r = (j * 2 + v) * 4 + rsub;
        const int m = m0 + r;
        const bool live = r < rows_per_panel && m < m_end;
        float s1 = 0.f, s2 = 0.f;
        if (live) {
          unsigned hin[6] = {rh4[j][0], rh4[j][1], rh4[j][2], rh4[j][3], rh2[j][0], rh2[j][1]};
          unsigned lin[6] = {rl4[j][0], rl4[j][1], rl4[j][2], rl4[j][3], rl2[j][0], rl2[j][1]};
          unsigned ho[6], lo[6];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const f32x4_t c = *reinterpret_cast<const f32x4_t*>(smem + PI_ST_OFF + stage_off_i(r, c16 * 3 + k));
            f32x4_t x;
            x[0] = bf2f(hin[2 * k] & 0xffffu) + bf2f(lin[2 * k] & 0xffffu) + p.alpha * c[0];
            x[1] = bf2f(hin[2 * k] >> 16) + bf2f(lin[2 * k] >> 16) + p.alpha * c[1];
            x[2] = bf2f(hin[2 * k + 1] & 0xffffu) + bf2f(lin[2 * k + 1] & 0xffffu) + p.alpha * c[2];
            x[3] = bf2f(hin[2 * k + 1] >> 16) + bf2f(lin[2 * k + 1] >> 16) + p.alpha * c[3];
            ho[2 * k] = pack_bf2(x[0], x[1]); ho[2 * k + 1] = pack_bf2(x[2], x[3]);
            lo[2 * k] = pack_bf2(x[0] - bf2f(ho[2 * k] & 0xffffu), x[1] - bf2f(ho[2 * k] >> 16));
            lo[2 * k + 1] = pack_bf2(x[2] - bf2f(ho[2 * k + 1] & 0xffffu), x[3] - bf2f(ho[2 * k + 1] >> 16));
            s1 += (x[0] + x[1]) + (x[2] + x[3]);
            s2 += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
          }
          *reinterpret_cast<u32x4_t*>(oh + vo) = (u32x4_t){ho[0], ho[1], ho[2], ho[3]};
          *reinterpret_cast<u32x2_t*>(oh + vo + 16) = (u32x2_t){ho[4], ho[5]};
          *reinterpret_cast<u32x4_t*>(ol + vo) = (u32x4_t){lo[0], lo[1], lo[2], lo[3]};
          *reinterpret_cast<u32x2_t*>(ol + vo + 16) = (u32x2_t){lo[4], lo[5]};
        }
        if (p.ln_stats_out) {            // 16 lanes of a row: row_shr scan, the total lands in the row's lane 15
          s1 = dpp_add<0x118, 0xf>(dpp_add<0x114, 0xf>(dpp_add<0x112, 0xf>(dpp_add<0x111, 0xf>(s1))));
          s2 = dpp_add<0x118, 0xf>(dpp_add<0x114, 0xf>(dpp_add<0x112, 0xf>(dpp_add<0x111, 0xf>(s2))));
          if (c16 == 15 && live)
            *reinterpret_cast<u32x2_t*>(p.ln_stats_out + (size_t)m * 8 + q * 2) = (u32x2_t){__float_as_uint(s1), __float_as_uint(s2)};
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // staged values read: the image may be overwritten
      flag_store(PI_F_DONE + v * 4, tile + 1);
      TR_MARK(tr_b);
      if (tile + 1 < ntiles) request(tile + 1);
      TR_MARK(tr_c);
    }
    if (v == 0) TR_FLUSH(24);
  }
  if (!ok && lane == 0 && fail_flag) *fail_flag = 1;
}

static int pipe_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus < 16) cus = 256;
  }
  return cus;
}

static bool pipe_plan(int M, int* panels_out, int* rows_out) {
  const int cus = pipe_cus();
  int panels = (M + PI_ROWS - 1) / PI_ROWS;
  if (panels < cus) return false;                              // fewer panels than CUs: the one-tile-per-CU kernels fill the chip better
  panels = (panels + cus - 1) / cus * cus;                     // whole rounds of the persistent grid
  const int rows = (M + panels - 1) / panels;
  if (rows > PI_ROWS || rows * 100 < PI_MT * 16 * 60) return false;
  *panels_out = (M + rows - 1) / rows; *rows_out = rows;
  return true;
}

static int* pipe_fail_flag() {       // device int, sticky: a bounded spin gave up (never expected; checked by the tests through sf_pipe_failed)
  static int* flag[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!flag[dev]) {
    if (hipMalloc((void**)&flag[dev], sizeof(int)) != hipSuccess) return nullptr;
    (void)hipMemset(flag[dev], 0, sizeof(int));
  }
  return flag[dev];
}
int sf_gemm_pipe_failed() {
  int* f = pipe_fail_flag();
  int h = 0;
  if (f) (void)hipMemcpy(&h, f, sizeof(int), hipMemcpyDeviceToHost);
  return h;
}

bool sf_gemm_pipe_supported(const SfGemmArgs& a, bool split) {
  static const bool on = getenv("SF_PANEL_PIPE") != nullptr && atoi(getenv("SF_PANEL_PIPE")) != 0;
  if (!on) return false;
  if (split || a.N != 768 || a.grp_rows > 0 || a.epi != SF_EPI_RESID_F32) return false;
  if (!a.resid_hi || !a.resid_lo || !a.out_hi || !a.out_lo || a.resid_mod > 0 || a.out_f32) return false;
  if (a.ln_stats || (a.ln_stats_out && !a.ln_stats_wide)) return false;
  if (a.K % 32 || a.K < 256 || a.ldc % 8) return false;
  if ((size_t)a.M * a.K * 2 >= ((size_t)1 << 32)) return false;
  static const int max_k = getenv("SF_PANEL_PIPE_MAX_K") ? atoi(getenv("SF_PANEL_PIPE_MAX_K")) : 768;
  if (a.K > max_k) return false;
  int panels, rows;
  return pipe_plan(a.M, &panels, &rows);
}

hipError_t sf_launch_gemm_pipe(const SfGemmArgs& a, hipStream_t s) {
  int panels = 0, rows = 0;
  if (!pipe_plan(a.M, &panels, &rows)) return hipErrorInvalidValue;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PI_LDS_BYTES);
  const int cus = pipe_cus();
  hipLaunchKernelGGL(sf_gemm_pipe_kernel, dim3(panels < cus ? panels : cus), dim3(PI_THREADS), PI_LDS_BYTES, s, a, rows, panels, pipe_fail_flag());
  return hipGetLastError();
}
