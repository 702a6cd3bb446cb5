// 256x256x64 "8-phase" MFMA GEMM for the large projections of the encoder (qkv, MLP up/down):
//   C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue), bf16 operands, fp32 accumulate.
//
// gfx950 structure (one workgroup per CU, 8 waves = 2 per SIMD, 128 KB of LDS):
//   * wave (wm, wn) of a 2x4 grid owns a 128x64 block of C, processed as four 64x32 quadrants;
//     one quadrant x one K-tile (BK = 64) = 16 MFMA 16x16x32 = one PHASE; 4 phases per K-tile.
//   * a K-tile is staged as four 16 KB "pieces" in consumption order: Bp0, Ap0, Bp1, Ap1
//     (Bp_q = the nq=q column halves of all four wave columns, Ap_q = the mq=q row halves of both
//     wave rows), so each piece is needed one phase later than the previous one.  Every phase issues
//     ONE piece (2 x global_load_lds_dwordx4 per lane) six pieces ahead of its use, into a 2-deep
//     ring per piece; `s_waitcnt vmcnt(8)` (never 0 in steady state) leaves four pieces in flight
//     across the barriers.
//   * the two wave rows run STAGGERED by one barrier: while wm=0 issues MFMAs, wm=1 issues its
//     ds_reads + LDS-DMA, and vice versa, so the matrix pipe of every SIMD always has one wave in a
//     pure-MFMA segment (raised priority) next to one in a memory segment.
//   * hazards are placed by count, not by luck: a piece is read one phase after the wait that retires
//     it (RAW: issuing waves' vmcnt -> barrier -> ds_read) and its ring slot is re-staged >= 2 phases
//     after its last ds_read (WAR, with the stagger).  See DESIGN.md "GEMM-256 schedule".
//   * LDS image of a piece is [128 rows][64 k] bf16, lane-linear for the DMA, with the 16-byte slot
//     XOR (row>>1)&7 applied to the SOURCE address and mirrored on ds_read_b128.
#include "sf_common.h"
#include "sf_switches.h"
#include <cstdlib>

#define G_THREADS 512
#define PIECE_BYTES 16384

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

SF_DEVICE f32x4_t mfma16b(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}

SF_DEVICE bf16x8_t rd_frag(const char* piece, int row, int kc) {
  return *reinterpret_cast<const bf16x8_t*>(piece + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
}
// bf16x3 (SPLIT) piece: [hi plane | lo plane], each [128 rows][32 k] = 64-byte rows, 16-byte slot XOR sf_swz64(row)
SF_DEVICE bf16x8_t rd_frag_s(const char* piece, int plane, int row, int g) {
  return *reinterpret_cast<const bf16x8_t*>(piece + plane * 8192 + row * 64 + ((g ^ sf_swz64(row)) << 4));
}

#ifdef SF_G256_TRACE
__device__ unsigned long long g256_trace[16];
#endif

template <int N>
SF_DEVICE void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BM = 256, or 224 when that tiles M with fewer idle tile slots (M = 25088 = 112 x 224: the qkv GEMM runs
// 4 full rounds of 224-row tiles instead of 3.45 -> 4 rounds of 256-row ones).  With BM = 224 a wave row
// owns 112 rows: quadrant mq = 0 has 4 m-tiles, mq = 1 has 3 (its A piece is padded with clamped rows).
#define G256_EPI_BF16_AUX 5     // kernel-internal: SF_EPI_BF16 with the training-step aux epilogue compiled in

//
// SPLIT = the fp32-accurate mode (SF_COMPUTE_BF16X3): both operands arrive as hi + lo bf16 planes.  A K-tile is 32
// wide and a 16 KB piece holds the hi plane and the lo plane of the same [128 rows][32 k] block (one DMA instruction
// each), so ring, piece order, phase schedule and fragment registers are those of the bf16 kernel with "k-step" read as
// "plane"; a phase is 24 MFMAs (hi*hi + hi*lo + lo*hi per fragment pair) over the same 12 ds_read_b128 + 2 DMA.
// WIDE (lab only, tools/g256_trace_lab.hip; not instantiated in the library): two quadrants per barrier interval (32 MFMAs,
// 48 in SPLIT) instead of one -- four intervals per K-tile instead of eight, same pieces, ring and registers.  It saves
// 11 % of the main loop's CYCLES (1162 against 2 x 651 per pair of quadrants) and nothing on the wall clock: the chip runs
// these loops against its power budget (1.88 GHz effective on random data) and gives the cycles back as clock.
template <int EPI, bool LNF, int BM, bool SPLIT = false, bool WIDE = false>
__global__ __launch_bounds__(G_THREADS) void sf_gemm256_kernel(SfGemmArgs p, int ntiles, int stagger_ticks, int stagger_groups) {
  constexpr int HR = BM / 2;                 // rows per wave row
  constexpr int MT1 = (HR - 64) / 16;        // m-tiles of the second row quadrant (BM = 256: 4, 224: 3, 192: 2, 160: 1)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;
  const int tiles_n = p.N >> 8;
  const int K = p.K;
  const int nkt = SPLIT ? (K >> 5) : (K >> 6);   // even, >= 2 (checked by the launcher)
  const int dma_lds = wave * 1024;   // wave-uniform part of the DMA destination inside a piece
  // persistent, XCD-aware walk: in round r the 32 workgroups of XCD x take 32 consecutive tiles
  // (consecutive tiles share the A row panel, so it is fetched into that XCD's L2 once)
  // operands through buffer descriptors: 32-bit per-lane byte offsets + a scalar K offset, so the
  // loop carries no 64-bit address arithmetic (each operand is < 4 GiB: checked by the launcher)
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a_hi, 0, (unsigned)p.M * (unsigned)K * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, (unsigned)p.N * (unsigned)K * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_al = __builtin_amdgcn_make_buffer_rsrc((void*)(SPLIT ? p.a_lo : p.a_hi), 0, (unsigned)p.M * (unsigned)K * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_wl = __builtin_amdgcn_make_buffer_rsrc((void*)(SPLIT ? p.w_lo : p.w_hi), 0, (unsigned)p.N * (unsigned)K * 2u, 0x00020000);
  const int cpx = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
  // walk (bits 8-11 of the fourth argument = column-group width cg in tiles, 0 = the row-major walk above; bit 12 = write-through stores):
  // row-major, an XCD round is 32/tiles_n row panels x ALL column tiles, so every XCD re-fetches the whole W in every round (r05 counters:
  // 1.9-2.1x the algorithmic bytes at the fabric).  With cg > 0 the tile order is column-group-major — group j = columns [c0_j, c0_j + w_j),
  // inside it row panels, inside a panel the group's columns — and XCD x owns the CONTIGUOUS eighth [x * per, (x + 1) * per) of that
  // order, walked 32 tiles a round: its W slice (w_j x 256 x K) stays the same for all its rounds (one switch at a group boundary) and
  // an A row panel is fetched once per column group.  Which tile a workgroup computes changes, nothing inside a tile does: bit-identical.
  const int walk_cg = (stagger_groups >> 8) & 15;
  const bool store_wt = (stagger_groups >> 12) & 1;
  const bool store_nt = (stagger_groups >> 13) & 1;
  stagger_groups &= 255;
  const int walk_ngrp = walk_cg ? (tiles_n + walk_cg - 1) / walk_cg : 1;
  const int walk_per = (ntiles + 7) >> 3;
  const int walk_panels = ntiles / tiles_n;

  // Phase stagger: every CU runs the same tile sequence, so without it all 256 CUs hit their HBM-bound
  // C-tile store phase at the same instant (32 MB bursts at the HBM write rate, nobody computing) and then
  // all compute while HBM idles.  Delaying one third of each XCD's workgroups by 1/3 and another third by
  // 2/3 of a tile period once, at kernel start, keeps the three groups out of phase for the whole launch:
  // one group's stores drain while the other two run their MFMA main loops.
  if (stagger_ticks > 0) {
    const int grp = slot_in_xcd % stagger_groups;
    if (grp) {
      const unsigned long long t0 = wall_clock64();
      const unsigned long long wait = (unsigned long long)grp * (unsigned long long)stagger_ticks;
      while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
  }

#ifdef SF_G256_TRACE
  unsigned tr_read = 0, tr_bar1 = 0, tr_mma = 0, tr_bar2 = 0;
#endif
  for (int round = 0;; ++round) {
#ifdef SF_G256_TRACE
    tr_read = tr_bar1 = tr_mma = tr_bar2 = 0;
#endif
    int tile = (round * 8 + xcd) * cpx + slot_in_xcd;
    int m0, n0;
    if (walk_cg == 15) {
      // row-major, but every (XCD, stagger group) walks its OWN contiguous run of tiles: the workgroups of one stagger group are the ones in
      // phase with each other, so only they share an A row panel through the L2 while it is hot (with slot % groups the nine column tiles of a
      // panel sit in three different phases, 7-14 us apart, and the panel is fetched once per phase)
      const int sg = stagger_groups, grp = slot_in_xcd % sg, ig = slot_in_xcd / sg;
      const int n_g = (cpx - grp + sg - 1) / sg;                       // slots of this group
      const int before = grp * (cpx / sg) + min(grp, cpx % sg);        // slots of the groups in front of it
      const int x0 = xcd * walk_per, x1 = min(x0 + walk_per, ntiles);
      const int lo = x0 + (int)((long)(x1 - x0) * before / cpx), hi = x0 + (int)((long)(x1 - x0) * (before + n_g) / cpx);
      tile = lo + round * n_g + ig;
      if (tile >= hi) break;
      m0 = (tile / tiles_n) * BM; n0 = (tile % tiles_n) << 8;
    } else if (walk_cg) {
      const int tloc = round * cpx + slot_in_xcd;
      tile = xcd * walk_per + tloc;
      if (tloc >= walk_per || tile >= ntiles) break;
      int rest = tile, c0 = 0, w = 0;
      for (int j = 0; j < walk_ngrp; ++j) {                 // balanced widths: the first tiles_n % ngrp groups are one column wider
        w = tiles_n / walk_ngrp + (j < tiles_n % walk_ngrp ? 1 : 0);
        if (rest < walk_panels * w) break;
        rest -= walk_panels * w; c0 += w;
      }
      m0 = (rest / w) * BM; n0 = (c0 + rest % w) << 8;
    } else {
      if (tile >= ntiles) break;
      m0 = (tile / tiles_n) * BM; n0 = (tile % tiles_n) << 8;
    }

    // ---- per-lane DMA source offsets (elements), two 16-byte chunks per piece --------------------
    unsigned offA[2][2], offB[2][2];
    int tid_p = threadIdx.x;
    asm volatile("" : "+v"(tid_p));     // recompute per tile instead of carrying (and spilling) invariants
#pragma unroll
    for (int i = 0; i < (SPLIT ? 1 : 2); ++i) {
      const int c = i * G_THREADS + tid_p;
      // SPLIT: one 16-byte chunk per lane and plane, the same offset in the hi and the lo array
      const int prow = SPLIT ? (c >> 2) : (c >> 3), slot = SPLIT ? (c & 3) : (c & 7);
      const int kc = SPLIT ? (slot ^ sf_swz64(prow)) : (slot ^ ((prow >> 1) & 7));
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int rr = prow & 63;
        if (q == 1 && rr >= MT1 * 16) rr = MT1 * 16 - 1;            // padding rows of the short quadrant
        int ar = m0 + (prow >> 6) * HR + q * 64 + rr;
        ar = ar < p.M ? ar : p.M - 1;
        offA[q][i] = ((unsigned)ar * (unsigned)K + kc * 8) * 2u;      // byte offsets (buffer voffset)
        int br = n0 + (prow >> 5) * 64 + q * 32 + (prow & 31);
        offB[q][i] = ((unsigned)br * (unsigned)K + kc * 8) * 2u;
      }
    }
    // piece j of K-tile t: j = 0 Bp0, 1 Ap0, 2 Bp1, 3 Ap1 ; ring slot (t & 1) * 4 + j
    auto issue = [&](int t, int j) {
      char* dst = smem + ((t & 1) * 4 + j) * PIECE_BYTES + dma_lds;
      const unsigned o0 = (j & 1) ? offA[j >> 1][0] : offB[j >> 1][0];
      const unsigned o1 = (j & 1) ? offA[j >> 1][SPLIT ? 0 : 1] : offB[j >> 1][SPLIT ? 0 : 1];
      const int kof = SPLIT ? t * 64 : t * 128;     // bytes along K: the scalar offset of the buffer load
      if (SPLIT) {
        if (j & 1) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)dst, 16, o0, kof, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_al, (lptr_t)(dst + 8192), 16, o1, kof, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)dst, 16, o0, kof, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_wl, (lptr_t)(dst + 8192), 16, o1, kof, 0, 0);
        }
      } else if (j & 1) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)dst, 16, o0, kof, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(dst + 8192), 16, o1, kof, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)dst, 16, o0, kof, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(dst + 8192), 16, o1, kof, 0, 0);
      }
    };

    f32x4_t acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    bf16x8_t af[4][2], b0[2][2], b1[2][2];
    auto read_b = [&](bf16x8_t (&b)[2][2], int par, int nq) {
      const char* pc = smem + (par * 4 + nq * 2) * PIECE_BYTES;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          b[nt][ks] = SPLIT ? rd_frag_s(pc, ks, wn * 32 + nt * 16 + l15, g) : rd_frag(pc, wn * 32 + nt * 16 + l15, ks * 4 + g);
    };
    auto read_a = [&](int par, int mq) {
      const char* pc = smem + (par * 4 + mq * 2 + 1) * PIECE_BYTES;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          if (mt < (mq ? MT1 : 4))
            af[mt][ks] = SPLIT ? rd_frag_s(pc, ks, wm * 64 + mt * 16 + l15, g) : rd_frag(pc, wm * 64 + mt * 16 + l15, ks * 4 + g);
    };
    auto mma = [&](int mq, int nq, bf16x8_t (&b)[2][2]) {
      __builtin_amdgcn_s_setprio(1);
      if (SPLIT) {      // [..][0] = hi plane, [..][1] = lo plane; the two small products first
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              if (mt < (mq ? MT1 : 4))
                acc[mq][nq][mt][nt] = mfma16b(b[nt][pr == 0 ? 1 : 0], af[mt][pr == 1 ? 1 : 0], acc[mq][nq][mt][nt]);
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              if (mt < (mq ? MT1 : 4)) acc[mq][nq][mt][nt] = mfma16b(b[nt][ks], af[mt][ks], acc[mq][nq][mt][nt]);
      }
      __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: six pieces in flight, the first two landed ------------------------------------
    // (older epilogue stores of the previous tile may still be outstanding: a counted wait stays
    //  correct because loads return in order among themselves -- a pending needed load implies all
    //  later loads pending, i.e. more than N outstanding)
    issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3); issue(1, 0); issue(1, 1);
    if (WIDE) wait_vm<6>(); else wait_vm<8>();      // WIDE reads three pieces in its first interval
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();     // stagger the second wave row by one barrier

#ifdef SF_G256_TRACE      // lab build (tools/g256_trace_lab.hip): where a barrier interval goes, summed per wave over the main loops
#define SF_TS(ACC) do { const unsigned tn_ = (unsigned)__builtin_readcyclecounter(); ACC += tn_ - tprev_; tprev_ = tn_; } while (0)
    unsigned tprev_ = (unsigned)__builtin_readcyclecounter();
#else
#define SF_TS(ACC) do { } while (0)
#endif
#define PHASE(READS, ISSUE_STMT, WAIT_STMT, MMA_STMT) \
  do {                                               \
    READS;                                           \
    ISSUE_STMT;                                      \
    WAIT_STMT;                                       \
    SF_TS(tr_read);                                  \
    __builtin_amdgcn_s_barrier();                    \
    SF_TS(tr_bar1);                                  \
    MMA_STMT;                                        \
    SF_TS(tr_mma);                                   \
    __builtin_amdgcn_s_barrier();                    \
    SF_TS(tr_bar2);                                  \
  } while (0)

    int t = 0;
    if (WIDE) {
      // interval A of K-tile t reads Bp0, Ap0, Bp1 and runs quadrants (0,0), (0,1); interval B reads Ap1 and runs (1,1), (1,0).
      // In flight behind the wait: A keeps the four pieces of K-tile t+1 (the next read, Ap1 of t, has landed);
      // B keeps (t+1,3), (t+2,0), (t+2,1) (the next reads, pieces 0..2 of t+1, have landed).
      for (; t + 2 < nkt; t += 2) {
        PHASE((read_b(b0, 0, 0), read_a(0, 0), read_b(b1, 0, 1)), (issue(t + 1, 2), issue(t + 1, 3)), wait_vm<8>(), (mma(0, 0, b0), mma(0, 1, b1)));
        PHASE(read_a(0, 1), (issue(t + 2, 0), issue(t + 2, 1)), wait_vm<6>(), (mma(1, 1, b1), mma(1, 0, b0)));
        PHASE((read_b(b0, 1, 0), read_a(1, 0), read_b(b1, 1, 1)), (issue(t + 2, 2), issue(t + 2, 3)), wait_vm<8>(), (mma(0, 0, b0), mma(0, 1, b1)));
        PHASE(read_a(1, 1), (issue(t + 3, 0), issue(t + 3, 1)), wait_vm<6>(), (mma(1, 1, b1), mma(1, 0, b0)));
      }
      PHASE((read_b(b0, 0, 0), read_a(0, 0), read_b(b1, 0, 1)), (issue(t + 1, 2), issue(t + 1, 3)), wait_vm<8>(), (mma(0, 0, b0), mma(0, 1, b1)));
      PHASE(read_a(0, 1), (void)0, wait_vm<2>(), (mma(1, 1, b1), mma(1, 0, b0)));
      PHASE((read_b(b0, 1, 0), read_a(1, 0), read_b(b1, 1, 1)), (void)0, wait_vm<0>(), (mma(0, 0, b0), mma(0, 1, b1)));
      PHASE(read_a(1, 1), (void)0, (void)0, (mma(1, 1, b1), mma(1, 0, b0)));
    } else {
    for (; t + 2 < nkt; t += 2) {
      // K-tile t (parity 0): phases issue pieces (t+1,2), (t+1,3), (t+2,0), (t+2,1)
      PHASE((read_b(b0, 0, 0), read_a(0, 0)), issue(t + 1, 2), wait_vm<8>(), mma(0, 0, b0));
      PHASE(read_b(b1, 0, 1), issue(t + 1, 3), wait_vm<8>(), mma(0, 1, b1));
      PHASE(read_a(0, 1), issue(t + 2, 0), wait_vm<8>(), mma(1, 1, b1));
      PHASE((void)0, issue(t + 2, 1), wait_vm<8>(), mma(1, 0, b0));
      // K-tile t+1 (parity 1): pieces (t+2,2), (t+2,3), (t+3,0), (t+3,1)
      PHASE((read_b(b0, 1, 0), read_a(1, 0)), issue(t + 2, 2), wait_vm<8>(), mma(0, 0, b0));
      PHASE(read_b(b1, 1, 1), issue(t + 2, 3), wait_vm<8>(), mma(0, 1, b1));
      PHASE(read_a(1, 1), issue(t + 3, 0), wait_vm<8>(), mma(1, 1, b1));
      PHASE((void)0, issue(t + 3, 1), wait_vm<8>(), mma(1, 0, b0));
    }
    // ---- tail: K-tiles nkt-2 (parity 0) and nkt-1 (parity 1); only two pieces are left to issue ---
    PHASE((read_b(b0, 0, 0), read_a(0, 0)), issue(t + 1, 2), wait_vm<8>(), mma(0, 0, b0));
    PHASE(read_b(b1, 0, 1), issue(t + 1, 3), wait_vm<8>(), mma(0, 1, b1));
    PHASE(read_a(0, 1), (void)0, (void)0, mma(1, 1, b1));
    PHASE((void)0, (void)0, wait_vm<4>(), mma(1, 0, b0));
    PHASE((read_b(b0, 1, 0), read_a(1, 0)), (void)0, wait_vm<2>(), mma(0, 0, b0));
    PHASE(read_b(b1, 1, 1), (void)0, wait_vm<0>(), mma(0, 1, b1));
    PHASE(read_a(1, 1), (void)0, (void)0, mma(1, 1, b1));
    PHASE((void)0, (void)0, (void)0, mma(1, 0, b0));
    }
#undef PHASE
#undef SF_TS
#ifdef SF_G256_TRACE
    if (blockIdx.x == SF_G256_TRACE && lane == 0 && wn == 0) {
      unsigned long long* tb = g256_trace + wm * 8;
      tb[0] += tr_read; tb[1] += tr_bar1; tb[2] += tr_mma; tb[3] += tr_bar2; tb[4] += (unsigned long long)(nkt * 4 / (WIDE ? 2 : 1));
    }
#endif
    if (wm == 0) __builtin_amdgcn_s_barrier();     // balance the stagger: every ring read is retired

    // ---- epilogue: stage the C tile in LDS (the ring is idle), then whole-row 16-byte stores -------
    if (p.act == 99) {   // lab: no stores (keeps the accumulators live through an impossible branch)
      float sacc = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 2; ++d) sacc += acc[a][b][c][d][0] + acc[a][b][c][d][1] + acc[a][b][c][d][2] + acc[a][b][c][d][3];
      if (sacc == 123.456f) p.out_hi[tid] = 1;
      continue;
    }
    // lane-derived epilogue indices are re-materialised from an opaque copy of the thread id so the
    // compiler does not hoist them (loop-invariant over tiles) across the register-starved K loop
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int l15 = tid_e & 15, g = (tid_e >> 4) & 3;
    const int tid = tid_e;
    f32x4_t bias4[2][2];
#pragma unroll
    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bias4[nq][nt] = p.bias ? *reinterpret_cast<const f32x4_t*>(p.bias + n0 + wn * 64 + nq * 32 + nt * 16 + g * 4)
                               : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // LayerNorm folded into this Linear: s_n = sum_k W'[n,k] per output column
    f32x4_t lns4[2][2];
    constexpr bool ln_fold = LNF;
    if (ln_fold) {
#pragma unroll
      for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) lns4[nq][nt] = *reinterpret_cast<const f32x4_t*>(p.ln_s + n0 + wn * 64 + nq * 32 + nt * 16 + g * 4);
    }
    if (!SPLIT && (EPI == SF_EPI_BF16 || EPI == SF_EPI_ACT_BF16 || EPI == G256_EPI_BF16_AUX)) {
      // [256 rows][32 chunks of 16 B], chunk index XOR (row & 31)
#pragma unroll
      for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int mt = 0; mt < (mq ? MT1 : 4); ++mt) {
          const int r = wm * HR + mq * 64 + mt * 16 + l15;
          float ln_mu = 0.f, ln_r = 1.f;
          if (ln_fold) {
            const int mrow = min(m0 + r, p.M - 1);
            f32x4_t st = *reinterpret_cast<const f32x4_t*>(p.ln_stats + (size_t)mrow * (p.ln_stats_wide ? 8 : 4));
            if (p.ln_stats_wide) {      // four pairs per row (one per 192-column quarter of the producer), fixed order
              const f32x4_t s2 = *reinterpret_cast<const f32x4_t*>(p.ln_stats + (size_t)mrow * 8 + 4);
              st[0] += st[2]; st[1] += st[3];
              st[2] = s2[0] + s2[2]; st[3] = s2[1] + s2[3];
            }
            const float invd = 1.0f / (float)K;
            ln_mu = (st[0] + st[2]) * invd;
            ln_r = rsqrtf((st[1] + st[3]) * invd - ln_mu * ln_mu + p.ln_eps);
          }
#pragma unroll
          for (int nq = 0; nq < 2; ++nq)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const int nl = wn * 64 + nq * 32 + nt * 16 + g * 4;
              f32x4_t v = acc[mq][nq][mt][nt];
              if (ln_fold) v = ln_r * (v - ln_mu * lns4[nq][nt]);
              v += bias4[nq][nt];
              if (EPI == SF_EPI_ACT_BF16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_bf16(v[j]);   // erf-GELU only (launcher checks)
              }
              const u32x2_t hv = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
              *reinterpret_cast<u32x2_t*>(smem + r * 512 + ((((nl >> 3)) ^ (r & 31)) << 4) + (nl & 4) * 2) = hv;
            }
        }
      __syncthreads();
#pragma unroll 4
      for (int it = 0; it < BM / 16; ++it) {
        const int idx = it * G_THREADS + tid;
        const int r = idx >> 5, c = idx & 31;
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + r * 512 + ((c ^ (r & 31)) << 4));
        const int m = m0 + r;
        if (m < p.M) {
          size_t orow = (size_t)m;
          if (p.grp_rows > 0) orow = sf_out_row(p, m);
          const size_t o = orow * (size_t)p.ldc + n0 + c * 8;
          if (EPI == G256_EPI_BF16_AUX && p.aux_mode == 1) {               // training forward: the GELU of the (bf16) pre-activation
            u32x4_t a;
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = pack_bf2(gelu_bf16(bf2f(v[j] & 0xffffu)), gelu_bf16(bf2f(v[j] >> 16)));
            *reinterpret_cast<u32x4_t*>(p.aux + o) = a;
          } else if (EPI == G256_EPI_BF16_AUX && p.aux_mode == 2) {        // training backward: d pre = d act * gelu'(pre)
            const u32x4_t a = *reinterpret_cast<const u32x4_t*>(p.aux + o);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              v[j] = pack_bf2(bf2f(v[j] & 0xffffu) * gelu_grad_fast(bf2f(a[j] & 0xffffu)),
                              bf2f(v[j] >> 16) * gelu_grad_fast(bf2f(a[j] >> 16)));
          }
          if (store_wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p.out_hi + o), "v"(v) : "memory");     // leaves the XCD's L2 to the operands
          else if (store_nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p.out_hi + o), "v"(v) : "memory");      // streamed once: do not displace the residual planes from L2 / Infinity Cache
          else *reinterpret_cast<u32x4_t*>(p.out_hi + o) = v;
        }
      }
      __syncthreads();   // staging reads retired before the next tile's DMA lands in the ring
    } else {
      // fp32 outputs (and the hi + lo bf16 planes of the accurate mode, split from the staged fp32 values on the way
      // out): two passes of [128 rows][64 chunks of 16 B], chunk index XOR (row & 63)
#pragma unroll
      for (int mq = 0; mq < 2; ++mq) {
#pragma unroll
        for (int mt = 0; mt < (mq ? MT1 : 4); ++mt) {
          const int r = wm * 64 + mt * 16 + l15;
          float ln_mu = 0.f, ln_r = 1.f;
          if (ln_fold) {      // fp32-accurate mode: A = the hi + lo planes of the raw residual rows, wide statistics from their producer
            const int mrow = min(m0 + wm * HR + mq * 64 + mt * 16 + l15, p.M - 1);
            const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(p.ln_stats + (size_t)mrow * 8);
            const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(p.ln_stats + (size_t)mrow * 8 + 4);
            const float invd = 1.0f / (float)K;
            ln_mu = ((s0[0] + s0[2]) + (s1[0] + s1[2])) * invd;        // four pairs (the fourth is zero at K = 768: x + 0 is exact)
            ln_r = rsqrtf(((s0[1] + s0[3]) + (s1[1] + s1[3])) * invd - ln_mu * ln_mu + p.ln_eps);
          }
#pragma unroll
          for (int nq = 0; nq < 2; ++nq)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const int nl = wn * 64 + nq * 32 + nt * 16 + g * 4;
              f32x4_t v = acc[mq][nq][mt][nt];
              if (ln_fold) v = ln_r * (v - ln_mu * lns4[nq][nt]);
              v += bias4[nq][nt];
              if (EPI == SF_EPI_ACT_BF16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = p.act == 0 ? apply_act_fast(v[j], 0) : apply_act(v[j], p.act);   // erf to 1.5e-7 abs (ocml erff costs ~100 us per launch here)
              }
              *reinterpret_cast<f32x4_t*>(smem + r * 1024 + (((nl >> 2) ^ (r & 63)) << 4)) = v;
            }
        }
        __syncthreads();
        if (EPI == SF_EPI_BF16 || EPI == SF_EPI_ACT_BF16) {
          // hi + lo planes: a lane takes 8 columns (two staged chunks) and writes 16 bytes to each plane
#pragma unroll 4
          for (int it = 0; it < 8; ++it) {
            const int idx = it * G_THREADS + tid;
            const int r = idx >> 5, c2 = idx & 31;
            const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(smem + r * 1024 + (((2 * c2) ^ (r & 63)) << 4));
            const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(smem + r * 1024 + (((2 * c2 + 1) ^ (r & 63)) << 4));
            const int m = m0 + (r >> 6) * HR + mq * 64 + (r & 63);
            if (m < p.M && (r & 63) < (mq ? MT1 : 4) * 16) {
              size_t orow = (size_t)m;
              if (p.grp_rows > 0) orow = sf_out_row(p, m);
              const size_t o = orow * (size_t)p.ldc + n0 + c2 * 8;
              u32x4_t h, l;
              h[0] = pack_bf2(v0[0], v0[1]); h[1] = pack_bf2(v0[2], v0[3]); h[2] = pack_bf2(v1[0], v1[1]); h[3] = pack_bf2(v1[2], v1[3]);
              l[0] = pack_bf2(v0[0] - bf2f(h[0] & 0xffffu), v0[1] - bf2f(h[0] >> 16));
              l[1] = pack_bf2(v0[2] - bf2f(h[1] & 0xffffu), v0[3] - bf2f(h[1] >> 16));
              l[2] = pack_bf2(v1[0] - bf2f(h[2] & 0xffffu), v1[1] - bf2f(h[2] >> 16));
              l[3] = pack_bf2(v1[2] - bf2f(h[3] & 0xffffu), v1[3] - bf2f(h[3] >> 16));
              *reinterpret_cast<u32x4_t*>(p.out_hi + o) = h;
              if (p.out_lo) *reinterpret_cast<u32x4_t*>(p.out_lo + o) = l;
            }
          }
        } else if (EPI == SF_EPI_RESID_F32 && p.resid_hi) {
          // residual stream as hi + lo bf16 planes (they ARE the operand planes of the folded Linear that follows) — the fp32-accurate
          // mode at any width, and since round 5 the bf16 mode at widths the panel kernel does not take (D = 1024): a lane takes
          // 8 columns — 16 bytes of each plane in, 16 out — and the 32 lanes of a row reduce this 256-column tile's
          // {sum x, sum x^2} into pair n0 / 256 of the row's wide statistics (up to four pairs: N <= 1024)
#pragma unroll 2
          for (int it = 0; it < 8; ++it) {
            const int idx = it * G_THREADS + tid;
            const int r = idx >> 5, c2 = idx & 31;
            const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(smem + r * 1024 + (((2 * c2) ^ (r & 63)) << 4));
            const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(smem + r * 1024 + (((2 * c2 + 1) ^ (r & 63)) << 4));
            const int m = m0 + (r >> 6) * HR + mq * 64 + (r & 63);
            const bool ok = m < p.M && (r & 63) < (mq ? MT1 : 4) * 16;        // uniform over the 32 lanes of a row
            float s1 = 0.f, s2 = 0.f;
            const size_t o = (size_t)(ok ? m : 0) * (size_t)p.ldc + n0 + c2 * 8;
            if (ok) {
              const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(p.resid_hi + o), li = *reinterpret_cast<const u32x4_t*>(p.resid_lo + o);
              u32x4_t l2i = {0u, 0u, 0u, 0u};
              if (p.resid_lo2) l2i = *reinterpret_cast<const u32x4_t*>(p.resid_lo2 + o);
              float x[8];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const f32x4_t& vv = j < 2 ? v0 : v1;
                x[2 * j] = (bf2f(hi[j] & 0xffffu) + bf2f(li[j] & 0xffffu)) + bf2f(l2i[j] & 0xffffu) + p.alpha * vv[(2 * j) & 3];
                x[2 * j + 1] = (bf2f(hi[j] >> 16) + bf2f(li[j] >> 16)) + bf2f(l2i[j] >> 16) + p.alpha * vv[(2 * j + 1) & 3];
              }
              u32x4_t h, l, l2;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                h[j] = pack_bf2(x[2 * j], x[2 * j + 1]);
                const float r0 = x[2 * j] - bf2f(h[j] & 0xffffu), r1 = x[2 * j + 1] - bf2f(h[j] >> 16);
                l[j] = pack_bf2(r0, r1);
                l2[j] = pack_bf2(r0 - bf2f(l[j] & 0xffffu), r1 - bf2f(l[j] >> 16));
                s1 += x[2 * j] + x[2 * j + 1];
                s2 += x[2 * j] * x[2 * j] + x[2 * j + 1] * x[2 * j + 1];
              }
              *reinterpret_cast<u32x4_t*>(p.out_hi + o) = h;
              *reinterpret_cast<u32x4_t*>(p.out_lo + o) = l;
              if (p.out_lo2) *reinterpret_cast<u32x4_t*>(p.out_lo2 + o) = l2;
            }
#pragma unroll
            for (int sh = 1; sh <= 16; sh <<= 1) {
              s1 += __shfl_xor(s1, sh, 64);
              s2 += __shfl_xor(s2, sh, 64);
            }
            if (ok && c2 == 0)
              *reinterpret_cast<u32x2_t*>(p.ln_stats_out + (size_t)m * 8 + (size_t)(n0 >> 8) * 2) = (u32x2_t){__float_as_uint(s1), __float_as_uint(s2)};
          }
        } else {
#pragma unroll 4
          for (int it = 0; it < 16; ++it) {
            const int idx = it * G_THREADS + tid;
            const int r = idx >> 6, c = idx & 63;
            f32x4_t v = *reinterpret_cast<const f32x4_t*>(smem + r * 1024 + ((c ^ (r & 63)) << 4));
            const int m = m0 + (r >> 6) * HR + mq * 64 + (r & 63);
            if (m < p.M && (r & 63) < (mq ? MT1 : 4) * 16) {
              size_t orow = (size_t)m;
              if (p.grp_rows > 0) orow = sf_out_row(p, m);
              const size_t o = orow * (size_t)p.ldc + n0 + c * 4;
              if (EPI == SF_EPI_RESID_F32) v = *reinterpret_cast<const f32x4_t*>(p.resid + o) + p.alpha * v;
              *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
            }
          }
        }
        __syncthreads();
      }
    }
  }
}

bool sf_gemm256_supported(const SfGemmArgs& a, bool split) {
  if (split && (!a.a_lo || !a.w_lo || a.aux_mode || sf_sw(SW_DISABLE_G256_SPLIT))) return false;
  if (split && a.ln_stats && (!a.ln_stats_wide || !a.ln_s || (a.K % 256) || a.K > 1024 || (a.epi != SF_EPI_BF16 && a.epi != SF_EPI_ACT_BF16))) return false;
  if (a.epi == SF_EPI_EMBED_F32) return false;
  bool planes = false;
  if (a.epi == SF_EPI_RESID_F32 && a.out_hi) {
    // No plain bf16 copy of the new residual here (panel kernel).  The residual as hi + lo planes, in and out, with the wide LayerNorm
    // statistics of the Linear that follows (one pair per 256-column tile: N <= 1024): the fp32-accurate mode, and the bf16 mode at
    // widths the panel kernel does not take
    if (!a.out_lo || !a.resid_hi || !a.resid_lo || !a.ln_stats_out || !a.ln_stats_wide || (a.N % 256) || a.N > 1024 || a.grp_rows > 0) return false;
    if (!split && (a.resid_lo2 || a.out_lo2)) return false;
    planes = true;
  } else if (a.resid_hi) {
    return false;
  }
  if (!split && a.epi == SF_EPI_ACT_BF16 && a.act != 0 && a.act != 99) return false;   // other activations: 128^2 kernel
  if (a.K % 128 || a.K < 128) return false;
  int min_n = 1024;                             // N = 768: 294 tiles on 256 CUs -> the panel / 128^2 kernels win
  if (split) {      // bf16x3 at the BASELINE batch: 392 / 120 us against 415 / 127 on the 128^2 kernel
    // one or two clips per call (M <= 6272): N = 768 is 60 / 120 tiles of 160 rows on 256 CUs — the 128^2 kernel's 150 / 294 tiles fill the chip
    // better (accurate forward of one clip 5.17 -> 4.42 ms, two clips 6.38 -> 6.28; four clips lose: profiles/r06_accurate_small_batch_ab.txt).
    // Returning false here also turns the accurate mode's LayerNorm fold off for such calls (ln_fold_acc_ok asks this function).
    min_n = a.M <= 6272 ? 1024 : 768;
    if (const char* e = sf_sw(SW_G256_SPLIT_MIN_N)) min_n = atoi(e);
  }
  if (a.N % 256 || a.N < min_n) return false;
  if (a.M < 2048 || (size_t)a.M * a.K * 2 >= ((size_t)1 << 32) || (size_t)a.N * a.K * 2 >= ((size_t)1 << 32)) return false;                 // small problems: the 128x128 kernel fills the chip better
  if (a.out_lo && !split && !planes) return false;
  return true;
}

bool sf_gemm256_aux_supported(const SfGemmArgs& a) {
  return a.epi == SF_EPI_BF16 && a.aux != nullptr && sf_gemm256_supported(a, false);
}

static int g256_grid() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (const char* e = sf_sw(SW_ASSUME_CUS)) cus = atoi(e);      // experiment: kernels sized for a CU-masked stream
    if (cus < 8) cus = 256;
    cus &= ~7;      // the XCD-aware walk wants a multiple of 8
  }
  return cus;
}

template <int BM, bool SPLIT_ONLY = false>
static hipError_t launch_bm(const SfGemmArgs& a_in, hipStream_t s) {
  SfGemmArgs a = a_in;
  if (SF_LAB_SWITCH("SF_G256_LAB_NOSTORE")) a.act = 99;      // lab builds only: main loops without stores (results are discarded)
  const int tiles = ((a.M + BM - 1) / BM) * (a.N / 256);
  const size_t lds = 8 * PIECE_BYTES;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
    if constexpr (!SPLIT_ONLY) {
#define SF_ATTR(E, L) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm256_kernel<E, L, BM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      SF_ATTR(SF_EPI_F32, false) SF_ATTR(SF_EPI_BF16, false) SF_ATTR(SF_EPI_ACT_BF16, false) SF_ATTR(SF_EPI_RESID_F32, false)
      SF_ATTR(SF_EPI_BF16, true) SF_ATTR(SF_EPI_ACT_BF16, true) SF_ATTR(G256_EPI_BF16_AUX, false)
#undef SF_ATTR
    }
#define SF_ATTR(E) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm256_kernel<E, false, BM, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SF_ATTR(SF_EPI_F32) SF_ATTR(SF_EPI_BF16) SF_ATTR(SF_EPI_ACT_BF16) SF_ATTR(SF_EPI_RESID_F32)
#undef SF_ATTR
#define SF_ATTR(E) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm256_kernel<E, true, BM, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SF_ATTR(SF_EPI_BF16) SF_ATTR(SF_EPI_ACT_BF16)
#undef SF_ATTR
  }
  const dim3 grid(g256_grid()), block(G_THREADS);
  // stagger as wall_clock64() ticks (sf_wall_clock_ticks), only when the launch runs
  // several rounds of tiles per CU; SF_G256_STAGGER_NS overrides (0 disables) for A/B measurements
  // Phase stagger (see the kernel): `sgroups` groups, each delayed by one more step; step = a fraction of the
  // estimated tile period (MFMA time at ~1 PF + the store burst).  Only when every CU runs >= 3 tiles.
  // Measured optimum on the SigLIP-base shapes: 3 groups, 0.19-0.23 of the period (6-7 us for the MLP up-projection): -5.8 % on the whole forward.
  // SF_G256_STAGGER_NS / SF_G256_STAGGER_PCT / SF_G256_STAGGER_GROUPS override for A/B runs (NS=0 disables).
  int stagger = 0, sgroups = 3;
  if (const char* ge = sf_sw(SW_G256_STAGGER_GROUPS)) sgroups = atoi(ge) > 1 ? atoi(ge) : 2;
  {
    const int rounds = (tiles + (int)grid.x - 1) / (int)grid.x;
    double pct = 21.0;
    if (const char* pe = sf_sw(SW_G256_STAGGER_PCT)) pct = atof(pe);
    const char* env = sf_sw(SW_G256_STAGGER_NS);
    if (env) stagger = sf_wall_clock_ticks(atoi(env));
    else if (rounds >= 3) {
      const double mfma_x = (a.a_lo && a.w_lo) ? 3.0 : 1.0;
      const double tile_ns = mfma_x * 2.0 * BM * 256.0 * a.K / 3.9e3 + 6500.0;     // flops / (3.9 TF per CU) + store burst
      stagger = sf_wall_clock_ticks((int)(tile_ns * pct / 100.0));
    }
  }
  if (const char* only = sf_sw(SW_G256_STAGGER_ONLY)) {      // A/B: "2" = only the GELU up-projection, "1" = only bf16 outputs
    if (a.epi != atoi(only)) stagger = 0;
  }
  {      // tile walk (see the kernel).  SF_G256_WALK = column-group width in tiles (0 = row-major), SF_G256_STORE_WT=1 = sc1 stores of bf16 outputs
    int cg = 0;
    if (const char* we = sf_sw(SW_G256_WALK)) cg = atoi(we);
    if (cg < 0 || cg > 15 || (cg != 15 && cg >= a.N / 256)) cg = 0;      // 15 = row-major with per-stagger-group runs
    sgroups = (sgroups & 255) | (cg << 8) | (sf_sw(SW_G256_STORE_WT) ? 1 << 12 : 0) | (sf_sw(SW_G256_STORE_NT) ? 1 << 13 : 0);
  }
#define SF_LAUNCH256(E, L, SP) hipLaunchKernelGGL((sf_gemm256_kernel<E, L, BM, SP>), grid, block, lds, s, a, tiles, stagger, sgroups)
  if (a.a_lo && a.w_lo) {      // fp32-accurate mode: hi + lo planes of both operands, three products per fragment pair
    if (a.aux_mode) return hipErrorInvalidValue;
    if (a.ln_stats) {          // LayerNorm folded into this Linear (wide statistics of the accurate mode)
      if (!a.ln_stats_wide || !a.ln_s) return hipErrorInvalidValue;
      if (a.epi == SF_EPI_BF16) SF_LAUNCH256(SF_EPI_BF16, true, true);
      else if (a.epi == SF_EPI_ACT_BF16) SF_LAUNCH256(SF_EPI_ACT_BF16, true, true);
      else return hipErrorInvalidValue;
      return hipGetLastError();
    }
    switch (a.epi) {
#define SF_CASE(E) case E: SF_LAUNCH256(E, false, true); break;
      SF_CASE(SF_EPI_F32) SF_CASE(SF_EPI_BF16) SF_CASE(SF_EPI_ACT_BF16) SF_CASE(SF_EPI_RESID_F32)
#undef SF_CASE
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  if constexpr (SPLIT_ONLY) return hipErrorInvalidValue;
  else {
  const bool lnf = a.ln_stats != nullptr;
  if (lnf && (!a.ln_s || (a.epi != SF_EPI_BF16 && a.epi != SF_EPI_ACT_BF16))) return hipErrorInvalidValue;
  switch (a.epi) {
    case SF_EPI_F32: SF_LAUNCH256(SF_EPI_F32, false, false); break;
    case SF_EPI_BF16:
      if (a.aux_mode) {
        if (lnf || !a.aux) return hipErrorInvalidValue;
        SF_LAUNCH256(G256_EPI_BF16_AUX, false, false);
      } else if (lnf) SF_LAUNCH256(SF_EPI_BF16, true, false);
      else SF_LAUNCH256(SF_EPI_BF16, false, false);
      break;
    case SF_EPI_ACT_BF16:
      if (lnf) SF_LAUNCH256(SF_EPI_ACT_BF16, true, false);
      else SF_LAUNCH256(SF_EPI_ACT_BF16, false, false);
      break;
    case SF_EPI_RESID_F32: SF_LAUNCH256(SF_EPI_RESID_F32, false, false); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
  }
}

hipError_t sf_launch_gemm256(const SfGemmArgs& a, hipStream_t s) {
  // pick the row-tile height that wastes fewer tile slots of the persistent grid
  const int g = g256_grid(), nt = a.N / 256;
  auto cost = [&](int bm) { const int tiles = ((a.M + bm - 1) / bm) * nt; return (long)((tiles + g - 1) / g) * bm; };
  if (a.a_lo && a.w_lo && !sf_sw(SW_G256_NO_SHORT_BM)) {
    // bf16x3 runs N = 768 here too (3 column tiles): shorter row tiles fill the second round of the persistent grid.
    // A K-tile of the short quadrant costs the read segment of the other wave row, not its own few MFMAs, hence the
    // per-height time factors (cycles per K-tile: 4 x max(MFMA segment, read segment ~ 350)).
    auto t = [&](int bm, double f) { return (double)cost(bm) / bm * f; };
    const double t256 = t(256, 1632), t224 = t(224, 1472), t192 = t(192, 1370), t160 = t(160, 1268);
    const double best = fmin(fmin(t256, t224), fmin(t192, t160));
    if (const char* fe = sf_sw(SW_G256_FORCE_BM)) {
      switch (atoi(fe)) { case 160: return launch_bm<160, true>(a, s); case 192: return launch_bm<192, true>(a, s); case 224: return launch_bm<224>(a, s); default: return launch_bm<256>(a, s); }
    }
    if (best == t160 && t160 < 0.97 * fmin(t224, t256)) return launch_bm<160, true>(a, s);
    if (best == t192 && t192 < 0.97 * fmin(t224, t256)) return launch_bm<192, true>(a, s);
  }
  if (cost(224) < cost(256) && !sf_sw(SW_G256_NO_BM224)) return launch_bm<224>(a, s);   // env: A/B switch
  return launch_bm<256>(a, s);
}
