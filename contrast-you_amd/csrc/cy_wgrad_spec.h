// Wave-specialised form of the twelve-wave weight-gradient kernel for the 64 x 64 (co, ci) block (16 of the
// U-Net's 21 layers).  Included by cy_wgrad.hip (WgradArgs, Wg12Cfg, WFrag, halo_chunk come from there).
//
// wgrad12_kernel gives every wave both jobs -- request / commit a slice of the next tile, then the MFMAs of one
// kernel row -- and its stamps say where a tile's 15 500 cycles go: 3 900 in the request
// burst (the CU's address unit takes ~40 cycles per 64-lane 16-byte gather, and all twelve waves queue there at the
// same time), 5 700 in an MFMA loop that the LDS side bounds (1.33 fragment reads per MFMA), 1 700 commit, 4 200
// barrier skew -- strictly one after the other, because every wave is in the same phase.
// Here the phases belong to different waves and overlap:
//  * waves 8-11 (256 threads, one per SIMD) are the LOADERS: commit tile j+1 (requested one tile period earlier, so
//    the global latency is covered by a whole tile), request tile j+2, barrier.  65 gathers per tile instead of 84,
//    no MFMA wave ever waits in the address unit's queue;
//  * waves 0-7 own one 32 x 32 block (wave & 3) and taps 0-4 (waves 0-3) or 5-8 (waves 4-7): one dy fragment +
//    4-5 input fragments for 4-5 MFMAs (1.2 fragment reads per MFMA), 80 accumulator registers; SIMD s runs waves
//    s and s + 4 = nine taps of one block each: the MFMA work is balanced over the four SIMDs.
// One barrier per tile as before: at the barrier tile j+1 is complete in the other LDS buffer and tile j has been
// consumed.  The slabs and their reduction are unchanged (bitwise the same summation order per slab).
#pragma once

template <bool V> struct Wg12sFlag {
  static constexpr bool value = V;
};

template <int WCO, int WCI> struct Wg12sCfg : Wg12Cfg<WCO, WCI, 4 / (WCO * WCI)> {
  using B = Wg12Cfg<WCO, WCI, 4 / (WCO * WCI)>;
  static constexpr int NBLK = WCO * WCI, WK = 4 / NBLK;  // 32 x 32 blocks, pixel splits inside a tap group
  static constexpr int SMEM = B::MAIN + 256 * 4 + 344 * 4 + 16;  // + the halo item table
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert(WK == 1 || 8 * 16 * 64 * 4 <= B::MAIN, "pixel-split reduction buffer");
};

// MFMA role of wgrad12s_kernel: block `blk` (co block = blk / WCI, ci block = blk % WCI), taps TAP0 .. TAP0 + NTAP - 1,
// steps wk, wk + WK, ... of every tile (WK = 4 / blocks pixel splits; their accumulators are added through LDS
// at the end, in split order)
// BLK (with the DMA loaders, TH % 4 == 0 and TW % 4 == 0): k-step s is a 4 x 4 patch of the tile (row quad s / (TW/4),
// column block s % (TW/4)) instead of 16 consecutive pixels of the row-major tile.  The reduction order over pixels is
// free as long as dy and the input use the same one, and with patches every fragment address is a LANE CONSTANT
// plus a wave-uniform offset of the step -- the swizzle terms included (a patch origin is a multiple of four pixels).
// Stamps with the loaders on DMA showed the MFMA waves as the long pole at ~500 cycles per k-step for 4-5 MFMAs, and
// the instruction mix why: ~35 vector-ALU instructions of index arithmetic per k-step and wave (pixel -> halo
// pixel, three swizzle variants, four quarter-rate multiplies) on a SIMD that three waves share; here it is 7 adds
// (7 100 -> 6 000 cycles per tile).  Requesting the fragments of the next k-step ahead of the MFMAs (two register
// sets, branch-free so that the LDS waits stay counted) was measured three times on the way and lost every time
// (+5-10 %): the twelve transposed reads per k-step and wave keep the LDS pipe busy ~4.5 cycles each, more reads in
// flight only lengthen its queue.
template <typename T, int WCO, int WCI, int TAP0, int NTAP, bool BLK = false>
__device__ __forceinline__ void wg12s_mfma_role(const WgradArgs& g, unsigned char* smem, const int* s_ktab, int blk,
                                                int wk, int lane, int nsteps, int nmine, int co0, int ci0,
                                                int split) {
  using C = Wg12sCfg<WCO, WCI>;
  constexpr int PA = C::PA, PB = C::PB, HW2 = C::HP, WK = C::WK, NBLK = C::NBLK;
  constexpr bool SWA = PA == 128, SWB = PB == 128;  // pair swizzle of 128-byte pixels
  const int r = lane & 31, h = lane >> 5;
  const int cb = blk / WCI, ib = blk % WCI;
  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // per-lane constants of the transposed fragment reads (see wgrad_kernel)
  const int q = (lane & 15) >> 2, p4 = lane & 3, gsel = (lane >> 4) & 1;
  const int cola = (cb * 32 + 16 * gsel + 4 * p4) * 2;
  const int colb = (ib * 32 + 16 * gsel + 4 * p4) * 2;

  __syncthreads();  // tables published
  __syncthreads();  // first tile staged
#ifdef CY_WGRAD_STAMPS
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool stamping = g.c.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0;
  unsigned long long ph_sum[4] = {0ull, 0ull, 0ull, 0ull}, t_last = stamping ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
  for (int j = 0; j < nmine; ++j) {
    const unsigned char* sDy = smem + (j & 1) * C::BUF;
    const unsigned char* sIn = sDy + C::A_BYTES;
    if constexpr (BLK) {
      const int l = 8 * h + q;  // this lane's pixel of a patch: row l >> 2 (+ 1 for the second read), column l & 3
      const unsigned a_lane = (unsigned)(l * PA + (cola ^ (SWA ? (((l >> 1) & 1) << 6) : 0)));
      const int p1l = (2 * h + 1) * HW2 + q + 1, p2l = p1l + HW2;  // halo pixel of patch (0, 0)'s pixel l / l + 4
      unsigned b1c[3], b2c[3];  // per tap column dw: base of tap (-1, -1); the taps are non-negative immediates
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        b1c[d] = (unsigned)((p1l - HW2 - 1) * PB + (colb ^ (SWB ? ((((p1l + d - 1) >> 1) & 1) << 6) : 0)));
        b2c[d] = (unsigned)((p2l - HW2 - 1) * PB + (colb ^ (SWB ? ((((p2l + d - 1) >> 1) & 1) << 6) : 0)));
      }
      const int cbn = g.TW >> 2;
      int rq = wk / cbn, cbk = wk - rq * cbn;  // (wave-uniform, advanced with the step)
      for (int step = wk; step < nsteps; step += WK) {
        const unsigned offA = (unsigned)(step * 16 * PA), offB = (unsigned)((rq * 4 * HW2 + cbk * 4) * PB);
        const unsigned char* arow = sDy + offA + a_lane;
        const typename Mma<T>::Frag af = WFrag<T>::load(arow, arow + 4 * PA);
        typename Mma<T>::Frag bf[NTAP];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          const int tap = TAP0 + t;
          const int imm = ((tap / 3) * HW2 + (tap % 3)) * PB;
          bf[t] = WFrag<T>::load(sIn + offB + b1c[tap % 3] + imm, sIn + offB + b2c[tap % 3] + imm);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) Mma<T>::mma(af, bf[t], acc[t]);
        __builtin_amdgcn_s_setprio(0);
        cbk += WK;
        while (cbk >= cbn) cbk -= cbn, ++rq;
      }
    } else
    for (int step = wk; step < nsteps; step += WK) {
      const int k1 = step * 16 + 8 * h + q;
      // (k1 >> 1) & 1 == ((k1 + 4) >> 1) & 1: both dy rows share the swizzle term
      const unsigned char* arow = sDy + k1 * PA + (cola ^ (SWA ? (((k1 >> 1) & 1) << 6) : 0));
      const typename Mma<T>::Frag af = WFrag<T>::load(arow, arow + 4 * PA);
      const int p1 = s_ktab[k1], p2 = s_ktab[k1 + 4];  // (pad pixels map to pixel 0; their dy rows are zero)
      // the swizzle term of pixel p + dh*HW2 + dw depends on dw only (HW2 % 4 == 0)
      int c1[3], c2[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        c1[d] = colb ^ (SWB ? ((((p1 + d - 1) >> 1) & 1) << 6) : 0);
        c2[d] = colb ^ (SWB ? ((((p2 + d - 1) >> 1) & 1) << 6) : 0);
      }
      const unsigned char* b1 = sIn + p1 * PB;
      const unsigned char* b2 = sIn + p2 * PB;
      typename Mma<T>::Frag bf[NTAP];
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const int tap = TAP0 + t;
        const int off = ((tap / 3 - 1) * HW2 + (tap % 3 - 1)) * PB;
        bf[t] = WFrag<T>::load(b1 + off + c1[tap % 3], b2 + off + c2[tap % 3]);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < NTAP; ++t) Mma<T>::mma(af, bf[t], acc[t]);
      __builtin_amdgcn_s_setprio(0);
    }
    W12_STAMP(1);
    __syncthreads();  // this tile consumed by every wave, the next one staged
    W12_STAMP(3);
  }
#ifdef CY_WGRAD_STAMPS
  if (stamping && lane == 0) {
    for (int q = 0; q < 4; ++q) g.c.stamps[wave * 8 + q] = ph_sum[q];
    g.c.stamps[wave * 8 + 4] = (unsigned long long)nmine;
  }
#endif

  // ---- write the split's slab: this wave's taps of its block ----
  float* slab = g.ws + (size_t)split * 9 * g.co_pad * g.ci_pad;
  if constexpr (WK > 1) {
    // the WK waves of a (tap group, block) add their accumulators in split order; five rounds for both tap groups
    // (the loader waves have ended: the barrier counts the eight MFMA waves)
    float* red = reinterpret_cast<float*>(smem);  // [wave 0..7][16][64]
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      __syncthreads();
      if (t < NTAP) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) red[(wv * 16 + reg) * 64 + lane] = acc[t < NTAP ? t : 0][reg];
      }
      __syncthreads();
      if (t < NTAP && wk == 0) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float sum = 0.f;
#pragma unroll
          for (int qq = 0; qq < WK; ++qq) sum += red[((wv + qq * NBLK) * 16 + reg) * 64 + lane];
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          slab[((size_t)(TAP0 + t) * g.co_pad + co0 + cb * 32 + row) * g.ci_pad + ci0 + ib * 32 + r] = sum;
        }
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        slab[((size_t)(TAP0 + t) * g.co_pad + co0 + cb * 32 + row) * g.ci_pad + ci0 + ib * 32 + r] = acc[t][reg];
      }
  }
}

// LOADER role by LDS-DMA (DMA = true; 64 x 64 blocks: 128-byte pixels on both sides): the same LDS image -- dy
// [tile pixel][128 B], halo [halo pixel at pitch 36][128 B], pair-swizzled chunks -- filled by
// `buffer_load_dwordx4 ... lds` instead of gather -> register -> ds_write.  A wave-instruction writes 1 KB = eight
// consecutive LDS pixels x eight 16-byte slots; lane (pixel 8q + lane/8, slot lane%8) fetches the channel chunk the
// swizzle puts into that slot (linear destination, permuted SOURCE, swizzled read: cdna_hip_programming.md rule 21;
// the swizzle term ((pixel >> 1) & 1) depends on the lane only, 8q being a multiple of 4, so a lane's chunk is the
// same for every item).  Padding, the pad pixels of the pitch and the rows of a short last k-step read out of range
// = zeros.  The item geometry (tile pixel -> (ty, tx), halo pixel -> (row, column)) is computed once per kernel;
// per tile an item costs a handful of VALU instructions and ONE vector-memory instruction, nothing is held in
// registers, nothing is committed.  The BN+ReLU prologue is an in-place LDS pass over the wave's own items.
// Stamps of the register-staging loaders: ~450 cycles per item on a SIMD they share with two MFMA
// waves, 8 700 cycles per tile against an MFMA loop of 4 000 -- the loaders were the long pole.
template <typename T, int WCO, int WCI>
__device__ __forceinline__ void wg12s_dma_loader(const WgradArgs& g, unsigned char* smem, int lw, int lane, int TH,
                                                 int TW, int npix, int npix_pad, int co0, int ci0, int split,
                                                 int nmine, bool blk_order) {
#if defined(__HIP_DEVICE_COMPILE__)
  using C = Wg12sCfg<WCO, WCI>;
  // 128-byte pixels (64 channels): 8 slots, 8 pixels per instruction, pair swizzle; 64-byte pixels (32 channels):
  // 4 slots, 16 pixels per instruction, no swizzle
  constexpr int PA = C::PA, PB = C::PB, SA = PA / 16, SB = PB / 16, PXA = 64 / SA, PXB = 64 / SB;
  constexpr int HW2 = C::HP, EPC = 8;
  constexpr unsigned OOB = 0x80000000u;  // (host: every tensor below 2 GiB)
  constexpr int NDI = (C::MAXPIX / PXA + 3) / 4, NHI = (C::MAXHALO / PXB + 3) / 4;
  const ConvArgs& a = g.c;
  const int suba = lane / SA, slota = lane % SA, subb = lane / SB, slotb = lane % SB;
  // logical 16-byte channel chunk of this lane's slot (the swizzle term of its pixel is a lane constant: an
  // instruction's first pixel is a multiple of four)
  const int chunka = PA == 128 ? slota ^ (((suba >> 1) & 1) << 2) : slota;
  const int chunk = PB == 128 ? slotb ^ (((subb >> 1) & 1) << 2) : slotb;
  const int ndq = npix_pad / PXA, nhq = ((TH + 2) * HW2 + PXB - 1) / PXB;
  const int cabs = ci0 + chunk * EPC;
  const bool in2 = ci0 >= a.C1;  // (host: C1 % (ci block) == 0 when there is a second source)
  const bool cvalid = cabs < a.C1 + a.C2, covalid = co0 + chunka * EPC < a.Cout;
  const bool up2 = !in2 && a.mode1 == CY_SRC_UP2;
  const int ldx = in2 ? a.ld2 : a.ld1;
  const unsigned xcol = (unsigned)((in2 ? cabs - a.C1 : cabs) * (int)sizeof(T));
  const unsigned ycol = (unsigned)((co0 + chunka * EPC) * (int)sizeof(T));
  const int Hs = up2 ? a.H >> 1 : a.H, Ws = up2 ? a.W >> 1 : a.W;  // source geometry per image
  // item geometry
  int dty[NDI], dtx[NDI], hhr[NHI], hhc[NHI];
#pragma unroll
  for (int i = 0; i < NDI; ++i) {
    const int k = (lw + 4 * i) * PXA + suba;
    if (blk_order) {  // k-step = a 4 x 4 patch (see wg12s_mfma_role): row k of the dy image holds that pixel
      const int cbn = TW >> 2, st = k >> 4, l = k & 15;
      const int rq = st / cbn, cbk = st - rq * cbn;
      dty[i] = 4 * rq + (l >> 2);
      dtx[i] = 4 * cbk + (l & 3);
    } else {
      dty[i] = k / TW;
      dtx[i] = k - dty[i] * TW;
    }
    if (k >= npix) dty[i] = -1;
  }
#pragma unroll
  for (int i = 0; i < NHI; ++i) {
    const int P = (lw + 4 * i) * PXB + subb;
    hhr[i] = P / HW2;
    hhc[i] = P - hhr[i] * HW2;
    if (hhr[i] >= TH + 2 || hhc[i] >= TW + 2) hhr[i] = -1000;
  }
  auto make_rsrc = [&](const void* p, long long bytes) {
    const unsigned long long b = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, (int)bytes, 0x00020000);
  };
  // per segment: extents of dy and of the source this ci block reads
  const int n_a = (g.tiles_h_a * TH) / a.H, n_b = a.N - n_a;
  auto seg_bytes = [&](int n, long px_per_img, int ld, int cols) { return n > 0 ? (((long long)n * px_per_img - 1) * ld + cols) * (long long)sizeof(T) : 0ll; };
  const long pxo = (long)a.H * a.W, pxs = (long)Hs * Ws;
  const int xcols = in2 ? a.C2 : a.C1;
  const void* xa = in2 ? a.src2 : a.src1;
  const void* xb = in2 ? g.src2_b : g.src1_b;
  const long long by_a = seg_bytes(n_a, pxo, g.ldy, a.Cout), by_b = g.dy_b ? seg_bytes(n_b, pxo, g.ldy, a.Cout) : 0;
  const long long bx_a = seg_bytes(n_a, pxs, ldx, xcols), bx_b = xb ? seg_bytes(n_b, pxs, ldx, xcols) : 0;
  const bool pro = a.prologue && !in2 && cvalid;
  float psc[EPC], psh[EPC];
  int coef_seg = -1;
  unsigned hok = 0;  // bit i: halo item i of the tile just requested holds data (else zeros: stays untouched)

  auto request = [&](int tr, unsigned char* sDy, unsigned char* sIn) {
    const int ct = tr % g.tiles_w, rt_all = tr / g.tiles_w;
    const bool second = rt_all >= g.tiles_h_a;
    const int rt = second ? rt_all - g.tiles_h_a : rt_all;
    const int R0 = rt * TH, w0 = ct * TW;
    const int n = R0 / a.H, hh0 = R0 - n * a.H;
    // (descriptors of this tile's segment, rebuilt per tile: four resident ones spill scalar registers)
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(second ? g.dy_b : g.dy, second ? by_b : by_a);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(second ? xb : xa, second ? bx_b : bx_a);
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
      const int q = lw + 4 * i;  // wave-uniform
      if (q >= ndq) continue;
      const int w = w0 + dtx[i];
      const bool ok = dty[i] >= 0 && covalid && w < a.W;
      const unsigned off = ok ? (unsigned)((R0 + dty[i]) * a.W + w) * (unsigned)(g.ldy * (int)sizeof(T)) + ycol : OOB;
      auto* dst = (__attribute__((address_space(3))) void*)(sDy + q * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, dst, 16, off, 0, 0, 0);
    }
    unsigned nhok = 0;
#pragma unroll
    for (int i = 0; i < NHI; ++i) {
      const int q = lw + 4 * i;
      if (q >= nhq) continue;
      const int hh = hh0 - 1 + hhr[i], w = w0 - 1 + hhc[i];
      const bool ok = cvalid && hh >= 0 && hh < a.H && w >= 0 && w < a.W;  // (hhr = -1000: never)
      const int pix = up2 ? (n * Hs + (hh >> 1)) * Ws + (w >> 1) : (n * a.H + hh) * a.W + w;
      const unsigned off = ok ? (unsigned)pix * (unsigned)(ldx * (int)sizeof(T)) + xcol : OOB;
      auto* dst = (__attribute__((address_space(3))) void*)(sIn + q * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, off, 0, 0, 0);
      nhok |= (ok ? 1u : 0u) << i;
    }
    hok = nhok;
    if (pro && (int)second != coef_seg) {  // coefficients of the segment whose data is on its way
      const float* sc = second ? g.scale_b : a.scale;
      const float* sh = second ? g.shift_b : a.shift;
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        psc[j] = sc[cabs + j];
        psh[j] = sh[cabs + j];
      }
      coef_seg = second ? 1 : 0;
    }
  };
  // BN+ReLU prologue over this wave's own halo items, after they have landed (padding stays zero).  All reads first,
  // then the arithmetic, then the writes: item by item the pass was a chain of twelve LDS round trips (stamps: 3 500
  // cycles per tile, which made the loaders the long pole again on the six layers with a prologue).
  auto transform = [&](unsigned char* sIn) {
    if (!pro) return;
    u32x4 v[NHI];
#pragma unroll
    for (int i = 0; i < NHI; ++i) {
      const int q = lw + 4 * i;
      v[i] = q < nhq ? *reinterpret_cast<const u32x4*>(sIn + q * 1024 + lane * 16) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < NHI; ++i) {
      float f[EPC];
      Chunk<T>::unpack(v[i], f);
#pragma unroll
      for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(psc[j], f[j], psh[j]), 0.f);
      v[i] = Chunk<T>::pack(f);
    }
#pragma unroll
    for (int i = 0; i < NHI; ++i) {
      const int q = lw + 4 * i;
      if (q < nhq && ((hok >> i) & 1u)) st16(sIn + q * 1024 + lane * 16, v[i]);
    }
  };

  __syncthreads();  // tables published
  if (nmine > 0) {
    request(split, smem, smem + C::A_BYTES);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    transform(smem + C::A_BYTES);
  }
  __syncthreads();  // first tile staged
#ifdef CY_WGRAD_STAMPS  // phases: 0 = DMA issue, 2 = wait for the data + prologue pass, 3 = barrier
  const bool stamping = g.c.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0;
  unsigned long long ph_sum[4] = {0ull, 0ull, 0ull, 0ull}, t_last = stamping ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
  for (int j = 0; j < nmine; ++j) {
    if (j + 1 < nmine) {
      unsigned char* nb = smem + ((j + 1) & 1) * C::BUF;
      request(split + (j + 1) * g.S, nb, nb + C::A_BYTES);
      W12_STAMP(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      transform(nb + C::A_BYTES);
      W12_STAMP(2);
    }
    __syncthreads();  // tile j consumed by the MFMA waves, tile j + 1 staged
    W12_STAMP(3);
  }
#ifdef CY_WGRAD_STAMPS
  if (stamping && lane == 0) {
    for (int q = 0; q < 4; ++q) g.c.stamps[(8 + lw) * 8 + q] = ph_sum[q];
    g.c.stamps[(8 + lw) * 8 + 4] = (unsigned long long)nmine;
  }
#endif
#endif
}

template <typename T, int WCO, int WCI, bool DMA = false, bool BLK = false>
__global__ void __launch_bounds__(768, 1)
    wgrad12s_kernel(const WgradArgs g) {
  static_assert(!BLK || DMA, "patch order of the k-steps: DMA loaders only");
  using C = Wg12sCfg<WCO, WCI>;
  constexpr int PA = C::PA, PB = C::PB, CPA = C::CPA, CPB = C::CPB;
  constexpr int EPC = 8;
  constexpr int SWA = PA == 128 ? 2 : 0, SWB = PB == 128 ? 2 : 0;  // pair swizzle of 128-byte pixels
  constexpr int HW2 = C::HP;
  constexpr int NL = 256;  // loader threads
  constexpr int NDY = (C::MAXPIX * CPA + NL - 1) / NL;
  constexpr int NHL = (C::MAXITEMS * CPB + NL - 1) / NL;
  static_assert(NL % CPA == 0 && NL % CPB == 0, "a loader thread keeps its channel chunk");
  static_assert(NHL <= 32, "hok mask");
  const ConvArgs& a = g.c;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* s_ktab = reinterpret_cast<int*>(smem + C::MAIN);  // tile pixel k -> halo pixel (ty+1)*HP + tx+1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 8;
  const int lt = tid - 512;  // loader thread index

  const int TH = g.TH, TW = g.TW;
  const int npix = TH * TW;
  const int nsteps = (npix + 15) / 16;
  const int npix_pad = nsteps * 16;
  const int ci_tiles = g.ci_pad / C::BCI;
  const int co_t = blockIdx.x / ci_tiles, ci_t = blockIdx.x % ci_tiles;
  const int co0 = co_t * C::BCO, ci0 = ci_t * C::BCI;
  const int split = blockIdx.y;
  const int ntiles = g.tiles_h * g.tiles_w;
  const int nmine = (ntiles - split + g.S - 1) / g.S;  // tiles split, split + S, ...

  if (tid < 256) {
    const int kk = tid < npix ? tid : 0;
    const int ty = kk / TW, tx = kk - ty * TW;
    s_ktab[tid] = (ty + 1) * HW2 + tx + 1;
  }
  // halo item -> LDS halo pixel (row * HP + column): with the two tables the loaders' per-item index arithmetic has
  // no division by a run-time tile width left (19 items x two such divisions per tile made the four loader waves
  // the slowest part of the first version of this kernel)
  int* s_htab = s_ktab + 256;
  {
    const int HWt_ = TW + 2, nh_ = (TH + 2) * HWt_;
    if (tid >= 256 && tid - 256 < nh_) {
      const int lin = tid - 256;
      const int hr = lin / HWt_, hc = lin - hr * HWt_;
      s_htab[lin] = hr * HW2 + hc;
    }
  }

  if constexpr (DMA) {
    if (loader) {
      __builtin_amdgcn_s_setprio(3);
      wg12s_dma_loader<T, WCO, WCI>(g, smem, wave - 8, lane, TH, TW, npix, npix_pad, co0, ci0, split, nmine, BLK);
      return;
    }
  }
  if (loader) {
    // (the loaders' few hundred instructions per tile go in front of the MFMA waves' on the shared SIMD)
    __builtin_amdgcn_s_setprio(3);
    // ---- staging: request (global -> registers) / commit (registers -> LDS buffer), 256 threads ----------
    const int ld1v = a.ld1, ld2v = a.ld2;
    const int cha = lt % CPA, chb = lt % CPB;
    const int HWt = TW + 2;
    const int nhalo = (TH + 2) * HWt;
    const int cabs = ci0 + chb * EPC;
    const bool in2 = cabs >= a.C1;
    const bool cvalid = cabs < a.C1 + a.C2;
    const bool pooled = a.mode1 == CY_SRC_POOL2 && !in2;  // 2x2 max on load: staged in the commit phase
    const bool pro = a.prologue && !in2 && cvalid;
    float psc[EPC], psh[EPC];
    int coef_seg = -1;
    struct Seg {
      const T* dy;
      const T* s1;
      const T* s2;
      int rt, id;
    };
    auto segment = [&](int rt) {
      Seg sg;
      const bool second = rt >= g.tiles_h_a;
      sg.id = second ? 1 : 0;
      sg.rt = second ? rt - g.tiles_h_a : rt;
      sg.dy = reinterpret_cast<const T*>(second ? g.dy_b : g.dy);
      sg.s1 = reinterpret_cast<const T*>(second ? g.src1_b : a.src1);
      sg.s2 = reinterpret_cast<const T*>(second ? g.src2_b : a.src2);
      return sg;
    };
    auto src_row = [&](int n, int hh) -> int {  // pixel index of (row hh of image n, column 0), or -1
      if (hh < 0 || hh >= a.H) return -1;
      if (in2 || a.mode1 == CY_SRC_DIRECT) return (n * a.H + hh) * a.W;
      if (a.mode1 == CY_SRC_POOL2) return (n * 2 * a.H + 2 * hh) * (2 * a.W);
      return (n * (a.H >> 1) + (hh >> 1)) * (a.W >> 1);
    };
    u32x4 dreg[NDY], hreg[NHL];
    unsigned dok = 0, hok = 0;  // bit i: item i holds loaded data (else it stands for zeros)
    // k / TW and lin / (TW + 2) by multiplication (exact for operands < 512, divisors <= 64): the loaders must not
    // depend on LDS table reads -- the LDS queue is full of the MFMA waves' fragment reads, and a dependent table
    // read per item put ~600 cycles of latency in front of every global load
    const int mtw = (65536 + TW - 1) / TW, mhw = (65536 + HWt - 1) / HWt;
    // One pass over the items: item i of the tile IN the registers goes to LDS (`tc`, if >= 0), then the same
    // register receives item i of tile `tr` (if >= 0).  Commit and request therefore overlap -- the LDS pipe (busy
    // with the MFMA waves' fragment reads) and the address unit work at the same time -- instead of costing their
    // sum (stamps of the first version: commit 5 200 + request 6 000 cycles per tile against an MFMA loop of 6 600).
    // Every load is unconditional (clamped address + validity bit): a conditional one makes the compiler wait for
    // ALL outstanding loads before each store.
    // (the item -> pixel index arithmetic of 19 items is loop invariant; hoisted out of the tile loop it costs more
    //  registers than there are -- an opaque copy of the thread index keeps it inside)
    auto swap = [&](auto DC, auto DR, int tc, unsigned char* sDy, unsigned char* sIn, int tr) {
      int lv = lt;
      asm volatile("" : "+v"(lv));
      constexpr bool do_c = decltype(DC)::value, do_r = decltype(DR)::value;  // (compile time: no branch in the item loops)
      const int trc = do_r ? tr : 0;
      const int ct = trc % g.tiles_w;
      const Seg sg = segment(trc / g.tiles_w);
      const int R0 = sg.rt * TH, w0 = ct * TW;
      const int n = R0 / a.H, hh0 = R0 - n * a.H;
      const int co = co0 + cha * EPC;
      unsigned ndok = 0, nhok = 0;
#pragma unroll
      for (int i = 0; i < NDY; ++i) {
        const int k = (lv + i * NL) / CPA;
        if (do_c && k < npix_pad) {
          const u32x4 v = ((dok >> i) & 1u) ? dreg[i] : u32x4{0u, 0u, 0u, 0u};
          st16(sDy + k * PA + (halo_chunk<PA, SWA>(k, cha) << 4), v);
        }
        if (do_r) {
          const int ty = (k * mtw) >> 16, tx = k - ty * TW;
          const int w = w0 + tx;
          const bool ok = k < npix && co < a.Cout && w < a.W;
          const size_t off = ok ? ((size_t)(R0 + ty) * a.W + w) * g.ldy + co : 0;
          dreg[i] = ld16(sg.dy + off);
          ndok |= (ok ? 1u : 0u) << i;
        }
      }
      if (pooled) {
        if (do_c) {  // 2x2 max on load: synchronous
          const int ctc = tc % g.tiles_w;
          const Seg sc = segment(tc / g.tiles_w);
          const int R0c = sc.rt * TH, w0c = ctc * TW;
          const int nc = R0c / a.H, hh0c = R0c - nc * a.H;
          for (int lin = lt / CPB; lin < nhalo; lin += NL / CPB) {
            const int pix = s_htab[lin];
            const int hr = pix / HW2, hc = pix - hr * HW2;
            const int w = w0c - 1 + hc;
            u32x4 v = {0u, 0u, 0u, 0u};
            const int rp = src_row(nc, hh0c - 1 + hr);
            if (cvalid && w >= 0 && w < a.W && rp >= 0) {
              const T* p = sc.s1 + (size_t)(rp + 2 * w) * a.ld1 + cabs;
              const size_t rowstep = (size_t)(2 * a.W) * a.ld1;
              const u32x4 v00 = ld16(p), v01 = ld16(p + a.ld1), v10 = ld16(p + rowstep),
                          v11 = ld16(p + rowstep + a.ld1);
              float f0[EPC], f1[EPC], f2[EPC], f3[EPC];
              Chunk<T>::unpack(v00, f0);
              Chunk<T>::unpack(v01, f1);
              Chunk<T>::unpack(v10, f2);
              Chunk<T>::unpack(v11, f3);
#pragma unroll
              for (int j = 0; j < EPC; ++j) f0[j] = fmaxf(fmaxf(f0[j], f1[j]), fmaxf(f2[j], f3[j]));
              v = Chunk<T>::pack(f0);
            }
            st16(sIn + pix * PB + (halo_chunk<PB, SWB>(pix, chb) << 4), v);
          }
        }
      } else {
        const T* base = in2 ? sg.s2 + (cabs - a.C1) : sg.s1 + cabs;
        const int ld = in2 ? ld2v : ld1v;
        const int wsh = (!in2 && a.mode1 == CY_SRC_UP2) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < NHL; ++i) {
          const int lin = (lv + i * NL) / CPB;
          const int hr = (lin * mhw) >> 16, hc = lin - hr * HWt;
          const int pix = hr * HW2 + hc;
          if (do_c && lin < nhalo) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if ((hok >> i) & 1u) {
              v = hreg[i];
              if (pro) {
                float f[EPC];
                Chunk<T>::unpack(v, f);
#pragma unroll
                for (int j = 0; j < EPC; ++j) f[j] = fmaxf(fmaf(psc[j], f[j], psh[j]), 0.f);
                v = Chunk<T>::pack(f);
              }
            }
            st16(sIn + pix * PB + (halo_chunk<PB, SWB>(pix, chb) << 4), v);
          }
          if (do_r) {
            const int w = w0 - 1 + hc;
            const int rp = src_row(n, hh0 - 1 + hr);
            const bool ok = lin < nhalo && cvalid && w >= 0 && w < a.W && rp >= 0;
            const size_t off = ok ? (size_t)(rp + (w >> wsh)) * ld : 0;
            hreg[i] = ld16((ok ? base : sg.s1) + off);
            nhok |= (ok ? 1u : 0u) << i;
          }
        }
      }
      if (do_r) {
        dok = ndok, hok = nhok;
        if (pro && sg.id != coef_seg) {  // coefficients of the segment whose data the registers now hold
          const float* sc = sg.id ? g.scale_b : a.scale;
          const float* sh = sg.id ? g.shift_b : a.shift;
#pragma unroll
          for (int j = 0; j < EPC; ++j) {
            psc[j] = sc[cabs + j];
            psh[j] = sh[cabs + j];
          }
          coef_seg = sg.id;
        }
      }
    };

    using Yes = Wg12sFlag<true>;
    using No = Wg12sFlag<false>;
    __syncthreads();  // tables published
    swap(No{}, Yes{}, -1, smem, smem + C::A_BYTES, split);
    if (nmine > 1)
      swap(Yes{}, Yes{}, split, smem, smem + C::A_BYTES, split + g.S);
    else
      swap(Yes{}, No{}, split, smem, smem + C::A_BYTES, -1);
    __syncthreads();  // first tile staged
#ifdef CY_WGRAD_STAMPS
    const bool stamping = g.c.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0;
    unsigned long long ph_sum[4] = {0ull, 0ull, 0ull, 0ull}, t_last = stamping ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
    int j = 0;
    for (; j + 2 < nmine; ++j) {  // steady state: straight-line item loops
      unsigned char* nb = smem + ((j + 1) & 1) * C::BUF;
      swap(Yes{}, Yes{}, split + (j + 1) * g.S, nb, nb + C::A_BYTES, split + (j + 2) * g.S);
      W12_STAMP(0);
      __syncthreads();  // tile j consumed by the MFMA waves, tile j+1 staged
      W12_STAMP(3);
    }
    for (; j < nmine; ++j) {
      if (j + 1 < nmine) {
        unsigned char* nb = smem + ((j + 1) & 1) * C::BUF;
        swap(Yes{}, No{}, split + (j + 1) * g.S, nb, nb + C::A_BYTES, -1);
      }
      W12_STAMP(0);
      __syncthreads();
      W12_STAMP(3);
    }
#ifdef CY_WGRAD_STAMPS
    if (stamping && lane == 0) {
      for (int q = 0; q < 4; ++q) g.c.stamps[wave * 8 + q] = ph_sum[q];
      g.c.stamps[wave * 8 + 4] = (unsigned long long)nmine;
    }
#endif
    return;
  }

  // ---- MFMA waves: tap group = wave >> 2 (one of each per SIMD), block and pixel split from wave & 3 ----
  const int sub = wave & 3;
  if (wave < 4)
    wg12s_mfma_role<T, WCO, WCI, 0, 5, BLK>(g, smem, s_ktab, sub % C::NBLK, sub / C::NBLK, lane, nsteps, nmine, co0, ci0,
                                            split);
  else
    wg12s_mfma_role<T, WCO, WCI, 5, 4, BLK>(g, smem, s_ktab, sub % C::NBLK, sub / C::NBLK, lane, nsteps, nmine, co0, ci0,
                                            split);
}

template <typename T, int WCO, int WCI, bool DMA = false, bool BLK = false>
int launch_wgrad12s(const WgradArgs& g, const WgPlan& p, hipStream_t st) {
  using C = Wg12sCfg<WCO, WCI>;
  auto kern = wgrad12s_kernel<T, WCO, WCI, DMA, BLK>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::SMEM) != hipSuccess)
      return CY_ERR_LAUNCH;
    attr_done = true;
  }
  dim3 grid((p.co_pad / C::BCO) * (p.ci_pad / C::BCI), p.S);
  WgradArgs ga = g;
  ga.c.stamps = g_w12_stamp_buf;
  hipLaunchKernelGGL(kern, grid, dim3(768), C::SMEM, st, ga);
  CY_CHECK_LAUNCH();
  return CY_OK;
}

template <typename T>
int dispatch_wgrad12s(const WgradArgs& g, const WgPlan& p, hipStream_t st) {
  if (p.wco == 2 && p.wci == 2) {
    if (p.blk_order) return launch_wgrad12s<T, 2, 2, true, true>(g, p, st);
    return p.dma ? launch_wgrad12s<T, 2, 2, true>(g, p, st) : launch_wgrad12s<T, 2, 2>(g, p, st);
  }
  // 32-channel blocks: only with DMA loaders and patch-ordered k-steps (plan_wgrad)
  if (p.blk_order && p.wco == 1 && p.wci == 2) return launch_wgrad12s<T, 1, 2, true, true>(g, p, st);
  if (p.blk_order && p.wco == 1 && p.wci == 1) return launch_wgrad12s<T, 1, 1, true, true>(g, p, st);
  return CY_ERR_SHAPE;  // (<2,1>, <1,2>, <1,1> compile and pass the parity tests but are slower than wgrad12_kernel: not built)
}
