// mlx_ro_dense.h -- REFERENCE-ORDER numerics (MLX_NUMERICS_REFERENCE_ORDER) for DENSE tiles; included by mlx_kernels.hip.
//
// Round 5 ran a dense tile through the CSR kernels of this mode, entry by entry. A tile needs none of that machinery: its chains are
// short, all of the same length and there are a million of them, so the reference's sequential loops run as they are, one chain per
// LANE, and the two kernels below are plain streaming reads of the tile -- two per tick where the fast contract's fused pass reads it
// once, because XTv's chain over the rows (llf/LogisticRegressionL2.java:131-150) needs every row's coefficient in row order, and a
// row's coefficient needs the whole row first (Xv, :115-129).
//   * k_ro_dense_rows: Xv. One lane per ROW, one running sum over the row's columns in ascending id, the bias entry last. A wave
//     owns 64 rows and stages [64 rows x 64 columns] of the fp32 tile at a time through LDS (coalesced 256-byte pieces of a row in,
//     one row per lane out, 16-byte slots rotated by the row so that both sides are free of bank conflicts); the gathered vector's 64
//     entries ride in one register per lane and reach the chain as scalars (v_readlane). The next tile's loads are in flight while
//     the chain runs. Row map = k_rowpass_lds<.., RO>'s (portable exp / log1p, the row's loss left in rowtmp[]).
//   * k_ro_dense_cols: XTv. One lane per COLUMN, one wave per strip of 64 columns walking ALL rows in order: acc += coef[i] * x[i][j]
//     (256-byte coalesced loads, 64 rows in flight per wave; the row coefficients of a 64-row batch ride in one register per lane).
//     Two more chains sit on spare lanes behind the last data column: the intercept's column (value 1.0: the sum of the coefficients
//     in row order) and, on EVAL ticks, the loss sum `f += weight*log(1+exp(..))` in row order (:172-183) -- so the step kernel's
//     l-long folds disappear for dense problems (k_ro_step reads csump[0] / lossp[0]).
// A zero entry of the tile adds +-0.0 to a running sum, which leaves every bit of it (the sums start at +0.0 and round to nearest:
// they never hold -0.0), so the result equals the entry-by-entry sums over the non-zeros the reference would store.
#pragma once

typedef float rod_f4 __attribute__((ext_vector_type(4)));

// ---- Xv: one lane per row --------------------------------------------------------------------------------------------------------
template <bool NT>
__global__ void __launch_bounds__(256)
k_ro_dense_rows(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char rod_smem[];      // [4 waves][64 rows][16 slots of 16 bytes]
    ProbDev &pr = probs[qlist[blockIdx.y]];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    const int l = pa.l, nf = pa.n_feat;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // (a scalar: the buffer descriptor below must be wave-uniform)
    const int row0 = (blockIdx.x * 4 + wave) * 64;
    if (row0 >= l) return;                                   // (wave-uniform; the waves of a workgroup share nothing: no barriers)
    const int64_t ld = pa.ld;
    const bool cg = (phase == PH_CG);
    const double *__restrict__ v = cg ? pr.d : pr.w_new;
    unsigned char *tile = rod_smem + wave * (64 * 256);
    // loads: instruction i of a tile fetches rows 4 i .. 4 i + 3, 16 lanes (256 bytes) per row. Buffer loads: one descriptor for the
    // wave's 64 rows, a lane offset that never changes and a SCALAR offset per instruction (no vector address arithmetic); the
    // descriptor's range ends with the partition's last row, so rows behind it read 0.0 (their sums are never stored) -- no clamps.
    const int lr = lane >> 4, lq = lane & 15;
    const int ntiles = (nf + 63) >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(pa.X + (int64_t)row0 * ld), 0, (int)((int64_t)min(64, l - row0) * ld * 4), 0x00020000);
    const int voff = (int)(((int64_t)lr * ld + 4 * lq) * 4);
    const int ld16 = (int)(ld * 16);                         // bytes of four rows
    rod_f4 xr[16];
    auto fetch = [&](int t) {
        const int k0b = min(t, ntiles - 1) << 8;             // byte offset of the tile's first column (unconditional: the last trip re-reads)
#pragma unroll
        for (int i = 0; i < 16; i++)
            xr[i] = __builtin_bit_cast(rod_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k0b + i * ld16, NT ? 2 : 0));
    };
    auto vchunk = [&](int t) { return gld(v + min((min(t, ntiles - 1) << 6) + lane, nf)); };      // (clamped: entries behind column nf - 1 are never used)
    fetch(0);
    double vn = vchunk(0);
    double acc = 0.0;
    for (int t = 0; t < ntiles; t++) {
        // registers -> LDS: row r's 16-byte slot q sits at position (q + r) mod 16 of the row's 256 bytes
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int r = 4 * i + lr;
            *reinterpret_cast<rod_f4 *>(tile + r * 256 + (((lq + r) & 15) << 4)) = xr[i];
        }
        const double vc = vn;
        fetch(t + 1);
        vn = vchunk(t + 1);
        const int k0 = t << 6;
        if (k0 + 64 <= nf) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const rod_f4 x4 = *reinterpret_cast<const rod_f4 *>(tile + lane * 256 + (((q + lane) & 15) << 4));
                acc = acc + (double)x4.x * mlx_wave_bcast(vc, 4 * q);            // Xv[i] += v[s.index - 1] * s.value (:124-126)
                acc = acc + (double)x4.y * mlx_wave_bcast(vc, 4 * q + 1);
                acc = acc + (double)x4.z * mlx_wave_bcast(vc, 4 * q + 2);
                acc = acc + (double)x4.w * mlx_wave_bcast(vc, 4 * q + 3);
            }
        } else {
            const int rem = nf - k0;                         // the last, partial tile (wave-uniform bounds)
            for (int q = 0; 4 * q < rem; q++) {
                const rod_f4 x4 = *reinterpret_cast<const rod_f4 *>(tile + lane * 256 + (((q + lane) & 15) << 4));
                const float xe[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (4 * q + e < rem) acc = acc + (double)xe[e] * mlx_wave_bcast(vc, 4 * q + e);
            }
        }
    }
    // the bias entry closes the row, then the row map (k_rowpass_lds<.., RO>'s)
    const int row = row0 + lane;
    if (row < l) {
        const double t = acc + gld(v + nf);
        double cfv;
        if (cg) {
            cfv = gld_nt(pr.wd[pr.dsel] + row) * t;                              // wa[i] = weight * D * (X s)[i] (:243)
        } else {
            double lossv, wdv;
            row_eval<true>(t + (double)gld(pa.off + row), (int)gld(pa.y + row), (double)gld(pa.wt + row), lossv, wdv, cfv);
            gst(pr.wd[pr.dsel ^ 1] + row, wdv);
            gst(pr.rowtmp + row, lossv);
        }
        gst(pr.coef + row, cfv);
    }
}

// ---- XTv: one lane per column, all rows in order ----------------------------------------------------------------------------------
// R waves RELAY one strip of 64 columns: a column's sum is one chain of dependent adds over all rows, so a strip cannot be split
// between waves -- but only the ADDS are sequential. Wave w takes the BR-row batches w, w + R, ...: its loads are in flight long before
// its turn, it forms the batch's TERMS c_i * x_ij (conversion, product, the scalar broadcasts of c_i: 4 of the 5 instructions per
// entry) while the other waves have their turns, and on its turn (relay_turn == b) it adds the BR terms to the 64 running sums it takes
// from LDS and hands them on. One wave alone keeps at most 63 loads = 16 KB of a strip in flight (the vmcnt counter), a CU 63 KB --
// the first form of this kernel was bound by memory latency at 0.45-0.54 of the HBM peak, the second (two waves, whole batches
// computed on the turn) by the 450 us its chain took even on an idle chip (profiles/r6_notes.md).
// VIRT: the strip holds the chains behind the last data column: the intercept's column (multiplicand 1.0) and, on EVAL ticks, the loss
// sum (rowtmp[i] for the coefficient, multiplicand 1.0).
template <bool VIRT, int BR>
__device__ __forceinline__ void rod_col_terms(double (&t)[BR], const float (&x)[BR], double cchunk, double lchunk, bool data_lane, float virt_x, bool loss_lane)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int r = 0; r < BR; r++) {
        double c = mlx_wave_bcast(cchunk, r);
        float xv = x[r];
        if (VIRT) {
            const double lv = mlx_wave_bcast(lchunk, r);
            c = loss_lane ? lv : c;
            xv = data_lane ? xv : virt_x;
        }
        t[r] = c * (double)xv;                               // v[i] * s.value (:143-145)
    }
}

template <bool NT, int R, int BR>
__device__ __forceinline__ void rod_cols_strip(const PartDev &pa, ProbDev &pr, int strip, int wave, int lane, volatile double *run, volatile int *turn)
{
#pragma clang fp contract(off)
    const int l = pa.l, nf = pa.n_feat;
    const bool ev = (pr.phase != PH_CG);
    const int nvc = nf + (ev ? 2 : 1);                       // data columns, the intercept's column, (EVAL) the loss sum
    const int jw0 = strip * 64;
    if (jw0 >= nvc) return;                                  // (uniform over the workgroup)
    const int j = jw0 + lane;
    const bool virt = jw0 + 64 > nf;                         // this strip also holds the chains behind the last data column
    const bool data_lane = j < nf, loss_lane = ev && j == nf + 1;
    const float virt_x = (j == nf || loss_lane) ? 1.0f : 0.0f;
    const int64_t ld = pa.ld;
    // A row's 64 entries = one buffer load: the descriptor starts at the batch's first row and the strip's first column, the lane offset
    // never changes, the row is a SCALAR offset (no vector address arithmetic). The descriptor's range ends with the batch's last
    // row (the partition's last row in the last batch): rows behind it read 0.0 -- no clamps.
    const float *__restrict__ xw = pa.X + jw0;
    const int ld4 = (int)(ld * 4);
    const double *__restrict__ coef = pr.coef, *__restrict__ rowtmp = pr.rowtmp;
    const int nb = (l + BR - 1) / BR;
    float x[BR];
    double t[BR];
    double cc, lc;
    auto fetch = [&](int b) {
        const int i0 = min(b, nb - 1) * BR;                  // (unconditional: a trip behind the last batch re-reads it)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xw + (int64_t)i0 * ld), 0, (int)(((int64_t)min(BR, l - i0) * ld - jw0) * 4), 0x00020000);
#pragma unroll
        for (int r = 0; r < BR; r++)
            x[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, r * ld4, NT ? 2 : 0));
        const int i = i0 + (lane & (BR - 1));
        cc = gld(coef + min(i, l - 1));
        cc = i < l ? cc : 0.0;                               // rows behind the last: coefficient 0.0 (acc + 0.0 * 0.0 keeps acc)
        lc = gld(rowtmp + min(i, l - 1));                    // (unconditional: a load under a branch makes the whole batch wait; only the loss lane of an EVAL tick uses it)
        lc = i < l ? lc : 0.0;
    };
    if (wave == 0) { run[lane] = 0.0; if (lane == 0) *turn = 0; }
    __syncthreads();
    if (wave < nb) {
        fetch(wave);
        for (int b = wave; b < nb; b += R) {
            if (virt) rod_col_terms<true, BR>(t, x, cc, lc, data_lane, virt_x, loss_lane);
            else rod_col_terms<false, BR>(t, x, cc, lc, true, 0.f, false);
            fetch(b + R);                                    // the registers of x[] are free again: the next batch's loads fly during the turn
            // my turn: the 64 running sums through batch b - 1 (LDS accesses of a CU are served in order: whoever sees relay_turn == b
            // also sees the sums written before it)
            while (*turn != b) { }
            double acc = run[lane];
#pragma unroll
            for (int r = 0; r < BR; r++) acc = acc + t[r];   // XTv[s.index - 1] += v[i] * s.value, rows ascending
            run[lane] = acc;
            if (lane == 0) *turn = b + 1;
        }
    }
    __syncthreads();
    if (wave == 0) {
        const double acc = run[lane];
        if (data_lane) gst(pr.c0f + j, acc);                 // xtc: X'c, columns 0 .. nf-1
        else if (j == nf) pr.csump[0] = acc;                 // the intercept's column: XTv[n-1] += v[i] * 1.0
        else if (loss_lane) pr.lossp[0] = acc;               // sum of the rows' losses in row order
    }
}

// Every workgroup of this kernel lives as long as its strip takes -- the whole launch -- so WHERE the dispatcher puts the workgroups
// decides the launch time, and the dispatcher fills a CU with as many as fit before it moves on. The grid is therefore the same for
// every launch (strips x problems of the list, spread evenly by an LDS allocation nobody touches: mlxk_ro_dense_passes) and a
// workgroup CLAIMS its (unfinished problem, strip) in the order the workgroups start: with half the problems done the first half of
// the workgroups -- two per CU, not four on half of the CUs -- do the work and the rest leave. claim[0] = next item, claim[1] =
// workgroups through; the last one through clears both for the next launch.
template <bool NT, int R, int BR>
__global__ void __launch_bounds__(64 * R)
k_ro_dense_cols(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq, int strips, int *__restrict__ claim)
{
    __shared__ double relay_run[64];
    __shared__ int relay_turn;
    __shared__ int item[2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // (a scalar: the buffer descriptors must be wave-uniform)
    if (wave == 0) {
        int idx = 0;
        if (lane == 0) idx = atomicAdd(&claim[0], 1);
        idx = __builtin_amdgcn_readfirstlane(idx);
        const int a = idx / strips;                          // the a-th unfinished problem of the list
        int q = -1, cnt = 0;
        for (int base = 0; base < nq && q < 0; base += 64) {
            const int i = base + lane;
            const bool act = i < nq && gld(&probs[gld(qlist + min(i, nq - 1))].phase) != PH_DONE;
            const unsigned long long m = __ballot(act);
            const int c = __popcll(m);
            if (a < cnt + c) {
                const int rank = __popcll(m & ((1ull << lane) - 1ull));
                q = base + __ffsll((unsigned long long)__ballot(act && rank == a - cnt)) - 1;
            }
            cnt += c;
        }
        if (lane == 0) { item[0] = q; item[1] = idx % strips; }
    }
    __syncthreads();
    const int qi = __builtin_amdgcn_readfirstlane(item[0]), strip = __builtin_amdgcn_readfirstlane(item[1]);
    if (qi >= 0) {
        ProbDev &pr = probs[qlist[qi]];
        rod_cols_strip<NT, R, BR>(parts[pr.part], pr, strip, wave, lane, relay_run, &relay_turn);
    }
    if (threadIdx.x == 0 && atomicAdd(&claim[1], 1) == (int)(gridDim.x * gridDim.y) - 1) { claim[0] = 0; claim[1] = 0; }
}

void mlxk_ro_dense_passes(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int max_l, int max_nfeat,
                          bool stream_once, int which, int *claim)
{
    if (nq <= 0) return;
    // A/B knobs (tools/r6): MLX_ROD_NT = 0 / 1 forces plain / streaming loads of the tile; MLX_ROD_CWG = waves relaying one strip of the column kernel (1, 2: 64-row batches; 22, 3, 4: 32-row batches)
    static const int nt_env = getenv("MLX_ROD_NT") ? atoi(getenv("MLX_ROD_NT")) : -1;
    static const int cwg = getenv("MLX_ROD_CWG") ? atoi(getenv("MLX_ROD_CWG")) : 4;
    const bool nt = nt_env < 0 ? stream_once : nt_env != 0;
    const dim3 grow((unsigned)((max_l + 255) / 256), (unsigned)nq);
    const size_t lds = 4 * 64 * 256;
    per_device_once(2, [&] {
        set_max_lds(reinterpret_cast<const void *>(&k_ro_dense_rows<true>), (int)lds);
        set_max_lds(reinterpret_cast<const void *>(&k_ro_dense_rows<false>), (int)lds);
    });
    if (which & 1) {
        if (nt) hipLaunchKernelGGL(k_ro_dense_rows<true>, grow, dim3(256), lds, st, parts, probs, qlist);
        else hipLaunchKernelGGL(k_ro_dense_rows<false>, grow, dim3(256), lds, st, parts, probs, qlist);
    }
    if (which & 2) {
        const dim3 gcol((unsigned)((max_nfeat + 2 + 63) / 64), (unsigned)nq);
        // (the same grid and the same LDS allocation for every launch: k_ro_dense_cols; MLX_ROD_WGPC forces the workgroups per CU, 0 = no allocation)
        static const int wgpc_env = getenv("MLX_ROD_WGPC") ? atoi(getenv("MLX_ROD_WGPC")) : -1;
        const int strips = (int)gcol.x;
        const int wgpc = wgpc_env >= 0 ? wgpc_env : ((int)(gcol.x * gcol.y) + 255) / 256;
        const size_t ballast = (wgpc >= 1 && wgpc <= 16) ? (size_t)(160 * 1024 / wgpc - 1024) : 0;
        if (ballast > 64 * 1024)
            per_device_once(4, [&] {
#define SETB(RW, B) set_max_lds(reinterpret_cast<const void *>(&k_ro_dense_cols<true, RW, B>), 159 * 1024); set_max_lds(reinterpret_cast<const void *>(&k_ro_dense_cols<false, RW, B>), 159 * 1024)
                SETB(1, 64); SETB(2, 64); SETB(2, 32); SETB(3, 32); SETB(4, 32);
#undef SETB
            });
#define LAUNCH_COLS(RW, B)                                                                                                               \
        do {                                                                                                                             \
            if (nt) hipLaunchKernelGGL((k_ro_dense_cols<true, RW, B>), gcol, dim3(64 * RW), ballast, st, parts, probs, qlist, nq, strips, claim);   \
            else hipLaunchKernelGGL((k_ro_dense_cols<false, RW, B>), gcol, dim3(64 * RW), ballast, st, parts, probs, qlist, nq, strips, claim);    \
        } while (0)
        switch (cwg) {
        case 1: LAUNCH_COLS(1, 64); break;
        case 2: LAUNCH_COLS(2, 64); break;
        case 22: LAUNCH_COLS(2, 32); break;
        case 3: LAUNCH_COLS(3, 32); break;
        default: LAUNCH_COLS(4, 32); break;
        }
#undef LAUNCH_COLS
    }
}
