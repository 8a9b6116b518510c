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
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * 64;
    if (row0 >= l) return;                                   // (wave-uniform; the waves of a workgroup share nothing: no barriers)
    const int64_t ld = pa.ld;
    const bool cg = (phase == PH_CG);
    const double *__restrict__ v = cg ? pr.d : pr.w_new;
    const float *__restrict__ X = pa.X;
    unsigned char *tile = rod_smem + wave * (64 * 256);
    // loads: instruction i of a tile fetches rows 4 i .. 4 i + 3, 16 lanes (256 bytes) per row
    const int lr = lane >> 4, lq = lane & 15;
    const int ntiles = (nf + 63) >> 6;
    rod_f4 xr[16];
    auto fetch = [&](int t) {
        const int k0 = min(t, ntiles - 1) << 6;              // (unconditional, clamped: a predicated load serialises the batch)
        const int col = min(k0 + 4 * lq, (int)ld - 4);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int row = min(row0 + 4 * i + lr, l - 1);
            const rod_f4 *p = reinterpret_cast<const rod_f4 *>(X + (int64_t)row * ld + col);
            xr[i] = NT ? gld_nt(p) : gld(p);
        }
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
// 64 rows of one batch: acc += c_i * x_i for i = 0 .. 63. VIRT: the wave holds the chains behind the last data column (their
// multiplicand is 1.0; the loss chain takes rowtmp[i] for the coefficient).
template <bool VIRT>
__device__ __forceinline__ double rod_col_batch(double acc, const float (&x)[64], double cchunk, double lchunk, bool data_lane, float virt_x, bool loss_lane)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int r = 0; r < 64; r++) {
        double c = mlx_wave_bcast(cchunk, r);
        float xv = x[r];
        if (VIRT) {
            const double lv = mlx_wave_bcast(lchunk, r);
            c = loss_lane ? lv : c;
            xv = data_lane ? xv : virt_x;
        }
        acc = acc + c * (double)xv;                          // XTv[s.index - 1] += v[i] * s.value (:143-145)
    }
    return acc;
}

template <bool NT>
__global__ void __launch_bounds__(256)
k_ro_dense_cols(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist)
{
#pragma clang fp contract(off)
    ProbDev &pr = probs[qlist[blockIdx.y]];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    const int l = pa.l, nf = pa.n_feat;
    const bool ev = (phase != PH_CG);
    const int nvc = nf + (ev ? 2 : 1);                       // data columns, the intercept's column, (EVAL) the loss sum
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int jw0 = (blockIdx.x * 4 + wave) * 64;
    if (jw0 >= nvc) return;                                  // (wave-uniform; no barriers)
    const int j = jw0 + lane;
    const bool virt = jw0 + 64 > nf;                         // this wave also holds the chains behind the last data column
    const bool data_lane = j < nf, loss_lane = ev && j == nf + 1;
    const float virt_x = (j == nf || loss_lane) ? 1.0f : 0.0f;
    const int64_t ld = pa.ld;
    const float *__restrict__ xcol = pa.X + min(j, (int)ld - 1);
    const double *__restrict__ coef = pr.coef, *__restrict__ rowtmp = pr.rowtmp;
    const int nb = (l + 63) >> 6;
    float xa[64], xb[64];
    double ca, cb, la = 0.0, lb = 0.0;
    auto fetch = [&](int b, float (&x)[64], double &cc, double &lc) {
        const int i0 = min(b, nb - 1) << 6;                  // (unconditional, clamped)
#pragma unroll
        for (int r = 0; r < 64; r++) {
            const float *p = xcol + (int64_t)min(i0 + r, l - 1) * ld;
            x[r] = NT ? gld_nt(p) : gld(p);
        }
        const int i = i0 + lane;
        cc = i < l ? gld(coef + i) : 0.0;                    // rows behind the last: coefficient 0.0 (acc + 0.0 * x keeps acc)
        if (virt && ev) lc = i < l ? gld(rowtmp + i) : 0.0;
    };
    double acc = 0.0;
    fetch(0, xa, ca, la);
    for (int b = 0; b < nb; b += 2) {
        fetch(b + 1, xb, cb, lb);
        acc = virt ? rod_col_batch<true>(acc, xa, ca, la, data_lane, virt_x, loss_lane) : rod_col_batch<false>(acc, xa, ca, la, true, 0.f, false);
        if (b + 1 >= nb) break;
        fetch(b + 2, xa, ca, la);
        acc = virt ? rod_col_batch<true>(acc, xb, cb, lb, data_lane, virt_x, loss_lane) : rod_col_batch<false>(acc, xb, cb, lb, true, 0.f, false);
    }
    if (data_lane) gst(pr.c0f + j, acc);                     // xtc: X'c, columns 0 .. nf-1
    else if (j == nf) pr.csump[0] = acc;                     // the intercept's column: XTv[n-1] += v[i] * 1.0
    else if (loss_lane) pr.lossp[0] = acc;                   // sum of the rows' losses in row order
}

void mlxk_ro_dense_passes(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int max_l, int max_nfeat,
                                 bool stream_once, int which)
{
    if (nq <= 0) return;
    const dim3 grow((unsigned)((max_l + 255) / 256), (unsigned)nq), gcol((unsigned)((max_nfeat + 2 + 255) / 256), (unsigned)nq);
    const size_t lds = 4 * 64 * 256;
    per_device_once(2, [&] {
        set_max_lds(reinterpret_cast<const void *>(&k_ro_dense_rows<true>), (int)lds);
        set_max_lds(reinterpret_cast<const void *>(&k_ro_dense_rows<false>), (int)lds);
    });
    if (which & 1) {
        if (stream_once) hipLaunchKernelGGL(k_ro_dense_rows<true>, grow, dim3(256), lds, st, parts, probs, qlist);
        else hipLaunchKernelGGL(k_ro_dense_rows<false>, grow, dim3(256), lds, st, parts, probs, qlist);
    }
    if (which & 2) {
        if (stream_once) hipLaunchKernelGGL(k_ro_dense_cols<true>, gcol, dim3(256), 0, st, parts, probs, qlist);
        else hipLaunchKernelGGL(k_ro_dense_cols<false>, gcol, dim3(256), 0, st, parts, probs, qlist);
    }
}
