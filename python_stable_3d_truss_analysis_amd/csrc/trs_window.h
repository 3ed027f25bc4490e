// trs_potrf_window_kernel - included by potrf.hip inside its anonymous namespace (Slab, tile_load, tile_store,
// chol16_invert, wfrag_lane).
//
// ONE WORK-GROUP per matrix, right-looking, 16 columns per step, and the whole active window of the
// factorisation in LDS: every stiffness tile is read from HBM once (when its row enters the window), every
// factor tile written once - no factor row is ever read back (the wave-per-matrix kernel re-reads each one
// ~1.7 x on cube trusses: EXPERIMENTS R4.1).
//
// Window.  At step t (pivot column tile t) the live tiles are (q, s), t <= s <= q < cend[t]: a triangle of side
// F_t = cend[t] - t <= M (M = the kernel's capacity, the matrix's `front` = max F_t <= M).  Tile (q, s) lives in
// the slot of the UNORDERED pair of residues {q mod M, s mod M}: inside any M consecutive chunks two different
// lower tiles never share such a pair, so M (M + 1) / 2 slots of 2 KB hold the window with a static formula,
// and the rows that enter at step t + 1 take cells of column t (dead by then) and of nothing else.
//
// Step t (two work-group barriers):
//   A  solve   X(q, t) = T(q, t) inv(L_tt)^T for the column's tiles (one per wave), in place in LDS, out to HBM,
//              and the load vector f_q -= X(q, t) y_t; the rows that ENTER at this step arrive here from the
//              registers they were prefetched into during step t - 1 (pivot-column tiles are solved straight
//              from the registers, the others go into their slots).
//   B  update  T(q, s) -= X(q, t) X(s, t)^T, t < s <= q: one job per tile over waves 1 .. NW-1, while
//      look-ahead: wave 0 updates the NEXT diagonal tile first, factors it (chol16_invert), leaves inv(L)
//              as operand fragments (double-buffered), stores U and forms y_{t+1} - the serial 16 x 16
//              factorisation hides behind the other waves' updates.
#ifndef TRS_WINDOW_WAVES
#define TRS_WINDOW_WAVES 16
#endif
#ifndef TRS_WINDOW_PREFETCH
#define TRS_WINDOW_PREFETCH 3
#endif
constexpr int WINW = TRS_WINDOW_WAVES;     // waves per work-group
constexpr int WPF = TRS_WINDOW_PREFETCH;   // entering tiles a wave holds in registers
constexpr int WIN_JOBS = TRS_WINDOW_MAX_FRONT * (TRS_WINDOW_MAX_FRONT + 1) / 2;

struct WinHead {
    ChScratch ch;
    double W[2][256];   // inv(L_tt) as A-fragments (PanelLds::W layout), by parity of t
    int info;
    unsigned char tri_i[WIN_JOBS], tri_j[WIN_JOBS];   // job k of the update = tile (t + 1 + tri_i[k], t + 1 + tri_j[k])
};

__host__ __device__ static inline size_t trs_window_lds_bytes(int M, int n_pad_max) {
    return ((size_t)M * (M + 1) / 2 * 256 + n_pad_max) * sizeof(double) + (size_t)2 * (n_pad_max / 16) * sizeof(int);
}

__global__ __launch_bounds__(64 * WINW) void trs_potrf_window_kernel(
    double* __restrict__ S_all, const int* __restrict__ n_free, const int ld, const size_t slab_stride,
    int* __restrict__ info, const int* __restrict__ env_all, const int n_pad_max, const int B,
    double* __restrict__ uf_all, const int ld_uf, const int M, const int front_above) {
    extern __shared__ double wdyn[];
    __shared__ WinHead hd;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 15, lq = lane >> 4;
    const int b = (int)gridDim.x - 1 - (int)blockIdx.x;   // descending, like the wave-per-matrix kernel
    if (b >= B) return;
    const int npad = trs_round_up(n_free[b], TRS_NB);
    if (npad == 0) return;   // (the wave-per-matrix launch reports info = 0 for an empty system)
    const TrsEnv env = trs_env_of(env_all, b, n_pad_max);
    if (!trs_env_is_window(env)) return;
    const int front = (env.last + n_pad_max / 64)[5];
    if (front <= front_above || front > M) return;   // another capacity class
    const int nch = npad / 16;
    double* const tiles = wdyn;
    double* const f = wdyn + (size_t)(M * (M + 1) / 2) * 256;
    int* const cendl = reinterpret_cast<int*>(f + n_pad_max);
    int* const kml = cendl + n_pad_max / 16;
    double* const ufb = uf_all + (size_t)b * ld_uf;
    for (int i = tid; i < nch; i += 64 * WINW) {
        cendl[i] = env.cend[i];
        kml[i] = env.kmask[i];
    }
    for (int i = tid; i < npad; i += 64 * WINW) f[i] = ufb[i];
    if (tid < WIN_JOBS) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= tid) ++i;
        hd.tri_i[tid] = (unsigned char)i;
        hd.tri_j[tid] = (unsigned char)(tid - i * (i + 1) / 2);
    }
    if (tid == 0) hd.info = 0;
    __syncthreads();

    Slab S;
    S.rs = __builtin_amdgcn_make_buffer_rsrc(S_all + (size_t)b * slab_stride, 0,
                                             (int)(slab_stride * sizeof(double)), 0x00020000);
    S.ld = ld;
    S.loff = ((unsigned)lq * (unsigned)ld + (unsigned)li) * 8u;

    auto lds_int = [&](const int* p) { return __builtin_amdgcn_readfirstlane(*p); };
    auto slot_of = [&](int a, int c) { const int hi = max(a, c), lo = min(a, c); return hi * (hi + 1) / 2 + lo; };
    auto lds_load = [&](d4& acc, int slot) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = tiles[slot * 256 + r * 64 + lane];
    };
    auto lds_store = [&](const d4& acc, int slot) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tiles[slot * 256 + r * 64 + lane] = acc[r];
    };
    // rows that enter at step t: qa .. cend[t] - 1 (row t itself is wave 0's: its only tile is the diagonal one),
    // row q with its tiles (q, t) .. (q, q); entry e counts them row by row
    auto enter_first = [&](int t) { return max(t > 0 ? lds_int(&cendl[t - 1]) : 0, t + 1); };
    auto enter_count = [&](int t, int qa) {
        const int n = lds_int(&cendl[t]) - qa;
        return n <= 0 ? 0 : n * (qa - t + 1) + n * (n - 1) / 2;
    };
    auto enter_decode = [&](int e, int t, int qa, int& q, int& s) {
        q = qa;
        int w = qa - t + 1;
        while (e >= w) {
            e -= w;
            ++q;
            ++w;
        }
        s = t + e;
    };
    auto k_load = [&](d4& acc, int q, int s) {   // stiffness tile (q, s); zeros where K_ff has no entry (kmask)
        const unsigned km = (unsigned)lds_int(&kml[s]);
        const unsigned vo = S.lane_off(((km >> (q - s)) & 1u) != 0u);
        const int o = S.at(16 * s, 16 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = S.load_at(vo, o + r * (S.ld * 32));
    };
    d4 pre[WPF];
    auto issue = [&](int t) {
        const int qa = enter_first(t), cnt = enter_count(t, qa);
#pragma unroll
        for (int p = 0; p < WPF; ++p) {
            const int e = wave + WINW * p;
            if (e < cnt) {
                int q, s;
                enter_decode(e, t, qa, q, s);
                k_load(pre[p], q, s);
            }
        }
    };
    // wave 0: factor the diagonal tile of step t (in registers, fully updated), leave inv(L) and y_t behind
    auto factor_diag = [&](d4 D, int t) {
        const Chol16 fc = chol16_invert(D, hd.ch, hd.W[t & 1]);
        __builtin_amdgcn_wave_barrier();
        if (fc.bad >= 0) {
            if (lane == 0) hd.info = 16 * t + fc.bad + 1;
            return;
        }
        tile_store(fc.u, S, 16 * t, 16 * t);
        // y_t = inv(L_tt) f_t (as in the wave-per-matrix kernel: fragments times the vector in column order)
        const double yr = f[16 * t + li];
        const int src0 = (lane & 48) | lq;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = fma(hd.W[t & 1][wfrag_lane(r, lane)], __shfl(yr, src0 + 4 * r), acc);
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        __builtin_amdgcn_wave_barrier();
        if (lq == 0) f[16 * t + li] = acc;
    };

    Stamps st;   // (diagnostic builds) wave 0: 0 A work, 1 A wait, 2 look-ahead, 3 B wait; wave 1: 4 .. 7 likewise
    const int sb = wave == 0 ? 0 : 4;
    st.start();
    issue(0);
    if (wave == 0) {
        d4 D;
        tile_load(D, S, 0, 0);
        factor_diag(D, 0);
    }
    __syncthreads();
    int tm = 0;   // t mod M
    for (int t = 0; t < nch; ++t) {
        if (hd.info != 0) break;
        const int ce = lds_int(&cendl[t]);
        auto resid = [&](int q) { const int r = tm + (q - t); return r >= M ? r - M : r; };
        // ---- A: the pivot column ---------------------------------------------------------------------------
        {
            double wf[4], yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wf[r] = hd.W[t & 1][wfrag_lane(r, lane)];
                yv[r] = f[16 * t + lq + 4 * r];
            }
            auto solve_store = [&](const d4& T, int q) {
                d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) x = mfma_f64(wf[r], T[r], x);
                lds_store(x, slot_of(resid(q), tm));
                tile_store(x, S, 16 * t, 16 * q);
                double part = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fma(x[r], yv[r], part);
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                if (lq == 0) f[16 * q + li] -= part;
            };
            const int qa = enter_first(t), cnt = enter_count(t, qa);
#pragma unroll
            for (int p = 0; p < WPF; ++p) {
                const int e = wave + WINW * p;
                if (e < cnt) {
                    int q, s;
                    enter_decode(e, t, qa, q, s);
                    if (s == t) solve_store(pre[p], q);
                    else lds_store(pre[p], slot_of(resid(q), resid(s)));
                }
            }
            for (int e = wave + WINW * WPF; e < cnt; e += WINW) {   // more rows at once than the registers hold
                int q, s;
                enter_decode(e, t, qa, q, s);
                d4 T;
                k_load(T, q, s);
                if (s == t) solve_store(T, q);
                else lds_store(T, slot_of(resid(q), resid(s)));
            }
            for (int q = t + 1 + (WINW - 1 - wave); q < qa; q += WINW) {   // resident tiles of the column
                d4 T;
                lds_load(T, slot_of(resid(q), tm));
                solve_store(T, q);
            }
        }
        st.mark(sb + 0);
        __syncthreads();
        st.mark(sb + 1);
        if (t + 1 >= nch) break;
        // ---- B: trailing update, look-ahead on the next diagonal tile --------------------------------------
        issue(t + 1);
        const int m = ce - t - 1;   // rows below the pivot tile
        if (wave == 0) {
            __builtin_amdgcn_s_setprio(3);
            d4 D;
            if (m >= 1) {
                const int a = resid(t + 1);
                d4 X;
                lds_load(D, slot_of(a, a));
                lds_load(X, slot_of(a, tm));
#pragma unroll
                for (int r = 0; r < 4; ++r) D = mfma_f64_negA(X[r], X[r], D);
            } else {
                tile_load(D, S, 16 * (t + 1), 16 * (t + 1));
            }
            factor_diag(D, t + 1);
            __builtin_amdgcn_s_setprio(0);
        } else {
            const int njobs = m * (m + 1) / 2;
#ifdef TRS_EXP_WINDOW_LONE   // the waves that share wave 0's SIMD stay out of the update
            const int nupd = WINW - WINW / 4, urank = wave - 1 - wave / 4;
            for (int k = 1 + urank; k < njobs && (wave & 3) != 0; k += nupd) {
#else
            for (int k = wave; k < njobs; k += WINW - 1) {
#endif
                const int q = t + 1 + __builtin_amdgcn_readfirstlane((int)hd.tri_i[k]);
                const int s = t + 1 + __builtin_amdgcn_readfirstlane((int)hd.tri_j[k]);
                const int rq = resid(q), rs_ = resid(s);
                d4 Xs, Xq, T;
                lds_load(Xs, slot_of(rs_, tm));
                lds_load(Xq, slot_of(rq, tm));
                lds_load(T, slot_of(rq, rs_));
#pragma unroll
                for (int r = 0; r < 4; ++r) T = mfma_f64_negA(Xs[r], Xq[r], T);
                lds_store(T, slot_of(rq, rs_));
            }
        }
        tm = tm + 1 == M ? 0 : tm + 1;
        st.mark(sb + 2);
        __syncthreads();
        st.mark(sb + 3);
    }
    if (wave <= 1) st.flush();
    for (int i = tid; i < npad; i += 64 * WINW) ufb[i] = f[i];
    if (tid == 0) info[b] = hd.info;
}
