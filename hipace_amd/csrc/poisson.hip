// poisson.hip -- transverse Poisson solve Lap(F) = S with homogeneous Dirichlet walls as a 2-D type-I
// discrete sine transform (DST-I).
//
// Replaces FFTPoissonSolverDirichletFast (fields/fft_poisson_solver/
// FFTPoissonSolverDirichletFast.cpp:195-328); eigenvalues / normalisation follow
// FFTPoissonSolverDirichletDirect.cpp:58-83 (FFTW RODFT00 convention), so CPU goldens apply.
//
// DST-I of length n through one length-N (N = n+1) inverse FFT with Hermitian input: with y the odd
// 2N-periodic extension of the data (y_m = x_{m-1}, y_0 = y_N = 0) feed
//        Z_p = (y_{2p+1} - y_{2p-1}) + i y_{2p},      p = 0 .. N/2
// to an unnormalised complex-to-real transform r = C2R(Z); then for q = 1..n
//        T_{q-1} = 2 sum_m y_m sin(pi m q / N) = (r_{N-q} - r_q)/2 + (r_q + r_{N-q}) / (4 sin(pi q/N)).
// (The even-index samples enter through Im Z, the odd-index samples through the first difference in
// Re Z, which pulls out the factor 2 sin(pi q/N).)
//
// Two back-ends for the length-N transform:
//  * k_dst_rows<N1,N2>: own LDS-resident FFT for N = N1*N2 with small dense DFT stages.  The
//    benchmark size nx = 1024 needs N = 1025 = 25*41, which rocFFT only does through Bluestein
//    (4x slower than a length-1024 transform).  Two real rows are packed into one complex transform
//    (W = Z_a + i Z_b), pre- and post-processing are fused, so a pass reads and writes each row once.
//    Pass structure per solve: DSTx | transpose | DSTy * eigenvalues | DSTy | transpose | DSTx.
//  * rocFFT batched C2R + separate pre/post kernels for every other N.
#include "common.h"
#include "poisson_src.h"

#include <rocfft/rocfft.h>
#include <vector>
#include <cmath>

namespace hps {

// value of the odd extension at sample m for a row of T values held behind functor get(j),
// j = m-1 in [0, n)
template <class G>
__device__ __forceinline__ double odd_ext (int m, int N, G&& get)
{
    double sgn = 1.0;
    if (m < 0) { m = -m; sgn = -1.0; }
    if (m > N) { m = 2*N - m; sgn = -sgn; }
    if (m == 0 || m == N) return 0.0;
    return sgn*get(m - 1);
}

// ... of both rows of a pair at once (one 16-byte LDS read)
template <class G>
__device__ __forceinline__ double2 odd_ext2 (int m, int N, G&& get)
{
    double sgn = 1.0;
    if (m < 0) { m = -m; sgn = -1.0; }
    if (m > N) { m = 2*N - m; sgn = -sgn; }
    if (m == 0 || m == N) return make_double2(0.0, 0.0);
    const double2 v = get(m - 1);
    return make_double2(sgn*v.x, sgn*v.y);
}

// post-processing of one C2R output row: r has N = n+1 entries, k in [0, n)
__device__ __forceinline__ double dst_from_r (const double* r, int k, int N, double isin4)
{
    const double a = r[k + 1];
    const double b = r[N - 1 - k];
    return 0.5*(b - a) + (a + b)*isin4;
}

// =================================================================================================
// own transform: N = N1*N2, dense DFT-N1 over the strided index, twiddle, dense DFT-N2
// =================================================================================================
constexpr int DST_T = 2;            // complex transforms (= pairs of real rows) per workgroup
constexpr int DST_MAXPLANES = 4;

struct DstArgs {
    const double* src[DST_MAXPLANES]; long src_pitch;
    double* dst[DST_MAXPLANES]; long dst_pitch;
    const double* scale;            // optional [rows_per_plane][n] factor applied to the output
    const double2* fa;              // [N1][N1] DFT matrix exp(+2 pi i n1 k1 / N1), row n1
    const double2* fb;              // [N2][N2] DFT matrix exp(+2 pi i n2 k2 / N2), row n2
    const double2* tw;              // [N1][N2] twiddles exp(+2 pi i n2 k1 / N), row k1
    const double* isin4;            // 1/(4 sin(pi (k+1)/N)), k < n
    const double* ma;               // MFMA operand tables of the two small-DFT stages (k_dst_rows_mfma)
    const double* mb;
    int rows_per_plane, nplanes;
    // k_dst_rows_sym<.., true>: the rows of plane b are not read but formed from other planes while they are loaded,
    // value = sum over the plane's pairs of c * (p[idx] - q[idx]) (q may be null), idx = row*src_pitch + column -- the Poisson
    // sources of a slice straight from the slab's charge and current planes (no staging planes, no source kernel)
    const double* sp[DST_MAXPLANES][2]; const double* sq[DST_MAXPLANES][2]; double sc[DST_MAXPLANES][2]; int npairs[DST_MAXPLANES];
    long long* dbg;                 // optional: shader-clock stamps of workgroup 0 at the phase boundaries
    // blocked intermediate planes (k_dst_rows_sym<.., LIN, LOUT>, see "blocked layout" below)
    const int* gate;                // optional device word: the kernel returns at once when *gate == 0 (poisson_set_gate)
    int rows_pad;                   // rows per plane rounded up to the workgroup's 2T rows: a workgroup never straddles planes
    int blk_cols;                   // columns of the blocked planes (= rows per plane of the transposed view)
};
#ifdef HPS_POISSON_STAMPS
// stamps are kept in registers and written by HPS_STAMP_FLUSH at the end of the kernel: a global
// store ahead of the DFT-table loads would keep those off the scalar path
#define HPS_STAMP_DECL long long stamp_[6] = {0, 0, 0, 0, 0, 0}
#define HPS_STAMP(i) do { stamp_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define HPS_STAMP_FLUSH do { if (a.dbg && threadIdx.x == 0) { const int g_ = (int)gridDim.x, b_ = (int)blockIdx.x; \
        const int s_ = b_ == 0 ? 0 : b_ == g_/3 ? 1 : b_ == 2*g_/3 ? 2 : b_ == g_ - 1 ? 3 : -1; \
        if (s_ >= 0) { for (int q_ = 0; q_ < 6; ++q_) a.dbg[6*s_ + q_] = stamp_[q_]; } } } while (0)      /* workgroups 0, g/3, 2g/3, g-1 */
#else
#define HPS_STAMP_DECL do { } while (0)
#define HPS_STAMP(i) do { } while (0)
#define HPS_STAMP_FLUSH do { } while (0)
#endif

typedef __attribute__((address_space(3))) double lds_double;

// complex LDS array access (16-byte aligned pairs -> ds_read_b128 / ds_write_b128)
__device__ __forceinline__ double2 ldc (const lds_double* c, int i) { return make_double2(c[2*i], c[2*i + 1]); }
__device__ __forceinline__ void stc (lds_double* c, int i, double re, double im) { c[2*i] = re; c[2*i + 1] = im; }

__device__ __forceinline__ void cmac (double2& acc, const double2 x, const double2 w)
{
    acc.x = fma(x.x, w.x, acc.x); acc.x = fma(-x.y, w.y, acc.x);
    acc.y = fma(x.x, w.y, acc.y); acc.y = fma(x.y, w.x, acc.y);
}

// coalesced copy of T row pairs into LDS as complex (x_a[j], x_b[j]) at [t][j], j < n = N-1.  All loads of a thread are
// issued before its first LDS store (clamped addresses + selects instead of branches: written as a loop with conditional
// pointers this was six rounds of load -> wait -> store at the head of every pass)
template <int T, int N, int NT = 256>
__device__ __forceinline__ void load_row_pairs (lds_double* cbuf, const DstArgs& a, int row0, int total_rows, int tid)
{
    constexpr int n = N - 1, NJ = (n + NT - 1)/NT;
    double va[T][NJ], vb[T][NJ];
    bool oka[T], okb[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ra = row0 + 2*t, rb = ra + 1;
        oka[t] = ra < total_rows; okb[t] = rb < total_rows;
        const int ca = min(ra, total_rows - 1), cb = min(rb, total_rows - 1);
        const int pla = ca / a.rows_per_plane, plb = cb / a.rows_per_plane;
        const double* pa = a.src[pla] + (long)(ca - pla*a.rows_per_plane)*a.src_pitch;
        const double* pb = a.src[plb] + (long)(cb - plb*a.rows_per_plane)*a.src_pitch;
#pragma unroll
        for (int m = 0; m < NJ; ++m) { const int j = min(tid + NT*m, n - 1); va[t][m] = pa[j]; vb[t][m] = pb[j]; }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int m = 0; m < NJ; ++m) {
            const int j = tid + NT*m;
            if (j < n) stc(cbuf, t*N + j, oka[t] ? va[t][m] : 0.0, okb[t] ? vb[t][m] : 0.0);
        }
    }
}

// the same with the rows formed from other planes while they are loaded (DstArgs::sp / sq / sc)
template <int T, int N, int NT = 256>
__device__ __forceinline__ void load_row_pairs_src (lds_double* cbuf, const DstArgs& a, int row0, int total_rows, int tid)
{
    constexpr int n = N - 1, NJ = (n + NT - 1)/NT;
    double v[T][2][NJ];
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = min(row0 + 2*t + h, total_rows - 1);
            const int pl = r / a.rows_per_plane;
            const long ro = (long)(r - pl*a.rows_per_plane)*a.src_pitch;
            const int np = a.npairs[pl];
            const double* p0 = a.sp[pl][0] + ro; const double* q0 = a.sq[pl][0] ? a.sq[pl][0] + ro : nullptr;
            const double c0 = a.sc[pl][0];
            if (np == 1) {
#pragma unroll
                for (int m = 0; m < NJ; ++m) { const int j = min(tid + NT*m, n - 1); v[t][h][m] = q0 ? c0*(p0[j] - q0[j]) : c0*p0[j]; }
            } else {
                const double* p1 = a.sp[pl][1] + ro; const double* q1 = a.sq[pl][1] + ro;
                const double c1 = a.sc[pl][1];
#pragma unroll
                for (int m = 0; m < NJ; ++m) { const int j = min(tid + NT*m, n - 1); v[t][h][m] = c0*(p0[j] - q0[j]) + c1*(p1[j] - q1[j]); }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const bool oka = row0 + 2*t < total_rows, okb = row0 + 2*t + 1 < total_rows;
#pragma unroll
        for (int m = 0; m < NJ; ++m) {
            const int j = tid + NT*m;
            if (j < n) stc(cbuf, t*N + j, oka ? v[t][0][m] : 0.0, okb ? v[t][1][m] : 0.0);
        }
    }
}

#ifndef HPS_PRE_B128
#define HPS_PRE_B128 1
#endif
// Z entries of both rows of a pair -> W[p] and W[N-p] (registers), rows read from LDS
template <int T, int N, int PP, int NT = 256>
__device__ __forceinline__ void pre_to_regs (const lds_double* cbuf, int tid, double (&wr)[T][PP], double (&wi)[T][PP],
                                             double (&vr)[T][PP], double (&vi)[T][PP])
{
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int m = 0; m < PP; ++m) {
            const int p = tid + NT*m;
            wr[t][m] = wi[t][m] = vr[t][m] = vi[t][m] = 0.0;
            if (p <= N/2) {
#if HPS_PRE_B128
                // three 16-byte reads (lanes 32 B apart: 2-way bank conflicts) instead of six 8-byte ones (4-way)
                auto gc = [&] (int j) { return ldc(cbuf, t*N + j); };
                const double2 e1 = odd_ext2(2*p + 1, N, gc), e0 = odd_ext2(2*p - 1, N, gc), e2 = odd_ext2(2*p, N, gc);
                const double are = e1.x - e0.x, aim = e2.x, bre = e1.y - e0.y, bim = e2.y;
#else
                auto ga = [&] (int j) { return (double)cbuf[2*(t*N + j)]; };
                auto gb = [&] (int j) { return (double)cbuf[2*(t*N + j) + 1]; };
                const double are = odd_ext(2*p + 1, N, ga) - odd_ext(2*p - 1, N, ga), aim = odd_ext(2*p, N, ga);
                const double bre = odd_ext(2*p + 1, N, gb) - odd_ext(2*p - 1, N, gb), bim = odd_ext(2*p, N, gb);
#endif
                wr[t][m] = are - bim; wi[t][m] = aim + bre;      // W[p]
                vr[t][m] = are + bim; vi[t][m] = bre - aim;      // W[N-p]
            }
        }
    }
}

// r_a = Re X, r_b = Im X with X[q] stored at [q % N1][q / N1]; T_k from r_{k+1}, r_{N-1-k}; store.  The factors a thread
// needs from global memory (1/(4 sin), the optional scale rows) are requested ahead of the loop over the row pairs.
template <int T, int N1, int N2, int NT = 256, bool NOSCALE = false>
__device__ __forceinline__ void post_store (const lds_double* cbuf, const DstArgs& a, int row0, int total_rows, int tid)
{
    constexpr int N = N1*N2, n = N - 1, NK = (n + NT - 1)/NT;
    const double* scale = NOSCALE ? nullptr : a.scale;
    double is[NK], sca[T][NK], scb[T][NK];
#pragma unroll
    for (int m = 0; m < NK; ++m) is[m] = a.isin4[min(tid + NT*m, n - 1)];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ra = min(row0 + 2*t, total_rows - 1), rb = min(row0 + 2*t + 1, total_rows - 1);
        const int ja = ra - (ra / a.rows_per_plane)*a.rows_per_plane, jb = rb - (rb / a.rows_per_plane)*a.rows_per_plane;
#pragma unroll
        for (int m = 0; m < NK; ++m) {
            const int k = min(tid + NT*m, n - 1);
            sca[t][m] = scale ? scale[(long)ja*n + k] : 1.0;
            scb[t][m] = scale ? scale[(long)jb*n + k] : 1.0;
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ra = row0 + 2*t, rb = ra + 1;
        if (ra >= total_rows) break;
        const int pla = ra / a.rows_per_plane, plb = rb / a.rows_per_plane;
        const int ja = ra - pla*a.rows_per_plane, jb = rb - plb*a.rows_per_plane;
        double* da = a.dst[pla] + (long)ja*a.dst_pitch;
        double* db = (rb < total_rows) ? a.dst[plb] + (long)jb*a.dst_pitch : nullptr;
#pragma unroll
        for (int m = 0; m < NK; ++m) {
            const int k = tid + NT*m;
            if (k < n) {
                const int q1 = k + 1, q2 = N - 1 - k;
                const double2 x1 = ldc(cbuf, t*N + (q1 % N1)*N2 + q1/N1);
                const double2 x2 = ldc(cbuf, t*N + (q2 % N1)*N2 + q2/N1);
                double ta = 0.5*(x2.x - x1.x) + (x1.x + x2.x)*is[m];
                double tb = 0.5*(x2.y - x1.y) + (x1.y + x2.y)*is[m];
                if (scale) { ta *= sca[t][m]; tb *= scb[t][m]; }
                da[k] = ta;
                if (db) db[k] = tb;
            }
        }
    }
}

// ---- blocked layout of the intermediate planes: a solve without transposes -------------------------------------------------
// The x pass works on B = 2T rows per workgroup, the y pass on B columns.  Between the passes a plane is stored in blocks of
// B rows, element (row r, column k) at ((r / B)*ncols + k)*B + r % B: the x pass writes / reads its B rows as ONE contiguous
// run of ncols*B doubles ([k][B]: a (row 2t, row 2t+1) pair of a column is an aligned double2), and the y pass -- B columns
// k0..k0+B-1, all rows -- finds B*B contiguous doubles per row block ([k - k0][r], 288 B for B = 6) and transforms in place.
// Both k_transpose launches of a solve (a read and a write of every plane each) are gone.  Planes are padded to whole
// blocks (rows_pad), so a workgroup never straddles two planes.
// Pitch of a (row block, column) entry of a blocked plane, in doubles.  -DHPS_BLK_PAD=1 pads the block's B = 2T rows to whole
// 64-byte entries (8 for B = 6): the y pass's B columns are then 384 bytes = three whole 128-byte lines instead of 288-byte runs
// that straddle lines other workgroups complete, and the x pass writes whole lines (the pad as zeros).  Measured (round 4, call
// 17): Poisson phase 123.4 against 120.7 us per slice, 2127-2129 against 2137 slices/s with three stages in flight -- the
// straddled lines are not what the blocked y pass costs, and the padded planes are a third larger.  Off.
#ifndef HPS_BLK_PAD
#define HPS_BLK_PAD 0
#endif
constexpr int blk_pitch (int T) { return HPS_BLK_PAD ? ((2*T + 7) & ~7) : 2*T; }
struct BlkRow { int plane, j0; };
__device__ __forceinline__ BlkRow blk_row (const DstArgs& a, int row0)
{
    BlkRow r; r.plane = row0 / a.rows_pad; r.j0 = row0 - r.plane*a.rows_pad; return r;
}

// row-major rows (or rows formed from other planes) with the padded row numbering
template <int T, int N, int NT, bool SRC>
__device__ __forceinline__ void load_row_pairs_p (lds_double* cbuf, const DstArgs& a, int row0, int tid)
{
    constexpr int n = N - 1, NJ = (n + NT - 1)/NT;
    const BlkRow br = blk_row(a, row0);
    const int pl = br.plane;
    double v[T][2][NJ];
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = min(br.j0 + 2*t + h, a.rows_per_plane - 1);
            const long ro = (long)r*a.src_pitch;
            if constexpr (SRC) {
                const int np = a.npairs[pl];
                const double* p0 = a.sp[pl][0] + ro; const double* q0 = a.sq[pl][0] ? a.sq[pl][0] + ro : nullptr;
                const double c0 = a.sc[pl][0];
                if (np == 1) {
#pragma unroll
                    for (int m = 0; m < NJ; ++m) { const int j = min(tid + NT*m, n - 1); v[t][h][m] = q0 ? c0*(p0[j] - q0[j]) : c0*p0[j]; }
                } else {
                    const double* p1 = a.sp[pl][1] + ro; const double* q1 = a.sq[pl][1] + ro;
                    const double c1 = a.sc[pl][1];
#pragma unroll
                    for (int m = 0; m < NJ; ++m) { const int j = min(tid + NT*m, n - 1); v[t][h][m] = c0*(p0[j] - q0[j]) + c1*(p1[j] - q1[j]); }
                }
            } else {
                const double* p0 = a.src[pl] + ro;
#pragma unroll
                for (int m = 0; m < NJ; ++m) { const int j = min(tid + NT*m, n - 1); v[t][h][m] = p0[j]; }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const bool oka = br.j0 + 2*t < a.rows_per_plane, okb = br.j0 + 2*t + 1 < a.rows_per_plane;
#pragma unroll
        for (int m = 0; m < NJ; ++m) {
            const int j = tid + NT*m;
            if (j < n) stc(cbuf, t*N + j, oka ? v[t][0][m] : 0.0, okb ? v[t][1][m] : 0.0);
        }
    }
}

// the workgroup's B rows from a blocked plane: one contiguous run of n*T double2, item f = k*T + t = rows (2t, 2t+1) at column k
template <int T, int N, int NT>
__device__ __forceinline__ void load_blocked_rows (lds_double* cbuf, const DstArgs& a, int row0, int tid)
{
    constexpr int n = N - 1, ITEMS = n*T, NI = (ITEMS + NT - 1)/NT, BP2 = blk_pitch(T)/2;
    const BlkRow br = blk_row(a, row0);
    const double2* base = reinterpret_cast<const double2*>(a.src[br.plane] + (long)(br.j0/(2*T))*n*(2*BP2));
    double2 v[NI];
#pragma unroll
    for (int m = 0; m < NI; ++m) { const int f = min(tid + NT*m, ITEMS - 1); const int k = f / T, t = f - k*T; v[m] = base[k*BP2 + t]; }
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        const int f = tid + NT*m;
        if (f < ITEMS) {
            const int k = f / T, t = f - k*T;
            const bool oka = br.j0 + 2*t < a.rows_per_plane, okb = br.j0 + 2*t + 1 < a.rows_per_plane;
            stc(cbuf, t*N + k, oka ? v[m].x : 0.0, okb ? v[m].y : 0.0);
        }
    }
}

// the workgroup's B columns (its "rows" in the transposed view) from a blocked plane: per row block B*B contiguous doubles
// [kk][r]; item f = (jb, kk, u) is the double2 (rows jb*B + 2u, + 1) of column k0 + kk
template <int T, int N, int NT>
__device__ __forceinline__ void load_blocked_cols (lds_double* cbuf, const DstArgs& a, int row0, int tid)
{
    constexpr int n = N - 1, B = 2*T, NJB = (n + B - 1)/B, PER = B*(B/2), ITEMS = NJB*PER, NI = (ITEMS + NT - 1)/NT;
    const BlkRow br = blk_row(a, row0);
    const int k0 = br.j0;                                // first column of the workgroup (a multiple of B)
    const double2* plane = reinterpret_cast<const double2*>(a.src[br.plane]);
    double2 v[NI];
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        const int f = min(tid + NT*m, ITEMS - 1);
        const int jb = f / PER, e = f - jb*PER;
        const int kk = min(e / (B/2), a.rows_per_plane - 1 - k0);        // (columns past the plane's last: clamped, zeroed below)
        const int u = e - (e / (B/2))*(B/2);
        v[m] = plane[((long)jb*a.blk_cols + k0 + kk)*(blk_pitch(T)/2) + u];
    }
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        const int f = tid + NT*m;
        if (f < ITEMS) {
            const int jb = f / PER, e = f - jb*PER;
            const int kk = e / (B/2), u = e - kk*(B/2);
            const int t = kk >> 1, h = kk & 1, j = jb*B + 2*u;
            const bool ok = k0 + kk < a.rows_per_plane;
            if (j < n) cbuf[2*(t*N + j) + h] = ok ? v[m].x : 0.0;
            if (j + 1 < n) cbuf[2*(t*N + j + 1) + h] = ok ? v[m].y : 0.0;
        }
    }
}

// T_k of a row pair from the transform's output X (stored at [q % N1][q / N1])
template <int N1, int N2>
__device__ __forceinline__ double2 dst_pair (const lds_double* cbuf, int t, int k, double is)
{
    constexpr int N = N1*N2;
    const int q1 = k + 1, q2 = N - 1 - k;
    const double2 x1 = ldc(cbuf, t*N + (q1 % N1)*N2 + q1/N1);
    const double2 x2 = ldc(cbuf, t*N + (q2 % N1)*N2 + q2/N1);
    return make_double2(0.5*(x2.x - x1.x) + (x1.x + x2.x)*is, 0.5*(x2.y - x1.y) + (x1.y + x2.y)*is);
}

// output rows -> their block of a blocked plane (contiguous double2 run, as load_blocked_rows reads it)
template <int T, int N1, int N2, int NT>
__device__ __forceinline__ void store_blocked_rows (const lds_double* cbuf, const DstArgs& a, int row0, int tid)
{
    constexpr int N = N1*N2, n = N - 1, BP2 = blk_pitch(T)/2, ITEMS = n*BP2, NI = (ITEMS + NT - 1)/NT;      // (pad slots included: whole 64-byte entries)
    const BlkRow br = blk_row(a, row0);
    double2* base = reinterpret_cast<double2*>(a.dst[br.plane] + (long)(br.j0/(2*T))*n*(2*BP2));
    double is[NI];
#pragma unroll
    for (int m = 0; m < NI; ++m) is[m] = a.isin4[min(tid + NT*m, ITEMS - 1)/BP2];
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        const int f = tid + NT*m;
        if (f < ITEMS) {
            const int k = f / BP2, t = f - k*BP2;
            base[f] = t < T ? dst_pair<N1, N2>(cbuf, t, k, is[m]) : make_double2(0.0, 0.0);
        }
    }
}

// output "rows" (columns k0 + 2t, k0 + 2t + 1 of the blocked plane), element j -> ((j / B)*ncols + column)*B + j % B
template <int T, int N1, int N2, int NT>
__device__ __forceinline__ void store_blocked_cols (const lds_double* cbuf, const DstArgs& a, int row0, int tid)
{
    constexpr int N = N1*N2, n = N - 1, B = 2*T, NK = (n + NT - 1)/NT;
    const BlkRow br = blk_row(a, row0);
    const int k0 = br.j0;
    double* plane = a.dst[br.plane];
    double is[NK];
#pragma unroll
    for (int m = 0; m < NK; ++m) is[m] = a.isin4[min(tid + NT*m, n - 1)];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (k0 + 2*t >= a.rows_per_plane) break;
        const bool okb = k0 + 2*t + 1 < a.rows_per_plane;
#pragma unroll
        for (int m = 0; m < NK; ++m) {
            const int j = tid + NT*m;
            if (j < n) {
                const double2 v = dst_pair<N1, N2>(cbuf, t, j, is[m]);
                const int jb = j / B, r = j - jb*B;
                double* p = plane + ((long)jb*a.blk_cols + k0 + 2*t)*blk_pitch(T) + r;
                p[0] = v.x;
                if (okb) p[blk_pitch(T)] = v.y;
            }
        }
    }
}

// row-major output with the padded row numbering (the last pass of a blocked solve writes the destination planes)
template <int T, int N1, int N2, int NT>
__device__ __forceinline__ void post_store_p (const lds_double* cbuf, const DstArgs& a, int row0, int tid)
{
    constexpr int N = N1*N2, n = N - 1, NK = (n + NT - 1)/NT;
    const BlkRow br = blk_row(a, row0);
    double is[NK];
#pragma unroll
    for (int m = 0; m < NK; ++m) is[m] = a.isin4[min(tid + NT*m, n - 1)];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ja = br.j0 + 2*t;
        if (ja >= a.rows_per_plane) break;
        double* da = a.dst[br.plane] + (long)ja*a.dst_pitch;
        double* db = (ja + 1 < a.rows_per_plane) ? da + a.dst_pitch : nullptr;
#pragma unroll
        for (int m = 0; m < NK; ++m) {
            const int k = tid + NT*m;
            if (k < n) {
                const double2 v = dst_pair<N1, N2>(cbuf, t, k, is[m]);
                da[k] = v.x;
                if (db) db[k] = v.y;
            }
        }
    }
}

template <int N1, int N2>
__global__ __launch_bounds__(256)
void k_dst_rows (DstArgs a)
{
    if (a.gate && *a.gate == 0) return;
    constexpr int N = N1*N2, T = DST_T;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_double* cbuf = (lds_double*)lds_raw;          // [T][N] complex working set
    lds_double* fa = cbuf + 2*T*N;                    // [N1][N1] first-stage DFT matrix
    lds_double* fb = fa + 2*N1*N1;                    // [N2][N2] second-stage DFT matrix
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int total_rows = a.rows_per_plane*a.nplanes;
    const int row0 = blockIdx.x*2*T;

    HPS_STAMP_DECL;
    HPS_STAMP(0);
    for (int k = tid; k < N1*N1; k += 256) { const double2 w = a.fa[k]; stc(fa, k, w.x, w.y); }
    for (int k = tid; k < N2*N2; k += 256) { const double2 w = a.fb[k]; stc(fb, k, w.x, w.y); }

    // ---- pre: W = Z_a + i Z_b with the Hermitian halves unfolded ------------------------------
    // (1) coalesced copy of the row pairs into LDS as (x_a[j], x_b[j]); (2) every thread forms its
    // Z entries in registers; (3) W overwrites the rows.
    load_row_pairs<T, N>(cbuf, a, row0, total_rows, tid);
    __syncthreads();
    HPS_STAMP(1);
    {
        constexpr int PP = (N/2 + 1 + 255)/256;
        double wr[T][PP], wi[T][PP], vr[T][PP], vi[T][PP];
        pre_to_regs<T, N, PP>(cbuf, tid, wr, wi, vr, vi);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int m = 0; m < PP; ++m) {
                const int p = tid + 256*m;
                if (p <= N/2) {
                    stc(cbuf, t*N + p, wr[t][m], wi[t][m]);
                    if (p > 0 && 2*p != N) stc(cbuf, t*N + N - p, vr[t][m], vi[t][m]);
                }
            }
        }
    }
    __syncthreads();

    HPS_STAMP(2);
    // ---- stage A: for every (t, n2) column the DFT over n1, then the twiddle w_N^(n2 k1) --------
    {
        constexpr int ITEMS = T*N2;
        constexpr int NW = (ITEMS + 63)/64;           // waves that cover all columns once
        constexpr int NG = (4/NW) > 0 ? (4/NW) : 1;   // groups of waves, each owning a block of k1
        constexpr int KB = (N1 + NG - 1)/NG;
        static_assert(NW <= 4, "too many columns for one workgroup");
        const int g = wave / NW;
        const int item = (wave % NW)*64 + lane;
        const bool active = (g < NG) && (item < ITEMS);
        const int t = item / N2, n2 = item - t*N2;
        const int k10 = g*KB;
        double2 acc[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) acc[kk] = make_double2(0.0, 0.0);
        if (active) {
            // the matrix row of n1 is contiguous in k1: constant LDS offsets, no index arithmetic
            for (int n1 = 0; n1 < N1; ++n1) {
                const double2 x = ldc(cbuf, t*N + n1*N2 + n2);
                const lds_double* frow = fa + 2*(n1*N1 + k10);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk)
                    if (k10 + kk < N1) cmac(acc[kk], x, ldc(frow, kk));
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const int k1 = k10 + kk;
                if (k1 < N1) {
                    const double2 w = a.tw[k1*N2 + n2];
                    stc(cbuf, t*N + k1*N2 + n2, acc[kk].x*w.x - acc[kk].y*w.y, acc[kk].x*w.y + acc[kk].y*w.x);
                }
            }
        }
        __syncthreads();
    }

    HPS_STAMP(3);
    // ---- stage B: for every (t, k1) row the DFT over n2; result X[k1 + N1 k2] stays at [k1][k2] ----
    {
        constexpr int ITEMS = T*N1;
        constexpr int NW = (ITEMS + 63)/64;
        constexpr int NG = (4/NW) > 0 ? (4/NW) : 1;
        constexpr int KB = (N2 + NG - 1)/NG;
        static_assert(NW <= 4, "too many rows for one workgroup");
        const int g = wave / NW;
        const int item = (wave % NW)*64 + lane;
        const bool active = (g < NG) && (item < ITEMS);
        const int t = item / N1, k1 = item - t*N1;
        const int k20 = g*KB;
        double2 acc[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) acc[kk] = make_double2(0.0, 0.0);
        if (active) {
            for (int n2 = 0; n2 < N2; ++n2) {
                const double2 x = ldc(cbuf, t*N + k1*N2 + n2);
                const lds_double* frow = fb + 2*(n2*N2 + k20);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk)
                    if (k20 + kk < N2) cmac(acc[kk], x, ldc(frow, kk));
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) if (k20 + kk < N2) stc(cbuf, t*N + k1*N2 + k20 + kk, acc[kk].x, acc[kk].y);
        }
        __syncthreads();
    }

    HPS_STAMP(4);
    // ---- post: r_a = Re X, r_b = Im X; T_k from r_{k+1} and r_{N-1-k}; optional scaling; store ----
    post_store<T, N1, N2>(cbuf, a, row0, total_rows, tid);
    HPS_STAMP(5);
    HPS_STAMP_FLUSH;
}


// ---- N a power of two: in-place radix-4 transform ------------------------------------------------
// The grids the reference recommends have 2^K - 1 cells per side (docs/source/run/parameters.rst:313-321), i.e. N = n + 1 =
// 2^K: no dense small-DFT stages there -- a decimation-in-frequency radix-4 pass per pair of bits (one radix-2 pass at the
// end when K is odd), in place in LDS, twiddles from a copy of w_N^k in LDS.  X[k] = sum_n x[n] exp(+2 pi i n k / N) ends up
// at position pow2_pos(k) (its base-4 digits reversed); the post-processing reads through that map, so there is no
// reordering pass.  Same pre- and post-processing, same DstArgs, same launch geometry (2T rows per workgroup) as k_dst_rows.
template <int LOGN>
__device__ __forceinline__ int pow2_pos (int k)
{
    int pos = 0, span = 1 << LOGN;
#pragma unroll
    for (int s = 0; s < LOGN/2; ++s) { span >>= 2; pos += (k & 3)*span; k >>= 2; }
    if (LOGN & 1) pos += (k & 1);
    return pos;
}

// LDS bank swizzle of the transform's working set (N = 4^5 only).  In place, a radix-4 pass with quarter span 4 or 1 walks the
// array with lanes 16 or 4 complex numbers apart, and the post-processing reads X[k] at its digit-reversed position -- lanes 256
// apart: 16 lanes of a ds_read_b128 on the same four banks.  With the base-4 digits a0..a4 of the index, the low two digits are
// stored as (a1 ^ a2 ^ a3, a0 ^ a2 ^ a4): every access pattern of the kernel -- natural order, the five passes, the digit-reversed
// reads -- then varies the stored low digits over all 16 values within 16 consecutive lanes.
#ifndef HPS_DSTP_SWIZZLE
#define HPS_DSTP_SWIZZLE 1
#endif
template <int LOGN>
__device__ __forceinline__ int pow2_swz (int i)
{
    if constexpr (LOGN == 10 && HPS_DSTP_SWIZZLE) {
        const int a2 = (i >> 4) & 3, a3 = (i >> 6) & 3, a4 = (i >> 8) & 3;
        return i ^ (((a2 ^ a3) << 2) | (a2 ^ a4));
    } else if constexpr (LOGN == 9 && HPS_DSTP_SWIZZLE) {
        // N = 4^4 * 2 (quarter spans 128, 32, 8, 2, then the radix-2 pass; digit-reversed reads vary bits 5..8): by bits b4..b8
        const int b4 = (i >> 4) & 1, b5 = (i >> 5) & 1, b6 = (i >> 6) & 1, b7 = (i >> 7) & 1, b8 = (i >> 8) & 1;
        return i ^ ((b4 ^ b7) | ((b4 ^ b8) << 1) | (b5 << 2) | ((b5 ^ b6) << 3));
    } else return i;
}

template <int LOGN, int T, int NT>
__device__ __forceinline__ void pow2_fft (lds_double* cbuf, const lds_double* twl, int tid)
{
    constexpr int N = 1 << LOGN;
#pragma unroll
    for (int s = 0; s < LOGN/2; ++s) {
        const int lq = LOGN - 2*s - 2;                 // log2 of the quarter span
        const int q = 1 << lq, L = q << 2;
#pragma unroll
        for (int b0 = 0; b0 < T*N/4; b0 += NT) {
            const int b = b0 + tid;
            if (T*N/4 % NT == 0 || b < T*N/4) {
                const int t = b >> (LOGN - 2), bb = b & (N/4 - 1);
                const int g = bb >> lq, j = bb & (q - 1);
                const int l0 = g*L + j, tb = t*N;
                const int p0 = tb + pow2_swz<LOGN>(l0), p1 = tb + pow2_swz<LOGN>(l0 + q), p2 = tb + pow2_swz<LOGN>(l0 + 2*q), p3 = tb + pow2_swz<LOGN>(l0 + 3*q);
                const double2 x0 = ldc(cbuf, p0), x1 = ldc(cbuf, p1), x2 = ldc(cbuf, p2), x3 = ldc(cbuf, p3);
                const double t0r = x0.x + x2.x, t0i = x0.y + x2.y, t1r = x0.x - x2.x, t1i = x0.y - x2.y;
                const double t2r = x1.x + x3.x, t2i = x1.y + x3.y, dr = x1.x - x3.x, di = x1.y - x3.y;
                double y1r = t1r - di, y1i = t1i + dr;          // t1 + i d
                double y2r = t0r - t2r, y2i = t0i - t2i;
                double y3r = t1r + di, y3i = t1i - dr;          // t1 - i d
                if (lq > 0) {
                    const int tj = j << (2*s);                  // exponent of w_N for w_L^j
                    const double2 w1 = ldc(twl, tj), w2 = ldc(twl, 2*tj), w3 = ldc(twl, 3*tj);
                    double a_;
                    a_ = y1r*w1.x - y1i*w1.y; y1i = y1r*w1.y + y1i*w1.x; y1r = a_;
                    a_ = y2r*w2.x - y2i*w2.y; y2i = y2r*w2.y + y2i*w2.x; y2r = a_;
                    a_ = y3r*w3.x - y3i*w3.y; y3i = y3r*w3.y + y3i*w3.x; y3r = a_;
                }
                stc(cbuf, p0, t0r + t2r, t0i + t2i);
                stc(cbuf, p1, y1r, y1i);
                stc(cbuf, p2, y2r, y2i);
                stc(cbuf, p3, y3r, y3i);
            }
        }
        __syncthreads();
    }
    if (LOGN & 1) {
#pragma unroll
        for (int b0 = 0; b0 < T*N/2; b0 += NT) {
            const int b = b0 + tid;
            if (T*N/2 % NT == 0 || b < T*N/2) {
                const int tb = (2*b) & ~(N - 1), l0 = (2*b) & (N - 1);
                const int p0 = tb + pow2_swz<LOGN>(l0), p1 = tb + pow2_swz<LOGN>(l0 + 1);
                const double2 x0 = ldc(cbuf, p0), x1 = ldc(cbuf, p1);
                stc(cbuf, p0, x0.x + x1.x, x0.y + x1.y);
                stc(cbuf, p1, x0.x - x1.x, x0.y - x1.y);
            }
        }
        __syncthreads();
    }
}

// post_store with the transform's output order behind a map (POS::at(q) = where X[q] sits)
template <int T, int N, int NT, class POS, bool NOSCALE = false>
__device__ __forceinline__ void post_store_map (const lds_double* cbuf, const DstArgs& a, int row0, int total_rows, int tid)
{
    constexpr int n = N - 1, NK = (n + NT - 1)/NT;
    const double* scale = NOSCALE ? nullptr : a.scale;
    double is[NK], sca[T][NK], scb[T][NK];
#pragma unroll
    for (int m = 0; m < NK; ++m) is[m] = a.isin4[min(tid + NT*m, n - 1)];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ra = min(row0 + 2*t, total_rows - 1), rb = min(row0 + 2*t + 1, total_rows - 1);
        const int ja = ra - (ra / a.rows_per_plane)*a.rows_per_plane, jb = rb - (rb / a.rows_per_plane)*a.rows_per_plane;
#pragma unroll
        for (int m = 0; m < NK; ++m) {
            const int k = min(tid + NT*m, n - 1);
            sca[t][m] = scale ? scale[(long)ja*n + k] : 1.0;
            scb[t][m] = scale ? scale[(long)jb*n + k] : 1.0;
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ra = row0 + 2*t, rb = ra + 1;
        if (ra >= total_rows) break;
        const int pla = ra / a.rows_per_plane, plb = rb / a.rows_per_plane;
        const int ja = ra - pla*a.rows_per_plane, jb = rb - plb*a.rows_per_plane;
        double* da = a.dst[pla] + (long)ja*a.dst_pitch;
        double* db = (rb < total_rows) ? a.dst[plb] + (long)jb*a.dst_pitch : nullptr;
#pragma unroll
        for (int m = 0; m < NK; ++m) {
            const int k = tid + NT*m;
            if (k < n) {
                const double2 x1 = ldc(cbuf, t*N + POS::at(k + 1));
                const double2 x2 = ldc(cbuf, t*N + POS::at(N - 1 - k));
                double ta = 0.5*(x2.x - x1.x) + (x1.x + x2.x)*is[m];
                double tb = 0.5*(x2.y - x1.y) + (x1.y + x2.y)*is[m];
                if (scale) { ta *= sca[t][m]; tb *= scb[t][m]; }
                da[k] = ta;
                if (db) db[k] = tb;
            }
        }
    }
}
template <int LOGN> struct Pow2Pos { static __device__ __forceinline__ int at (int q) { return pow2_swz<LOGN>(pow2_pos<LOGN>(q)); } };

#ifndef HPS_DSTP_T
#define HPS_DSTP_T 2
#endif
#ifndef HPS_DSTP_NT
#define HPS_DSTP_NT 256
#endif
constexpr int DSTP_T = HPS_DSTP_T, DSTP_NT = HPS_DSTP_NT;      // row pairs and threads per workgroup of the power-of-two kernel
// SRC: the rows are formed from other planes while they are loaded (DstArgs::sp / sq / sc, as k_dst_rows_sym<.., true>)
// TWICE: the two y passes of a solve in one launch -- transform, times a.scale (the inverse eigenvalues), transform again
template <int LOGN, bool SRC = false, bool TWICE = false>
__global__ __launch_bounds__(DSTP_NT)
void k_dst_rows_pow2 (DstArgs a)
{
    if (a.gate && *a.gate == 0) return;
    constexpr int N = 1 << LOGN, n = N - 1, T = DSTP_T, NT = DSTP_NT;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_double* cbuf = (lds_double*)lds_raw;          // [T][N] complex working set
    lds_double* twl = cbuf + 2*T*N;                   // [N] w_N^k
    const int tid = threadIdx.x;
    const int total_rows = a.rows_per_plane*a.nplanes;
    const int row0 = blockIdx.x*2*T;
    {   constexpr int NW = N/NT > 0 ? N/NT : 1;
        double2 w[NW];
#pragma unroll
        for (int m = 0; m < NW; ++m) w[m] = a.tw[min(tid + NT*m, N - 1)];
        if constexpr (SRC) load_row_pairs_src<T, N, NT>(cbuf, a, row0, total_rows, tid);
        else load_row_pairs<T, N, NT>(cbuf, a, row0, total_rows, tid);
#pragma unroll
        for (int m = 0; m < NW; ++m) { const int k = tid + NT*m; if (k < N) stc(twl, k, w[m].x, w[m].y); }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < (TWICE ? 2 : 1); ++pass) {
        {
            constexpr int PP = (N/2 + 1 + NT - 1)/NT;
            double wr[T][PP], wi[T][PP], vr[T][PP], vi[T][PP];
            pre_to_regs<T, N, PP, NT>(cbuf, tid, wr, wi, vr, vi);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int m = 0; m < PP; ++m) {
                    const int p = tid + NT*m;
                    if (p <= N/2) {
                        stc(cbuf, t*N + pow2_swz<LOGN>(p), wr[t][m], wi[t][m]);
                        if (p > 0 && 2*p != N) stc(cbuf, t*N + pow2_swz<LOGN>(N - p), vr[t][m], vi[t][m]);
                    }
                }
            }
        }
        __syncthreads();
        pow2_fft<LOGN, T, NT>(cbuf, twl, tid);
        if (TWICE && pass == 0) {
            // T_k of both rows of every pair (times the scale) -> registers -> back to [t][k] as the next pass's input
            constexpr int NK = (n + NT - 1)/NT;
            double ta[T][NK], tb[T][NK];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int ra = min(row0 + 2*t, total_rows - 1), rb = min(row0 + 2*t + 1, total_rows - 1);
                const int ja = ra - (ra / a.rows_per_plane)*a.rows_per_plane, jb = rb - (rb / a.rows_per_plane)*a.rows_per_plane;
#pragma unroll
                for (int m = 0; m < NK; ++m) {
                    const int k = min(tid + NT*m, n - 1);
                    const double2 x1 = ldc(cbuf, t*N + Pow2Pos<LOGN>::at(k + 1));
                    const double2 x2 = ldc(cbuf, t*N + Pow2Pos<LOGN>::at(N - 1 - k));
                    const double is = a.isin4[k];
                    ta[t][m] = (0.5*(x2.x - x1.x) + (x1.x + x2.x)*is)*a.scale[(long)ja*n + k];
                    tb[t][m] = (0.5*(x2.y - x1.y) + (x1.y + x2.y)*is)*a.scale[(long)jb*n + k];
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int m = 0; m < NK; ++m) {
                    const int k = tid + NT*m;
                    if (k < n) stc(cbuf, t*N + k, ta[t][m], tb[t][m]);
                }
            }
            __syncthreads();
        }
    }
    post_store_map<T, N, NT, Pow2Pos<LOGN>, TWICE>(cbuf, a, row0, total_rows, tid);
}

// ---- odd factors: conjugate-symmetric small DFTs -------------------------------------------------
// For odd M and H = (M-1)/2, with s_n = x_n + x_{M-n}, d_n = x_n - x_{M-n}:
//   X_0 = x_0 + sum_n s_n,   X_k = x_0 + P_k + i Q_k,   X_{M-k} = x_0 + P_k - i Q_k   (k = 1..H)
//   P_k = sum_{n=1..H} s_n cos(2 pi n k / M),   Q_k = sum_{n=1..H} d_n sin(2 pi n k / M)
// i.e. M^2 real FMAs per DFT instead of 4 M^2, with one (cos, sin) table read per 4 FMAs.
#ifndef HPS_SYM_UNROLL
#define HPS_SYM_UNROLL 2
#endif
#ifndef HPS_DSTS_T
#define HPS_DSTS_T 3
#endif
#ifndef HPS_DSTS_NT
#define HPS_DSTS_NT 512
#endif
constexpr int DSTS_T = HPS_DSTS_T;
constexpr int DSTS_NT = HPS_DSTS_NT;        // threads per workgroup of the symmetric kernel

#ifndef HPS_SYM_LEFTOVER
#define HPS_SYM_LEFTOVER 1
#endif
constexpr int cmin_i (int a, int b) { return a < b ? a : b; }

// does sym_stage<M, .., ITEMS_PER_T, .., T, NWAVES> take the LEFTOVER scheme (below)?  Its caller then keeps the stage's table in LDS.
template <int M, int ITEMS_PER_T, int T, int NWAVES>
constexpr bool sym_stage_leftover ()
{
    constexpr int H = (M - 1)/2, ITEMS = T*ITEMS_PER_T, NW = (ITEMS + 63)/64;
    constexpr int NG0 = (NWAVES/NW) > 0 ? (NWAVES/NW) : 1, NG = NG0 > H ? H : NG0, KB = (H + NG - 1)/NG;
    constexpr int FULL = ITEMS/64, REM = ITEMS%64;
    constexpr int NGX0 = (FULL >= 1 && REM > 0) ? cmin_i(cmin_i(64/REM, (NWAVES - 1)/FULL), H) : 0;
    constexpr int KBX = NGX0 > 0 ? (H + NGX0 - 1)/NGX0 : 1, NGX = (H + KBX - 1)/KBX, NGU = (H + KB - 1)/KB;
    // Measured (MI355X, profiles/r04g_ab_poisson_leftover.txt): the 19-point stage of 512^2 (81 items, H = 9: 4 wave-tasks of blocks
    // of 3 instead of 8) -7.7 us per slice (Poisson 64.6 -> 56.9 us, 3331 -> 3407 slices/s); the 41-point stage of 1024^2 (75 items,
    // H = 20: 6 wave-tasks of blocks of 4 instead of 8 of 5) +4.7 us on row-major planes and +37 us on blocked ones -- the second
    // loop body costs the kernel 27 VGPRs (76 -> 103; the two-transform kernel 115 -> 132: one workgroup per CU).  Small factors only.
    return HPS_SYM_LEFTOVER && M <= 27 && NGX0 >= 2 && (FULL*NGX + 1)*(6 + 4*KBX) < NW*NGU*(6 + 4*KB);
}

// ltab: the stage's (cos, sin) table in LDS ([H][H] complex) for the wave of the remaining items, or null: plain scheme only
template <int M, int STRIDE, int KSTRIDE, int ITEMS_PER_T, int ITEM_STRIDE, bool TWIDDLE, int T, int NWAVES, bool LEFTOK = false>
__device__ __forceinline__ void sym_stage (lds_double* cbuf, const double2* __restrict__ cs, const double2* __restrict__ tw, int wave, int lane,
                                           const lds_double* ltab = nullptr)
{
    // one item = one DFT of size M over elements base + n*STRIDE; results go to base + k*KSTRIDE
    constexpr int H = (M - 1)/2;
    constexpr int ITEMS = T*ITEMS_PER_T;
    constexpr int NW = (ITEMS + 63)/64;
    constexpr int NG0 = (NWAVES/NW) > 0 ? (NWAVES/NW) : 1;
    constexpr int NG = NG0 > H ? H : NG0;
    constexpr int KB = (H + NG - 1)/NG;
    constexpr int NT = M*ITEMS_PER_T;                 // complex elements per transform
    static_assert(NW <= NWAVES, "too many items for one workgroup");
    // Who does what.  Plain scheme: NW waves cover the items once, NG groups of such waves share the outputs k = 1..H in blocks
    // of KB (a wave's block is uniform: its table rows come through the scalar cache).  When the items do not fill their last
    // wave -- the 41-point stage at 1024^2 has 3 x 25 = 75: the second wave of every group works with 11 of its 64 lanes --
    // the LEFTOVER scheme gives every full wave of items one k block and packs the remaining items of ALL k blocks into one
    // more wave (lane -> (item, block); that wave reads its table entries per lane, from a copy of the table in LDS -- per-lane
    // global loads were a chain of 20 dependent trips to the L2: the stage took 34 us longer): 75 items, H = 20:
    // 5 waves with blocks of 4 + 1 wave instead of 8 waves with blocks of 5 -- 6 x (6 + 16) instead of 8 x (6 + 20) instructions
    // per term of the sum.  Taken where it is less work.
    constexpr int FULL = ITEMS/64, REM = ITEMS%64;
    constexpr int NGX0 = (FULL >= 1 && REM > 0) ? cmin_i(cmin_i(64/REM, (NWAVES - 1)/FULL), H) : 0;
    constexpr int KBX = NGX0 > 0 ? (H + NGX0 - 1)/NGX0 : 1;
    constexpr int NGX = (H + KBX - 1)/KBX;            // blocks of KBX that cover 1..H
    constexpr int NGU = (H + KB - 1)/KB;              // (plain scheme: groups that have a block at all)
    constexpr bool LEFT = LEFTOK && sym_stage_leftover<M, ITEMS_PER_T, T, NWAVES>();
    constexpr int KBE = LEFT ? KBX : KB;
    const bool lw = LEFT && (wave == FULL*NGX);       // the wave of the remaining items (wave-uniform)
    const int gs = LEFT ? wave / (FULL > 0 ? FULL : 1) : wave / NW;      // this wave's k block where it is uniform
    int g = gs, item = LEFT ? (wave - gs*FULL)*64 + lane : (wave % NW)*64 + lane;
    bool active = LEFT ? (wave < FULL*NGX) : ((gs < NG) && (item < ITEMS));
    if (lw) { g = lane / (REM > 0 ? REM : 1); item = 64*FULL + (lane - g*REM); active = lane < REM*NGX; }
    const int t = item / ITEMS_PER_T, r = item - t*ITEMS_PER_T;
    const int base = t*NT + r*ITEM_STRIDE;
    const int k0 = 1 + g*KBE;                         // first k of this lane's block
    double pr[KBE], pim[KBE], qr[KBE], qi[KBE];
#pragma unroll
    for (int kk = 0; kk < KBE; ++kk) pr[kk] = pim[kk] = qr[kk] = qi[kk] = 0.0;
    double2 x0 = make_double2(0.0, 0.0), ssum = make_double2(0.0, 0.0);
    if (active) {
        x0 = ldc(cbuf, base);
        if (!lw) {
            const int k0s = 1 + gs*KBE;
            // (requesting step n+1's data pair and table row ahead of step n's FMAs by hand was measured slower)
#pragma unroll HPS_SYM_UNROLL
            for (int n = 1; n <= H; ++n) {
                const double2 xa = ldc(cbuf, base + n*STRIDE), xb = ldc(cbuf, base + (M - n)*STRIDE);
                const double sr = xa.x + xb.x, si = xa.y + xb.y, dr = xa.x - xb.x, di = xa.y - xb.y;
                ssum.x += sr; ssum.y += si;
                // wave-uniform address: the table is read through the scalar cache, not the LDS pipe
                // (through the constant address space: the load stays a scalar one whatever else the kernel does ahead of it --
                //  a harmless edit at the kernel's head once turned these into vector loads: 74 -> 108 VGPRs, 150 -> 185 us)
                typedef const __attribute__((address_space(4))) double cdouble;
                cdouble* row = (cdouble*)(cs + ((n - 1)*H + (k0s - 1)));
#pragma unroll
                for (int kk = 0; kk < KBE; ++kk) {
                    if (k0s + kk <= H) {
                        const double2 c = make_double2(row[2*kk], row[2*kk + 1]);            // (cos, sin)(2 pi n k / M)
                        pr[kk] = fma(sr, c.x, pr[kk]); pim[kk] = fma(si, c.x, pim[kk]);
                        qr[kk] = fma(dr, c.y, qr[kk]); qi[kk] = fma(di, c.y, qi[kk]);
                    }
                }
            }
        } else {
            // the remaining items: the block differs from lane to lane, the table entries are per-lane loads
#pragma unroll 2
            for (int n = 1; n <= H; ++n) {
                const double2 xa = ldc(cbuf, base + n*STRIDE), xb = ldc(cbuf, base + (M - n)*STRIDE);
                const double sr = xa.x + xb.x, si = xa.y + xb.y, dr = xa.x - xb.x, di = xa.y - xb.y;
                ssum.x += sr; ssum.y += si;
                const int row = (n - 1)*H + (k0 - 1);
#pragma unroll
                for (int kk = 0; kk < KBE; ++kk) {
                    const double2 c = ldc(ltab, row + min(kk, H - k0));      // (clamped: a k beyond H is not stored)
                    pr[kk] = fma(sr, c.x, pr[kk]); pim[kk] = fma(si, c.x, pim[kk]);
                    qr[kk] = fma(dr, c.y, qr[kk]); qi[kk] = fma(di, c.y, qi[kk]);
                }
            }
        }
    }
    __syncthreads();
    if (active) {
        auto put = [&] (int k, double re, double im) {
            if (TWIDDLE) { const double2 w = tw[k*ITEMS_PER_T + r]; const double a = re*w.x - im*w.y; im = re*w.y + im*w.x; re = a; }
            stc(cbuf, base + k*KSTRIDE, re, im);
        };
        if (g == 0) stc(cbuf, base, x0.x + ssum.x, x0.y + ssum.y);      // k = 0, twiddle 1
#pragma unroll
        for (int kk = 0; kk < KBE; ++kk) {
            const int k = k0 + kk;
            if (k <= H) {
                const double ar = x0.x + pr[kk], ai = x0.y + pim[kk];
                put(k, ar - qi[kk], ai + qr[kk]);               // x0 + P + iQ
                put(M - k, ar + qi[kk], ai - qr[kk]);           // x0 + P - iQ
            }
        }
    }
    __syncthreads();
}

template <int N1, int N2, bool SRC = false, bool TWICE = false, int LIN = 0, int LOUT = 0>
__global__ __launch_bounds__(DSTS_NT) void k_dst_rows_sym (DstArgs a)
{
    // LIN / LOUT: layout of the planes read / written -- 0 row-major, 1 blocked planes seen by rows, 2 blocked planes seen
    // by columns ("blocked layout" above); any of them non-zero: padded row numbering (a.rows_pad)
    constexpr bool BLK = (LIN != 0 || LOUT != 0);
    if (a.gate && *a.gate == 0) return;
    // TWICE: the two y passes of a solve in one kernel -- transform the rows, multiply by a.scale (the inverse eigenvalues),
    // transform them again, all in LDS: one launch, one write and one read of the planes less than two passes
    static_assert(N1 % 2 == 1 && N2 % 2 == 1, "symmetric kernel needs odd factors");
    constexpr int N = N1*N2, n = N - 1, T = DSTS_T, NT = DSTS_NT;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_double* cbuf = (lds_double*)lds_raw;          // [T][N] complex working set
    const double2* __restrict__ csa = a.fa;           // [H1][H1] (cos, sin)(2 pi n k / N1)
    const double2* __restrict__ csb = a.fb;           // [H2][H2] (cos, sin)(2 pi n k / N2)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int total_rows = a.rows_per_plane*a.nplanes;
    const int row0 = blockIdx.x*2*T;

    HPS_STAMP_DECL;
    HPS_STAMP(0);
    // a stage that packs its remaining items into one wave (sym_stage, LEFTOVER scheme) reads its table per lane: copy in LDS,
    // behind the working set (the launch's LDS has room for both tables: hps_poisson_create)
    constexpr int H1 = (N1 - 1)/2, H2 = (N2 - 1)/2;
    constexpr bool LEFT_A = sym_stage_leftover<N1, N2, T, NT/64>(), LEFT_B = sym_stage_leftover<N2, N1, T, NT/64>();
    lds_double* ltab_a = cbuf + 2*T*N;
    lds_double* ltab_b = ltab_a + 2*H1*H1;
    if constexpr (LEFT_A) { for (int k = tid; k < H1*H1; k += NT) { const double2 w = csa[k]; stc(ltab_a, k, w.x, w.y); } }
    if constexpr (LEFT_B) { for (int k = tid; k < H2*H2; k += NT) { const double2 w = csb[k]; stc(ltab_b, k, w.x, w.y); } }
    if constexpr (LIN == 1) load_blocked_rows<T, N, NT>(cbuf, a, row0, tid);
    else if constexpr (LIN == 2) load_blocked_cols<T, N, NT>(cbuf, a, row0, tid);
    else if constexpr (BLK) load_row_pairs_p<T, N, NT, SRC>(cbuf, a, row0, tid);
    else if (SRC) load_row_pairs_src<T, N, NT>(cbuf, a, row0, total_rows, tid);
    else load_row_pairs<T, N, NT>(cbuf, a, row0, total_rows, tid);
    __syncthreads();
    HPS_STAMP(1);
#pragma unroll
    for (int pass = 0; pass < (TWICE ? 2 : 1); ++pass) {
        {
            constexpr int PP = (N/2 + NT)/NT;
            double wr[T][PP], wi[T][PP], vr[T][PP], vi[T][PP];
            pre_to_regs<T, N, PP, NT>(cbuf, tid, wr, wi, vr, vi);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int m = 0; m < PP; ++m) {
                    const int p = tid + NT*m;
                    if (p <= N/2) {
                        stc(cbuf, t*N + p, wr[t][m], wi[t][m]);
                        if (p > 0) stc(cbuf, t*N + N - p, vr[t][m], vi[t][m]);
                    }
                }
            }
        }
        __syncthreads();
        HPS_STAMP(2);
        // stage A: DFT-N1 over n1 (stride N2) of column n2, twiddle w_N^(n2 k1), result at [k1][n2]
        sym_stage<N1, N2, N2, N2, 1, true, T, NT/64, true>(cbuf, csa, a.tw, wave, lane, ltab_a);
        HPS_STAMP(3);
        // stage B: DFT-N2 over n2 (stride 1) of row k1, result X[k1 + N1 k2] at [k1][k2]
        sym_stage<N2, 1, 1, N1, N2, false, T, NT/64, true>(cbuf, csb, nullptr, wave, lane, ltab_b);
        HPS_STAMP(4);
        if (TWICE && pass == 0) {
            // T_k of both rows of every pair (times the scale) -> registers -> back to [t][k] as the next pass's input
            constexpr int NK = (n + NT - 1)/NT;
            double ta[T][NK], tb[T][NK];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                int ja, jb;
                if constexpr (BLK) {
                    const int j0 = row0 - (row0 / a.rows_pad)*a.rows_pad;
                    ja = min(j0 + 2*t, a.rows_per_plane - 1); jb = min(j0 + 2*t + 1, a.rows_per_plane - 1);
                } else {
                    const int ra = min(row0 + 2*t, total_rows - 1), rb = min(row0 + 2*t + 1, total_rows - 1);
                    ja = ra - (ra / a.rows_per_plane)*a.rows_per_plane; jb = rb - (rb / a.rows_per_plane)*a.rows_per_plane;
                }
#pragma unroll
                for (int m = 0; m < NK; ++m) {
                    const int k = min(tid + NT*m, n - 1);
                    const int q1 = k + 1, q2 = N - 1 - k;
                    const double2 x1 = ldc(cbuf, t*N + (q1 % N1)*N2 + q1/N1);
                    const double2 x2 = ldc(cbuf, t*N + (q2 % N1)*N2 + q2/N1);
                    const double is = a.isin4[k];
                    ta[t][m] = (0.5*(x2.x - x1.x) + (x1.x + x2.x)*is)*a.scale[(long)ja*n + k];
                    tb[t][m] = (0.5*(x2.y - x1.y) + (x1.y + x2.y)*is)*a.scale[(long)jb*n + k];
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int m = 0; m < NK; ++m) {
                    const int k = tid + NT*m;
                    if (k < n) stc(cbuf, t*N + k, ta[t][m], tb[t][m]);
                }
            }
            __syncthreads();
        }
    }
    if constexpr (LOUT == 1) store_blocked_rows<T, N1, N2, NT>(cbuf, a, row0, tid);
    else if constexpr (LOUT == 2) store_blocked_cols<T, N1, N2, NT>(cbuf, a, row0, tid);
    else if constexpr (BLK) post_store_p<T, N1, N2, NT>(cbuf, a, row0, tid);
    else post_store<T, N1, N2, NT, TWICE>(cbuf, a, row0, total_rows, tid);
    HPS_STAMP(5);
    HPS_STAMP_FLUSH;
}

// ---- the small-DFT stages as fp64 MFMA contractions --------------------------------------------
// The folded DFT of the symmetric kernel is a real matrix product: with s_n = x_n + x_{M-n}, d_n = x_n - x_{M-n},
//   P[k][col] = sum_n cos(2 pi n k / M) s[n][col]      (row k = 0: all ones -> sum_n s_n)
//   Q[k][col] = sum_n sin(2 pi n k / M) d[n][col]
// over k = 0..H, n = 1..H and the columns col = 2*item + (re | im) of all items of the workgroup.  One wave owns a
// tile of 16 columns (8 items) and both k tiles: v_mfma_f64_16x16x4_f64, the constant matrices as A operands in
// registers (host tables in the instruction's lane layout: lane l holds A[l % 16][l / 16]), s / d as B operands
// built from two LDS reads per lane and k step (lane l: n = 1 + 4*ks + l / 16, column l % 16); the accumulator
// holds D[4 r + l / 16][l % 16] in register r.  No LDS table traffic, no per-step load -> use round trip.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

template <int M> struct MfmaDims { static constexpr int H = (M - 1)/2, MT = (H + 1 + 15)/16, KS = (H + 3)/4; };

// the constant A operands of a stage (requested at kernel start: an L2 / HBM round trip that must not sit at the
// head of the stage)
template <int M>
__device__ __forceinline__ void mfma_load_tables (const double* __restrict__ tab, int lane,
                                                  double (&ac)[MfmaDims<M>::MT][MfmaDims<M>::KS], double (&as)[MfmaDims<M>::MT][MfmaDims<M>::KS])
{
    constexpr int MT = MfmaDims<M>::MT, KS = MfmaDims<M>::KS;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ac[mt][ks] = tab[(mt*KS + ks)*64 + lane];
            as[mt][ks] = tab[((MT + mt)*KS + ks)*64 + lane];
        }
    }
}

template <int M, int STRIDE, int ITEMS_PER_T, int ITEM_STRIDE, bool TWIDDLE, int T, int NWAVES>
__device__ __forceinline__ void mfma_stage (lds_double* cbuf, const double (&ac)[MfmaDims<M>::MT][MfmaDims<M>::KS],
                                            const double (&as)[MfmaDims<M>::MT][MfmaDims<M>::KS],
                                            const double2* __restrict__ tw, int wave, int lane)
{
    constexpr int H = MfmaDims<M>::H, MT = MfmaDims<M>::MT, KS = MfmaDims<M>::KS;
    constexpr int ITEMS = T*ITEMS_PER_T, NCOL = 2*ITEMS, NTILES = (NCOL + 15)/16;
    constexpr int NTE = M*ITEMS_PER_T;                // complex elements per transform
    const int jl = lane & 15, nq = lane >> 4;
    for (int nt = wave; nt < NTILES; nt += NWAVES) {
        const int col = nt*16 + jl;
        const bool cok = col < NCOL;
        const int item = min(col, NCOL - 1) >> 1, c = col & 1;
        const int t = item / ITEMS_PER_T, r = item - t*ITEMS_PER_T;
        const int base = t*NTE + r*ITEM_STRIDE;
        mfma_d4 Pa[MT], Qa[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { Pa[mt] = mfma_d4{0.0, 0.0, 0.0, 0.0}; Qa[mt] = mfma_d4{0.0, 0.0, 0.0, 0.0}; }
        // the twiddles of this lane's outputs are requested now: their (L2) latency hides behind the MFMAs
        double2 w1[MT][4], w2[MT][4];
        if (TWIDDLE) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int kc = min(mt*16 + 4*rr + nq, H);
                    w1[mt][rr] = tw[kc*ITEMS_PER_T + r];
                    w2[mt][rr] = tw[(M - max(kc, 1))*ITEMS_PER_T + r];
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int n = 1 + ks*4 + nq;
            const int nn = min(n, H);
            const double xa = cbuf[2*(base + nn*STRIDE) + c], xb = cbuf[2*(base + (M - nn)*STRIDE) + c];
            const bool ok = cok && (n <= H);
            const double sv = ok ? xa + xb : 0.0, dv = ok ? xa - xb : 0.0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                Pa[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[mt][ks], sv, Pa[mt], 0, 0, 0);
                Qa[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[mt][ks], dv, Qa[mt], 0, 0, 0);
            }
        }
        const double x0c = cbuf[2*base + c];
        // X_k = x0 + P_k + i Q_k, X_{M-k} = x0 + P_k - i Q_k: the other component of Q sits in the neighbouring lane
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int k = mt*16 + 4*rr + nq;
                const double pv = Pa[mt][rr], qv = Qa[mt][rr];
                const double qo = __shfl_xor(qv, 1);
                double vk = x0c + pv + (c ? qo : -qo);        // re: P_re - Q_im; im: P_im + Q_re
                double vm = x0c + pv - (c ? qo : -qo);
                if (TWIDDLE) {
                    const double ok_ = __shfl_xor(vk, 1), om_ = __shfl_xor(vm, 1);
                    const double2 wa = w1[mt][rr], wb = w2[mt][rr];
                    const double re1 = c ? ok_ : vk, im1 = c ? vk : ok_;
                    const double re2 = c ? om_ : vm, im2 = c ? vm : om_;
                    vk = c ? re1*wa.y + im1*wa.x : re1*wa.x - im1*wa.y;
                    vm = c ? re2*wb.y + im2*wb.x : re2*wb.x - im2*wb.y;
                }
                if (cok && k <= H) {
                    if (k == 0) cbuf[2*base + c] = x0c + pv;                     // twiddle 1
                    else { cbuf[2*(base + k*STRIDE) + c] = vk; cbuf[2*(base + (M - k)*STRIDE) + c] = vm; }
                }
            }
        }
    }
    __syncthreads();
}

template <int N1, int N2>
__global__ __launch_bounds__(DSTS_NT)
void k_dst_rows_mfma (DstArgs a)
{
    static_assert(N1 % 2 == 1 && N2 % 2 == 1, "symmetric kernel needs odd factors");
    constexpr int N = N1*N2, T = DSTS_T, NT = DSTS_NT;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_double* cbuf = (lds_double*)lds_raw;          // [T][N] complex working set
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int total_rows = a.rows_per_plane*a.nplanes;
    const int row0 = blockIdx.x*2*T;
    HPS_STAMP_DECL;
    HPS_STAMP(0);
    double ca[MfmaDims<N1>::MT][MfmaDims<N1>::KS], sa[MfmaDims<N1>::MT][MfmaDims<N1>::KS];
    double cb[MfmaDims<N2>::MT][MfmaDims<N2>::KS], sb[MfmaDims<N2>::MT][MfmaDims<N2>::KS];
    mfma_load_tables<N1>(a.ma, lane, ca, sa);
    mfma_load_tables<N2>(a.mb, lane, cb, sb);
    load_row_pairs<T, N, NT>(cbuf, a, row0, total_rows, tid);
    __syncthreads();
    HPS_STAMP(1);
    {
        constexpr int PP = (N/2 + NT)/NT;
        double wr[T][PP], wi[T][PP], vr[T][PP], vi[T][PP];
        pre_to_regs<T, N, PP, NT>(cbuf, tid, wr, wi, vr, vi);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int m = 0; m < PP; ++m) {
                const int p = tid + NT*m;
                if (p <= N/2) {
                    stc(cbuf, t*N + p, wr[t][m], wi[t][m]);
                    if (p > 0) stc(cbuf, t*N + N - p, vr[t][m], vi[t][m]);
                }
            }
        }
    }
    __syncthreads();
    HPS_STAMP(2);
    // stage A: DFT-N1 over n1 (stride N2) of column n2, twiddle w_N^(n2 k1), result at [k1][n2]
    mfma_stage<N1, N2, N2, 1, true, T, NT/64>(cbuf, ca, sa, a.tw, wave, lane);
    HPS_STAMP(3);
    // stage B: DFT-N2 over n2 (stride 1) of row k1, result X[k1 + N1 k2] at [k1][k2]
    mfma_stage<N2, 1, N1, N2, false, T, NT/64>(cbuf, cb, sb, nullptr, wave, lane);
    HPS_STAMP(4);
    post_store<T, N1, N2, NT>(cbuf, a, row0, total_rows, tid);
    HPS_STAMP(5);
    HPS_STAMP_FLUSH;
}

// ---- y direction on column blocks, in place -----------------------------------------------------
// One workgroup owns 2*CT adjacent columns of one plane (64-byte row segments for CT = 4): it
// transforms them along y, multiplies by the inverse eigenvalues and transforms back, all in LDS --
// the two transposes and one round trip through HBM of the row-wise formulation disappear.
// a.src/a.dst: planes of `rows_per_plane` rows (y) with pitches src_pitch / dst_pitch; a.scale =
// [nx][n] inverse eigenvalues (row = column index kx); N = ny + 1 = N1*N2.
#ifndef HPS_DSTC_T
#define HPS_DSTC_T 4
#endif
constexpr int DSTC_T = HPS_DSTC_T;
template <int N1, int N2>
__global__ __launch_bounds__(DSTS_NT)
void k_dst_cols_sym (DstArgs a, int ncols)
{
    static_assert(N1 % 2 == 1 && N2 % 2 == 1, "symmetric kernel needs odd factors");
    constexpr int N = N1*N2, n = N - 1, T = DSTC_T, NT = DSTS_NT, CB = 2*T;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_double* cbuf = (lds_double*)lds_raw;          // [T][N] complex: (column 2t, column 2t+1) at row j
    const double2* __restrict__ csa = a.fa;
    const double2* __restrict__ csb = a.fb;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nblk = (ncols + CB - 1)/CB;
    const int plane = blockIdx.x / nblk, c0 = (blockIdx.x - plane*nblk)*CB;
    const double* __restrict__ src = a.src[plane];
    double* __restrict__ dst = a.dst[plane];

    // rows j of the column block, CB doubles per row
    for (int e = tid; e < n*CB; e += NT) {
        const int j = e / CB, c = e - j*CB;
        const double v = (c0 + c < ncols) ? src[(long)j*a.src_pitch + c0 + c] : 0.0;
        cbuf[2*((c >> 1)*N + j) + (c & 1)] = v;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        {
            constexpr int PP = (N/2 + NT)/NT;
            double wr[T][PP], wi[T][PP], vr[T][PP], vi[T][PP];
            pre_to_regs<T, N, PP, NT>(cbuf, tid, wr, wi, vr, vi);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int m = 0; m < PP; ++m) {
                    const int p = tid + NT*m;
                    if (p <= N/2) {
                        stc(cbuf, t*N + p, wr[t][m], wi[t][m]);
                        if (p > 0) stc(cbuf, t*N + N - p, vr[t][m], vi[t][m]);
                    }
                }
            }
        }
        __syncthreads();
        sym_stage<N1, N2, N2, N2, 1, true, T, NT/64>(cbuf, csa, a.tw, wave, lane);
        sym_stage<N2, 1, 1, N1, N2, false, T, NT/64>(cbuf, csb, nullptr, wave, lane);
        {
            // T_k of both columns of every pair -> registers -> back to [t][k]
            constexpr int KP = (n + NT - 1)/NT;
            double ta[T][KP], tb[T][KP];
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int m = 0; m < KP; ++m) {
                    const int k = tid + NT*m;
                    ta[t][m] = tb[t][m] = 0.0;
                    if (k < n) {
                        const int q1 = k + 1, q2 = N - 1 - k;
                        const double2 x1 = ldc(cbuf, t*N + (q1 % N1)*N2 + q1/N1);
                        const double2 x2 = ldc(cbuf, t*N + (q2 % N1)*N2 + q2/N1);
                        const double is = a.isin4[k];
                        double va = 0.5*(x2.x - x1.x) + (x1.x + x2.x)*is;
                        double vb = 0.5*(x2.y - x1.y) + (x1.y + x2.y)*is;
                        if (pass == 0) {
                            const int ca = min(c0 + 2*t, ncols - 1), cb = min(c0 + 2*t + 1, ncols - 1);
                            va *= a.scale[(long)ca*n + k];
                            vb *= a.scale[(long)cb*n + k];
                        }
                        ta[t][m] = va; tb[t][m] = vb;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int m = 0; m < KP; ++m) {
                    const int k = tid + NT*m;
                    if (k < n) stc(cbuf, t*N + k, ta[t][m], tb[t][m]);
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < n*CB; e += NT) {
        const int j = e / CB, c = e - j*CB;
        if (c0 + c < ncols) dst[(long)j*a.dst_pitch + c0 + c] = cbuf[2*((c >> 1)*N + j) + (c & 1)];
    }
}

// plane-wise transpose: dst[k][j] = src[j][k], src has `rows` rows of `cols` entries
__global__ __launch_bounds__(256)
void k_transpose (const double* __restrict__ src, double* __restrict__ dst, int rows, int cols, long plane_stride)
{
    __shared__ double tile[32][33];
    const double* s = src + blockIdx.z*plane_stride;
    double* d = dst + blockIdx.z*plane_stride;
    const int c0 = blockIdx.x*32, r0 = blockIdx.y*32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int rr = ty; rr < 32; rr += 8)
        if (r0 + rr < rows && c0 + tx < cols) tile[rr][tx] = s[(long)(r0 + rr)*cols + c0 + tx];
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8)
        if (c0 + cc < cols && r0 + tx < rows) d[(long)(c0 + cc)*rows + r0 + tx] = tile[tx][cc];
}

// =================================================================================================
// y direction as tridiagonal solves (round 6; HPS_POISSON_TRIDIAG, default on)
// After the DST along x every x mode k is an independent system along y,
//        u[j-1] + b_k u[j] + u[j+1] = dy^2 f[j],    b_k = -2 - 4 sin^2(pi (k+1) / (2 (nx+1))) dy^2/dx^2,   u[-1] = u[ny] = 0
// -- the 5-point operator whose eigenvalues FFTPoissonSolverDirichletDirect.cpp:58-83 divides by, so DST_y . 1/eig . DST_y is
// exactly this solve, and two of the four transform passes of a solve (and both transposes / the blocked planes) are gone.
// The matrix is constant, so its LU factors c'_j(k) = 1/(b_k - c'_{j-1}) are a table made once (hps_poisson_create):
//        forward   d_j = c'_j (alpha f_j - d_{j-1}),     backward   u_j = d_j - c'_j u_{j+1},     alpha = dy^2 / (2 (nx+1))
// (alpha carries the normalisation of the two x transforms).  Work split: a thread owns M consecutive rows of one column
// (registers), a workgroup COLS columns x all segments, so for a fixed row the lanes read COLS consecutive doubles of a
// row-major plane -- whole 128-byte lines for COLS = 16, no LDS staging of the data.  Both sweeps are first-order linear
// recurrences: a thread runs its segment with zero inflow, the (multiplier, value) pairs of the segments are chained through
// LDS by one lane per column (nseg <= 64 dependent FMAs), and the inflow's multiple is added back -- |c'| < 1, every
// multiplier is a product of them, so nothing grows.
// =================================================================================================
struct TriArgs {
    double* plane[DST_MAXPLANES]; long pitch;        // row-major [j][k] planes, solved in place
    const double* cp;                                // [ny][pitch] LU factors c'_j(k)
    double alpha;
    int nx, ny, nseg;
    const int* gate;
};

template <int M, int COLS>
__global__ __launch_bounds__(COLS*64)
void k_tridiag_y (TriArgs a)
{
    if (a.gate && *a.gate == 0) return;
    __shared__ double sA[64][COLS], sB[64][COLS], sD[64][COLS];
    const int tid = threadIdx.x, kk = tid % COLS, s = tid / COLS;
    const int k = blockIdx.x*COLS + kk, kc = min(k, a.nx - 1);
    double* __restrict__ p = a.plane[blockIdx.y];
    const int j0 = s*M;
    double e[M], c[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const int jc = min(j0 + i, a.ny - 1);
        e[i] = p[(long)jc*a.pitch + kc];
        c[i] = a.cp[(long)jc*a.pitch + kc];
    }
    // rows past the plane: identity rows (c' = 0, f = 0)
#pragma unroll
    for (int i = 0; i < M; ++i) if (j0 + i >= a.ny) { e[i] = 0.0; c[i] = 0.0; }
    // forward sweep of the segment, zero inflow
    {
        double d = 0.0, pm = 1.0;
#pragma unroll
        for (int i = 0; i < M; ++i) { d = c[i]*fma(a.alpha, e[i], -d); e[i] = d; pm *= -c[i]; }
        sA[s][kk] = pm; sB[s][kk] = d;
    }
    __syncthreads();
    if (tid < COLS) {
        double D = 0.0;
        for (int q = 0; q < a.nseg; ++q) { const double A = sA[q][kk], B = sB[q][kk]; sD[q][kk] = D; D = fma(A, D, B); }
    }
    __syncthreads();
    // true d_j = e_j + (prod_{q <= j} -c'_q) D; backward sweep of the segment, zero inflow
    {
        const double D = sD[s][kk];
        double pm = 1.0;
#pragma unroll
        for (int i = 0; i < M; ++i) { pm *= -c[i]; e[i] = fma(pm, D, e[i]); }
        double h = 0.0, qm = 1.0;
#pragma unroll
        for (int i = M - 1; i >= 0; --i) { h = fma(-c[i], h, e[i]); e[i] = h; qm *= -c[i]; }
        sA[s][kk] = qm; sB[s][kk] = h;         // (the forward chain's reads of sA / sB lie behind the barrier above)
    }
    __syncthreads();
    if (tid < COLS) {
        double U = 0.0;
        for (int q = a.nseg - 1; q >= 0; --q) { const double A = sA[q][kk], B = sB[q][kk]; sD[q][kk] = U; U = fma(A, U, B); }
    }
    __syncthreads();
    {
        const double U = sD[s][kk];
        double qm = 1.0;
#pragma unroll
        for (int i = M - 1; i >= 0; --i) { qm *= -c[i]; e[i] = fma(qm, U, e[i]); }
    }
    if (k < a.nx) {
#pragma unroll
        for (int i = 0; i < M; ++i) if (j0 + i < a.ny) p[(long)(j0 + i)*a.pitch + k] = e[i];
    }
}
typedef void (*tri_kernel_t)(TriArgs);
struct TriImpl { int M, cols; tri_kernel_t kernel; };
// rows per thread by plane height: at most 64 segments per column
static TriImpl find_tri_impl (int ny)
{
    // (A/B: HPS_TRI_M, HPS_TRI_COLS pick another instantiated shape that still covers the plane with <= 64 segments)
    {   const char* vm = getenv("HPS_TRI_M"); const char* vc = getenv("HPS_TRI_COLS");
        if (vm || vc) {
            const int m = vm ? atoi(vm) : 16, c = vc ? atoi(vc) : 16;
            static const TriImpl all[] = {{4, 16, k_tridiag_y<4, 16>}, {8, 16, k_tridiag_y<8, 16>}, {16, 16, k_tridiag_y<16, 16>}, {32, 8, k_tridiag_y<32, 8>},
                                          {16, 8, k_tridiag_y<16, 8>}, {16, 4, k_tridiag_y<16, 4>}, {8, 8, k_tridiag_y<8, 8>}, {32, 4, k_tridiag_y<32, 4>}};
            for (const TriImpl& t : all) if (t.M == m && t.cols == c && (long)m*64 >= ny) return t;
        } }
    if (ny <= 256) return TriImpl{4, 16, k_tridiag_y<4, 16>};
    if (ny <= 512) return TriImpl{8, 16, k_tridiag_y<8, 16>};
    if (ny <= 1024) return TriImpl{16, 16, k_tridiag_y<16, 16>};
    if (ny <= 2048) return TriImpl{32, 8, k_tridiag_y<32, 8>};
    return TriImpl{0, 0, nullptr};
}

// =================================================================================================
// dense back-end: n + 1 without a built factorisation (prime 257 of the 256^2 decks, ...), n <= 512.
// DST-I as a matrix product with S[j][k] = 2 sin(pi (j+1)(k+1)/(n+1)) (FFTW's RODFT00 scaling, as the row kernels):
// x direction X.S_x, y direction S_y.X -- four products per solve, no transposes.  One workgroup per 32 x 32 tile of
// the product, operands staged through LDS in 32-deep slabs, four waves on fp64 MFMA (a real contraction: 2 n^3 flops per
// product; a VALU version with 2 x 2 outputs per lane was LDS-bandwidth bound at 14 us per product).
// =================================================================================================
struct GemmArgs {
    const double* A[DST_MAXPLANES]; long lda;
    const double* B[DST_MAXPLANES]; long ldb;
    double* C[DST_MAXPLANES]; long ldc;
    int M, N, K;
    const double* scale; long scale_r, scale_c;      // optional factor scale[r*scale_r + c*scale_c] on the output
    const int* gate;                                 // optional device word: return at once when *gate == 0
};

// The operands of DEPTH 32-deep slabs are in flight at once (registers), refilled as slabs are consumed; LDS slabs are
// double-buffered (one barrier per slab), two accumulator chains.  Measured on config 2 (bench.py --config2, two runs each):
// depth 1 2894-2924 slices/s, 2 2999-3016, 4 2943-3010, 8 (all slabs of a 256-deep product, 252 VGPRs) 2970-2973: the
// product is not waiting for its operands; 2 it is.
#ifndef HPS_DENSE_DEPTH
#define HPS_DENSE_DEPTH 2
#endif
__global__ __launch_bounds__(256)
void k_dense_product (GemmArgs g)
{
    if (g.gate && *g.gate == 0) return;
    constexpr int D = HPS_DENSE_DEPTH;
    // pitches chosen so that the 32 lanes of one LDS pass hit 32 different 8-byte slots: A rows 34 apart, B rows 48
    __shared__ double As[2][32][34];
    __shared__ double Bs[2][32][48];
    const int pl = blockIdx.z;
    const double* __restrict__ A = g.A[pl];
    const double* __restrict__ B = g.B[pl];
    const int r0 = blockIdx.y*32, c0 = blockIdx.x*32;
    // four waves, one 16 x 16 block of the tile each, on v_mfma_f64_16x16x4_f64: lane l feeds A[l%16][l/16] and
    // B[l/16][l%16] of the 4-deep step and holds D[4r + l/16][l%16], r = 0..3
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wr = (wave >> 1)*16, wc = (wave & 1)*16;
    const int lm = lane & 15, lk = lane >> 4;
    mfma_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};      // two chains: an MFMA does not wait for the one before it
    double ra[D][4], rb[D][4];
    auto fetch = [&] (int d, int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = threadIdx.x + 256*q, r = e >> 5, k = e & 31;
            ra[d][q] = (r0 + r < g.M && k0 + k < g.K) ? A[(long)(r0 + r)*g.lda + k0 + k] : 0.0;
            rb[d][q] = (k0 + r < g.K && c0 + k < g.N) ? B[(long)(k0 + r)*g.ldb + c0 + k] : 0.0;
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) if (32*d < g.K) fetch(d, 32*d);
    for (int kb = 0; kb < g.K; kb += 32*D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int k0 = kb + 32*d;
            if (k0 < g.K) {                      // (uniform)
                const int buf = d & 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = threadIdx.x + 256*q, r = e >> 5, k = e & 31;
                    As[buf][r][k] = ra[d][q]; Bs[buf][r][k] = rb[d][q];
                }
                __syncthreads();
                if (k0 + 32*D < g.K) fetch(d, k0 + 32*D);
#pragma unroll
                for (int kk = 0; kk < 32; kk += 8) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(As[buf][wr + lm][kk + lk], Bs[buf][kk + lk][wc + lm], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(As[buf][wr + lm][kk + 4 + lk], Bs[buf][kk + 4 + lk][wc + lm], acc1, 0, 0, 0);
                }
            }
        }
    }
    double* __restrict__ C = g.C[pl];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + wr + 4*q + lk, c = c0 + wc + lm;
        if (r < g.M && c < g.N) {
            double v = acc0[q] + acc1[q];
            if (g.scale) v *= g.scale[(long)r*g.scale_r + (long)c*g.scale_c];
            C[(long)r*g.ldc + c] = v;
        }
    }
}

#ifndef HPS_POISSON_POW2
#define HPS_POISSON_POW2 1
#endif
typedef void (*dst_kernel_t)(DstArgs);
typedef void (*dst_cols_kernel_t)(DstArgs, int);
struct DstImpl { int N, N1, N2; dst_kernel_t kernel; bool sym; int T; int nt; dst_cols_kernel_t cols; dst_kernel_t mfma; dst_kernel_t src; dst_kernel_t twice;
                 // the passes of a solve on blocked intermediate planes (no transposes): first pass from row-major rows / from
                 // rows formed out of other planes, both y passes in place on the blocked planes, last pass to row-major rows
                 dst_kernel_t b_first, b_first_src, b_twice, b_last;
                 bool pow2 = false; };           // N a power of two: k_dst_rows_pow2 (tables: w_N^k only)

#define HPS_DST_POW2(LOGN) DstImpl{1 << (LOGN), 1 << (LOGN), 1, k_dst_rows_pow2<LOGN>, false, DSTP_T, DSTP_NT, nullptr, nullptr, k_dst_rows_pow2<LOGN, true>, k_dst_rows_pow2<LOGN, false, true>, nullptr, nullptr, nullptr, nullptr, true}
#define HPS_DST_IMPL(N1, N2) DstImpl{(N1)*(N2), N1, N2, k_dst_rows<N1, N2>, false, DST_T, 256, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}
#define HPS_DST_SYM(N1, N2) DstImpl{(N1)*(N2), N1, N2, k_dst_rows_sym<N1, N2>, true, DSTS_T, DSTS_NT, k_dst_cols_sym<N1, N2>, k_dst_rows_mfma<N1, N2>, k_dst_rows_sym<N1, N2, true>, k_dst_rows_sym<N1, N2, false, true>, \
                                    k_dst_rows_sym<N1, N2, false, false, 0, 1>, k_dst_rows_sym<N1, N2, true, false, 0, 1>, k_dst_rows_sym<N1, N2, false, true, 2, 2>, k_dst_rows_sym<N1, N2, false, false, 1, 0>}
static const DstImpl g_dst_impls[] = {
    HPS_DST_SYM(25, 41),    // nx = 1024
    HPS_DST_SYM(19, 27),    // 512
    HPS_DST_SYM(3, 43),     // 128
    HPS_DST_SYM(5, 13),     // 64
    HPS_DST_SYM(3, 11),     // 32
#if HPS_POISSON_POW2
    HPS_DST_POW2(11),       // 2047
    HPS_DST_POW2(10),       // 1023
    HPS_DST_POW2(9),        // 511
    HPS_DST_POW2(8),        // 255
    HPS_DST_POW2(7),        // 127
    HPS_DST_POW2(6),        // 63
#else
    HPS_DST_IMPL(32, 32),   // 1023
    HPS_DST_IMPL(16, 32),   // 511
    HPS_DST_IMPL(16, 16),   // 255
    HPS_DST_IMPL(8, 16),    // 127
    HPS_DST_IMPL(8, 8),     // 63
#endif
    HPS_DST_SYM(9, 11),     // 98
    HPS_DST_SYM(7, 11),     // 76
};

static const DstImpl* find_dst_impl (int N)
{
    for (const auto& d : g_dst_impls) if (d.N == N) return &d;
    return nullptr;
}

// =================================================================================================
// rocFFT back-end kernels (any N)
// =================================================================================================
__global__ __launch_bounds__(256)
void k_pre_rows (const double* __restrict__ src, long src_pitch, double2* __restrict__ z, int n, int nh, int nrows)
{
    const int p = blockIdx.x*blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (p >= nh || row >= nrows) return;
    const double* x = src + (long)row*src_pitch;
    const int N = n + 1;
    auto get = [&] (int j) { return x[j]; };
    const double re = odd_ext(2*p + 1, N, get) - odd_ext(2*p - 1, N, get);
    const double im = odd_ext(2*p, N, get);
    z[(long)row*nh + p] = make_double2(re, im);
}

constexpr int TP_K = 32;      // k-extent of a tile
constexpr int TP_P = 16;      // p-extent of a tile -> needs 2*TP_P + 2 values of j
constexpr int TP_J = 2*TP_P + 2;

__global__ __launch_bounds__(256)
void k_post_transpose_pre (const double* __restrict__ r, int n_in, int nrows_in,
                           const double* __restrict__ isin4, double2* __restrict__ z, int nh_out)
{
    __shared__ double tile[TP_J][TP_K + 1];
    const int N_in = n_in + 1;
    const int N_out = nrows_in + 1;
    const int k0 = blockIdx.x*TP_K;
    const int p0 = blockIdx.y*TP_P;
    const int j0 = 2*p0 - 2;
    {
        const int kk = threadIdx.x % TP_K;
        const int jr = threadIdx.x / TP_K;
        const int k = k0 + kk;
        for (int jj = jr; jj < TP_J; jj += 256/TP_K) {
            const int j = j0 + jj;
            double v = 0.0;
            if (k < n_in && j >= 0 && j < nrows_in) v = dst_from_r(r + (long)j*N_in, k, N_in, isin4[k]);
            tile[jj][kk] = v;
        }
    }
    __syncthreads();
    {
        const int pp = threadIdx.x % TP_P;
        const int kr = threadIdx.x / TP_P;
        const int p = p0 + pp;
        for (int kk = kr; kk < TP_K; kk += 256/TP_P) {
            const int k = k0 + kk;
            if (k < n_in && p < nh_out) {
                auto get = [&] (int j) { return tile[j - j0][kk]; };
                const double re = odd_ext(2*p + 1, N_out, get) - odd_ext(2*p - 1, N_out, get);
                const double im = odd_ext(2*p, N_out, get);
                z[(long)k*nh_out + p] = make_double2(re, im);
            }
        }
    }
}

__global__ __launch_bounds__(256)
void k_post_mult_pre (const double* __restrict__ r, int n, int nrows, const double* __restrict__ isin4,
                      const double* __restrict__ eig, double2* __restrict__ z, int nh)
{
    const int p = blockIdx.x*blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (p >= nh || row >= nrows) return;
    const int N = n + 1;
    const double* rr = r + (long)row*N;
    const double* ee = eig + (long)row*n;
    auto get = [&] (int l) { return ee[l]*dst_from_r(rr, l, N, isin4[l]); };
    const double re = odd_ext(2*p + 1, N, get) - odd_ext(2*p - 1, N, get);
    const double im = odd_ext(2*p, N, get);
    z[(long)row*nh + p] = make_double2(re, im);
}

__global__ __launch_bounds__(256)
void k_post_to_slab (const double* __restrict__ r, int n, int nrows, const double* __restrict__ isin4,
                     double* __restrict__ dst, long dst_pitch)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (k >= n || row >= nrows) return;
    dst[(long)row*dst_pitch + k] = dst_from_r(r + (long)row*(n + 1), k, n + 1, isin4[k]);
}

// =================================================================================================
struct Poisson {
    int nx = 0, ny = 0;
    // own-transform back-end
    dst_kernel_t kx = nullptr, ky = nullptr;
    dst_kernel_t kx_src = nullptr;      // the x pass with its rows formed from other planes (sym kernels only)
    dst_kernel_t ky2 = nullptr;         // both y passes (transform, inverse eigenvalues, transform) in one launch (sym kernels; HPS_POISSON_Y2=0: off)
    dst_cols_kernel_t kcols = nullptr; size_t lds_cols = 0;     // y direction on column blocks (symmetric factorisations)
    // blocked intermediate planes (HPS_POISSON_BLOCKED, default on; 0: row-major planes and two transposes): three launches per solve
    dst_kernel_t kb_first = nullptr, kb_first_src = nullptr, kb_twice = nullptr, kb_last = nullptr;
    long blk_plane = 0;                 // doubles per blocked plane: ny rounded up to whole row blocks, times nx
    bool blocked () const { return kb_first != nullptr; }
    double2 *tab_x = nullptr, *tab_y = nullptr;        // each: [fa | fb | tw] concatenated
    double *mtab_x = nullptr, *mtab_y = nullptr;       // MFMA operand tables [stage A | stage B] (k_dst_rows_mfma)
    const double *ma_x = nullptr, *mb_x = nullptr, *ma_y = nullptr, *mb_y = nullptr;
    const double2 *fa_x = nullptr, *fb_x = nullptr, *tw_x = nullptr, *fa_y = nullptr, *fb_y = nullptr, *tw_y = nullptr;
    double *buf_a = nullptr, *buf_b = nullptr;         // [DST_MAXPLANES][nx*ny] ping-pong
    double *S_x = nullptr, *S_y = nullptr;             // dense back-end: [n][n] sine matrices (S_y = S_x if nx == ny)
    // y direction as tridiagonal solves (k_tridiag_y; HPS_POISSON_TRIDIAG=0: off): needs a transform along x only
    tri_kernel_t ktri = nullptr; int tri_M = 0, tri_cols = 0; double* tri_cp = nullptr; double tri_alpha = 0.0;
    long pa = 0;                        // row pitch of the intermediate planes between the x passes (own kernels: nx rounded up to whole 128-byte lines; dense: nx)
    bool tri () const { return ktri != nullptr; }
    long long* dbg = nullptr;
    const int* gate = nullptr;          // poisson_set_gate: the launches of the following solves return at once when *gate == 0
    bool gate_ok = false;               // every kernel of the own back-end as configured looks at the gate (the transposes only touch scratch)
    size_t lds_x = 0, lds_y = 0; int tx = DST_T, ty = DST_T, ntx = 256, nty = 256;     // LDS bytes, row pairs and threads per workgroup
    // rocFFT back-end
    rocfft_plan plan_x = nullptr, plan_y = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr; size_t work_bytes = 0;
    double2* zbuf = nullptr; double* rbuf = nullptr;
    hipStream_t bound_stream = nullptr; bool stream_bound = false;
    // shared
    double* eig = nullptr; double* isin_x = nullptr; double* isin_y = nullptr;

    bool own () const { return kx && (ky || ktri); }
    bool dense () const { return S_x != nullptr; }
    ~Poisson () {
        if (plan_x) rocfft_plan_destroy(plan_x);
        if (plan_y) rocfft_plan_destroy(plan_y);
        if (info) rocfft_execution_info_destroy(info);
        (void)hipFree(work); (void)hipFree(zbuf); (void)hipFree(rbuf); (void)hipFree(eig);
        (void)hipFree(isin_x); (void)hipFree(isin_y); (void)hipFree(tab_x); (void)hipFree(tab_y); (void)hipFree(mtab_x); (void)hipFree(mtab_y);
        (void)hipFree(buf_a); (void)hipFree(buf_b); (void)hipFree(tri_cp);
        if (S_y != S_x) (void)hipFree(S_y);
        (void)hipFree(S_x);
    }
};

static bool g_rocfft_setup = false;

static int make_plan (rocfft_plan* plan, int N, int batch)
{
    const size_t len[1] = {(size_t)N};
    if (rocfft_plan_create(plan, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                           rocfft_precision_double, 1, len, (size_t)batch, nullptr) != rocfft_status_success) {
        set_error("rocfft_plan_create failed for length " + std::to_string(N));
        return HPS_ERR_FFT;
    }
    return HPS_OK;
}

// DFT matrices of both stages and the inter-stage twiddles, concatenated [fa | fb | tw].
// sym: only the (cos, sin) of n, k = 1..(M-1)/2 are stored.
static int upload_tables (int N1, int N2, bool sym, double2** out, size_t* na, size_t* nb, bool pow2 = false)
{
    const int N = N1*N2;
    if (pow2) {                                   // w_N^k, k < N, where the other factorisations keep their [N1][N2] twiddles
        *na = *nb = 0;
        std::vector<double2> h((size_t)N);
        const long double pi2 = 6.283185307179586476925286766559L;
        for (int k = 0; k < N; ++k) { const long double ang = pi2*k/N; h[(size_t)k] = make_double2((double)cosl(ang), (double)sinl(ang)); }
        HPS_HIP_CHECK(hipMalloc(out, h.size()*sizeof(double2)));
        HPS_HIP_CHECK(hipMemcpy(*out, h.data(), h.size()*sizeof(double2), hipMemcpyHostToDevice));
        return HPS_OK;
    }
    const int a0 = sym ? 1 : 0, a1 = sym ? (N1 - 1)/2 : N1 - 1, b1 = sym ? (N2 - 1)/2 : N2 - 1;
    *na = (size_t)(a1 - a0 + 1)*(a1 - a0 + 1); *nb = (size_t)(b1 - a0 + 1)*(b1 - a0 + 1);
    std::vector<double2> h(*na + *nb + N);
    const long double pi2 = 6.283185307179586476925286766559L;
    auto root = [&] (long num, long den) { const long double ang = pi2*(num % den)/den;
                                           return make_double2((double)cosl(ang), (double)sinl(ang)); };
    size_t o = 0;
    for (int n1 = a0; n1 <= a1; ++n1) for (int k1 = a0; k1 <= a1; ++k1) h[o++] = root((long)n1*k1, N1);
    for (int n2 = a0; n2 <= b1; ++n2) for (int k2 = a0; k2 <= b1; ++k2) h[o++] = root((long)n2*k2, N2);
    for (int k1 = 0; k1 < N1; ++k1) for (int n2 = 0; n2 < N2; ++n2) h[o++] = root((long)n2*k1, N);
    HPS_HIP_CHECK(hipMalloc(out, h.size()*sizeof(double2)));
    HPS_HIP_CHECK(hipMemcpy(*out, h.data(), h.size()*sizeof(double2), hipMemcpyHostToDevice));
    return HPS_OK;
}

// A-operand tables of mfma_stage for the two factors: per factor M (H = (M-1)/2, MT k tiles, KS steps of 4 in n)
// [cos | sin][MT][KS][64 lanes], lane l <-> (k = 16 mt + l % 16, n = 1 + 4 ks + l / 16); zero outside k <= H, n <= H
static int upload_mfma_tables (int N1, int N2, double** out, size_t* na)
{
    std::vector<double> h;
    const long double pi2 = 6.283185307179586476925286766559L;
    size_t first = 0;
    for (int M : {N1, N2}) {
        const int H = (M - 1)/2, MT = (H + 1 + 15)/16, KS = (H + 3)/4;
        for (int part = 0; part < 2; ++part)
            for (int mt = 0; mt < MT; ++mt) for (int ks = 0; ks < KS; ++ks) for (int l = 0; l < 64; ++l) {
                const int k = 16*mt + l % 16, n = 1 + 4*ks + l/16;
                double v = 0.0;
                if (k <= H && n <= H) {
                    const long double ang = pi2*(((long)n*k) % M)/M;
                    v = part == 0 ? (double)cosl(ang) : (double)sinl(ang);
                }
                h.push_back(v);
            }
        if (M == N1 && first == 0) first = h.size();
    }
    *na = first;
    HPS_HIP_CHECK(hipMalloc(out, h.size()*sizeof(double)));
    HPS_HIP_CHECK(hipMemcpy(*out, h.data(), h.size()*sizeof(double), hipMemcpyHostToDevice));
    return HPS_OK;
}

int poisson_create (int nx, int ny, double dx, double dy, bool allow_own, Poisson** out)
{
    Poisson* P = new Poisson;
    P->nx = nx; P->ny = ny;
    const int Nx = nx + 1, Ny = ny + 1;
    const DstImpl* ix = allow_own ? find_dst_impl(Nx) : nullptr;
    const DstImpl* iy = allow_own ? find_dst_impl(Ny) : nullptr;
    // the y direction as tridiagonal solves wherever a transform along x exists (own kernels or the dense product)
    const TriImpl ti = find_tri_impl(ny);
    const bool want_tri = [&] { const char* v = getenv("HPS_POISSON_TRIDIAG");
                                return allow_own && ti.kernel && !(v && atoi(v) == 0) && !getenv("HPS_POISSON_MFMA") && !getenv("HPS_POISSON_COLS"); }();
    auto make_tri = [&] () -> int {
        // LU factors of tridiag(1, b_k, 1) in extended precision, rounded once: c'_0 = 1/b, c'_j = 1/(b - c'_{j-1})
        const long pa = P->pa;            // (the table has the planes' pitch: same addresses modulo a line, pad columns 0)
        std::vector<double> h((size_t)pa*ny, 0.0);
        const long double pi = 3.14159265358979323846264338327950288L;
        for (int k = 0; k < nx; ++k) {
            const long double sk = sinl(pi*(k + 1)/(2.0L*(nx + 1)));
            const long double b = -2.0L - 4.0L*sk*sk*((long double)dy*dy)/((long double)dx*dx);
            long double cprev = 0.0L;
            for (int j = 0; j < ny; ++j) { cprev = 1.0L/(b - cprev); h[(size_t)j*pa + k] = (double)cprev; }
        }
        HPS_HIP_CHECK(hipMalloc(&P->tri_cp, h.size()*sizeof(double)));
        HPS_HIP_CHECK(hipMemcpy(P->tri_cp, h.data(), h.size()*sizeof(double), hipMemcpyHostToDevice));
        P->ktri = ti.kernel; P->tri_M = ti.M; P->tri_cols = ti.cols;
        P->tri_alpha = dy*dy/(2.0*(nx + 1));
        return HPS_OK;
    };
    P->pa = nx;
    if (ix && want_tri) {
        P->pa = ((long)nx + 15)/16*16;        // rows of the intermediate planes start on 128-byte lines (nx = 2^K - 1: 24.7 -> 18 us for the y solves)
        P->kx = ix->kernel; P->kx_src = ix->src;
        int e;
        size_t nax, nbx;
        if ((e = upload_tables(ix->N1, ix->N2, ix->sym, &P->tab_x, &nax, &nbx, ix->pow2)) || (e = make_tri())) { delete P; return e; }
        P->fa_x = P->tab_x; P->fb_x = P->fa_x + nax; P->tw_x = P->fb_x + nbx;
        P->tx = ix->T; P->ntx = ix->nt;
        P->gate_ok = true;
        P->lds_x = ((size_t)ix->T*Nx + nax + nbx + (ix->pow2 ? Nx : 0))*sizeof(double2);
        for (dst_kernel_t kf : {P->kx, P->kx_src})
            if (kf && P->lds_x > 64*1024) HPS_HIP_CHECK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_x));
        HPS_HIP_CHECK(hipMalloc(&P->buf_a, ((size_t)DST_MAXPLANES*P->pa*ny + 64)*sizeof(double)));
    } else
    if (ix && iy) {
        P->kx = ix->kernel; P->ky = iy->kernel; P->kx_src = ix->src;
        {   const char* v = getenv("HPS_POISSON_Y2"); P->ky2 = (v && atoi(v) == 0) ? nullptr : iy->twice; }
        int e;
        size_t nax, nbx, nay, nby;
        if ((e = upload_tables(ix->N1, ix->N2, ix->sym, &P->tab_x, &nax, &nbx, ix->pow2)) ||
            (e = upload_tables(iy->N1, iy->N2, iy->sym, &P->tab_y, &nay, &nby, iy->pow2))) { delete P; return e; }
        P->fa_x = P->tab_x; P->fb_x = P->fa_x + nax; P->tw_x = P->fb_x + nbx;
        P->fa_y = P->tab_y; P->fb_y = P->fa_y + nay; P->tw_y = P->fb_y + nby;
        P->tx = ix->T; P->ty = iy->T; P->ntx = ix->nt; P->nty = iy->nt;
        P->gate_ok = true;
        // fp64-MFMA form of the small-DFT stages: parity-tested, but not the default -- v_mfma_f64_16x16x4 runs at the
        // vector fp64 rate on gfx950 (71 TFLOP/s measured), the padded tiles do 1.5x the flops, and with two workgroups
        // per CU the MFMA pipes are the bottleneck: 24.5 us per pass against 22 us for the vector kernel (a lone
        // workgroup per CU is 20 % faster with MFMA)
        if (ix->mfma && iy->mfma && getenv("HPS_POISSON_MFMA")) {
            size_t fx = 0, fy = 0;
            if ((e = upload_mfma_tables(ix->N1, ix->N2, &P->mtab_x, &fx)) || (e = upload_mfma_tables(iy->N1, iy->N2, &P->mtab_y, &fy))) { delete P; return e; }
            P->ma_x = P->mtab_x; P->mb_x = P->mtab_x + fx; P->ma_y = P->mtab_y; P->mb_y = P->mtab_y + fy;
            P->kx = ix->mfma; P->ky = iy->mfma; P->kx_src = nullptr; P->ky2 = nullptr; P->gate_ok = false;
        }
        if (iy->cols && getenv("HPS_POISSON_COLS")) {      // measured no faster than rows + transposes (0.143 ms both): off by default
            P->kcols = iy->cols; P->gate_ok = false;
            P->lds_cols = (size_t)DSTC_T*Ny*sizeof(double2);
            if (P->lds_cols > 64*1024) HPS_HIP_CHECK(hipFuncSetAttribute((const void*)P->kcols, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_cols));
        }
        P->lds_x = ((size_t)ix->T*Nx + nax + nbx + (ix->pow2 ? Nx : 0))*sizeof(double2);
        P->lds_y = ((size_t)iy->T*Ny + nay + nby + (iy->pow2 ? Ny : 0))*sizeof(double2);
        if (P->lds_x > 64*1024) HPS_HIP_CHECK(hipFuncSetAttribute((const void*)P->kx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_x));
        if (P->kx_src && P->lds_x > 64*1024) HPS_HIP_CHECK(hipFuncSetAttribute((const void*)P->kx_src, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_x));
        if (P->lds_y > 64*1024) HPS_HIP_CHECK(hipFuncSetAttribute((const void*)P->ky, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_y));
        if (P->ky2 && P->lds_y > 64*1024) HPS_HIP_CHECK(hipFuncSetAttribute((const void*)P->ky2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_y));
        size_t plane_doubles = (size_t)nx*ny;
        {   const char* v = getenv("HPS_POISSON_BLOCKED");
            // Measured (profiles/r04b_poisson_blocked_vs_transposes.txt, r04g_ab_inflight_byte_cuts.txt): with ONE stage on the GPU the
            // two transposes cost 18.4 us per slice and the three passes on blocked planes 23.5 us more than on row-major ones at
            // 1024^2 (the y pass 45.9 against 32.1 us: with B = 6 its 288-byte runs straddle 128-byte lines that other workgroups
            // -- on other XCDs -- complete): 0.3 % slower.  With three stages in flight -- where the GPU's bandwidth is what is
            // shared -- the 100 MB less per slice are +2 % (512^2: +3 % with one stage, +6 % with three).  On by default; 0: off.
            const bool want = !(v && atoi(v) == 0);
            if (want && ix->sym && iy->sym && ix->T == iy->T && P->ky2 && !P->kcols && !P->mtab_x) {
                const int B = 2*ix->T;
                P->kb_first = ix->b_first; P->kb_first_src = ix->b_first_src; P->kb_twice = iy->b_twice; P->kb_last = ix->b_last;
                P->blk_plane = (long)((ny + B - 1)/B)*blk_pitch(ix->T)*nx;
                plane_doubles = std::max(plane_doubles, (size_t)P->blk_plane);
                for (dst_kernel_t kf : {P->kb_first, P->kb_first_src, P->kb_last})
                    if (P->lds_x > 64*1024) HPS_HIP_CHECK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_x));
                if (P->lds_y > 64*1024) HPS_HIP_CHECK(hipFuncSetAttribute((const void*)P->kb_twice, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_y));
            }
        }
        HPS_HIP_CHECK(hipMalloc(&P->buf_a, (DST_MAXPLANES*plane_doubles + 64)*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&P->buf_b, (size_t)DST_MAXPLANES*nx*ny*sizeof(double)));
    } else if (allow_own && nx <= 512 && ny <= 512) {
        P->kx = P->ky = nullptr;
        auto sines = [] (int n, double** out) -> int {
            std::vector<double> h((size_t)n*n);
            for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k)
                h[(size_t)j*n + k] = (double)(2.0L*sinl(3.14159265358979323846264338327950288L*(long double)((j + 1.0L)*(k + 1.0L))/(long double)(n + 1)));
            HPS_HIP_CHECK(hipMalloc(out, h.size()*sizeof(double)));
            HPS_HIP_CHECK(hipMemcpy(*out, h.data(), h.size()*sizeof(double), hipMemcpyHostToDevice));
            return (int)HPS_OK;
        };
        if (int e = sines(nx, &P->S_x)) { delete P; return e; }
        if (ny == nx) P->S_y = P->S_x;
        else if (int e = sines(ny, &P->S_y)) { delete P; return e; }
        HPS_HIP_CHECK(hipMalloc(&P->buf_a, (size_t)DST_MAXPLANES*nx*ny*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&P->buf_b, (size_t)DST_MAXPLANES*nx*ny*sizeof(double)));
        if (want_tri) { if (int e = make_tri()) { delete P; return e; } }
    } else {
        P->kx = P->ky = nullptr;
        if (!g_rocfft_setup) { rocfft_setup(); g_rocfft_setup = true; }
        const int nhx = Nx/2 + 1, nhy = Ny/2 + 1;
        int e;
        if ((e = make_plan(&P->plan_x, Nx, ny)) || (e = make_plan(&P->plan_y, Ny, nx))) { delete P; return e; }
        size_t wx = 0, wy = 0;
        rocfft_plan_get_work_buffer_size(P->plan_x, &wx);
        rocfft_plan_get_work_buffer_size(P->plan_y, &wy);
        P->work_bytes = std::max(wx, wy);
        rocfft_execution_info_create(&P->info);
        if (P->work_bytes) {
            HPS_HIP_CHECK(hipMalloc(&P->work, P->work_bytes));
            rocfft_execution_info_set_work_buffer(P->info, P->work, P->work_bytes);
        }
        const size_t zc = std::max((size_t)nhx*ny, (size_t)nhy*nx);
        const size_t rc = std::max((size_t)Nx*ny, (size_t)Ny*nx);
        HPS_HIP_CHECK(hipMalloc(&P->zbuf, zc*sizeof(double2)));
        HPS_HIP_CHECK(hipMalloc(&P->rbuf, rc*sizeof(double)));
    }
    HPS_HIP_CHECK(hipMalloc(&P->eig, (size_t)nx*ny*sizeof(double)));
    HPS_HIP_CHECK(hipMalloc(&P->isin_x, nx*sizeof(double)));
    HPS_HIP_CHECK(hipMalloc(&P->isin_y, ny*sizeof(double)));

    // spectral operator in transposed (x-frequency major, y-frequency contiguous) layout
    std::vector<double> h_eig((size_t)nx*ny), hx(nx), hy(ny);
    const double pi = 3.14159265358979323846;
    const double sxf = pi/(2.*(nx + 1)), syf = pi/(2.*(ny + 1));
    const double norm_fac = 0.5/(2*((double)(nx + 1)*(ny + 1)));
    for (int k = 0; k < nx; ++k) {
        const double sxq = std::sin((k + 1)*sxf)*std::sin((k + 1)*sxf);
        for (int l = 0; l < ny; ++l) {
            const double syq = std::sin((l + 1)*syf)*std::sin((l + 1)*syf);
            h_eig[(size_t)k*ny + l] = (sxq != 0 && syq != 0) ? norm_fac/(-4.0*(sxq/(dx*dx) + syq/(dy*dy))) : 0.0;
        }
    }
    for (int k = 0; k < nx; ++k) hx[k] = 1.0/(4.0*std::sin(pi*(k + 1.0)/(nx + 1.0)));
    for (int l = 0; l < ny; ++l) hy[l] = 1.0/(4.0*std::sin(pi*(l + 1.0)/(ny + 1.0)));
    HPS_HIP_CHECK(hipMemcpy(P->eig, h_eig.data(), h_eig.size()*sizeof(double), hipMemcpyHostToDevice));
    HPS_HIP_CHECK(hipMemcpy(P->isin_x, hx.data(), nx*sizeof(double), hipMemcpyHostToDevice));
    HPS_HIP_CHECK(hipMemcpy(P->isin_y, hy.data(), ny*sizeof(double), hipMemcpyHostToDevice));
    *out = P;
    return HPS_OK;
}

static int run_fft (Poisson* P, rocfft_plan plan)
{
    void* in[1] = {P->zbuf};
    void* outp[1] = {P->rbuf};
    if (rocfft_execute(plan, in, outp, P->info) != rocfft_status_success) {
        set_error("rocfft_execute failed");
        return HPS_ERR_FFT;
    }
    return HPS_OK;
}

static int solve_rocfft (Poisson* P, const double* src, long src_pitch, double* dst, long dst_pitch, hipStream_t st)
{
    const int nx = P->nx, ny = P->ny;
    const int Nx = nx + 1, Ny = ny + 1;
    const int nhx = Nx/2 + 1, nhy = Ny/2 + 1;
    if (!P->stream_bound || P->bound_stream != st) {
        rocfft_execution_info_set_stream(P->info, st);
        P->bound_stream = st; P->stream_bound = true;
    }
    int e;
    hipLaunchKernelGGL(k_pre_rows, dim3(ceil_div(nhx, 256), ny), dim3(256), 0, st, src, src_pitch, P->zbuf, nx, nhx, ny);
    if ((e = run_fft(P, P->plan_x))) return e;
    hipLaunchKernelGGL(k_post_transpose_pre, dim3(ceil_div(nx, TP_K), ceil_div(nhy, TP_P)), dim3(256), 0, st,
                       P->rbuf, nx, ny, P->isin_x, P->zbuf, nhy);
    if ((e = run_fft(P, P->plan_y))) return e;
    hipLaunchKernelGGL(k_post_mult_pre, dim3(ceil_div(nhy, 256), nx), dim3(256), 0, st,
                       P->rbuf, ny, nx, P->isin_y, P->eig, P->zbuf, nhy);
    if ((e = run_fft(P, P->plan_y))) return e;
    hipLaunchKernelGGL(k_post_transpose_pre, dim3(ceil_div(ny, TP_K), ceil_div(nhx, TP_P)), dim3(256), 0, st,
                       P->rbuf, ny, nx, P->isin_y, P->zbuf, nhx);
    if ((e = run_fft(P, P->plan_x))) return e;
    hipLaunchKernelGGL(k_post_to_slab, dim3(ceil_div(nx, 256), ny), dim3(256), 0, st,
                       P->rbuf, nx, ny, P->isin_x, dst, dst_pitch);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

// nb independent solves in one batch: src[b] = nx*ny source with row pitch src_pitch,
// dst[b] = pointer to cell (0,0) of the target plane with row pitch dst_pitch
static double* slab_cell00 (const hps_slab& s, int comp)
{
    return s.p + (long)comp*s.nstride + s.ng + (long)s.ng*s.jstride;
}

// `spec` (optional, own sym transform without the column kernel only: poisson_sources_fusable): the sources are not read from
// `src` but formed from other planes while the first pass loads its rows (PoissonSrc: value = sum of c*(p - q) pairs)
static int poisson_solve_batch_impl (Poisson* P, int nb, const double* const* src, long src_pitch, double* const* dst, long dst_pitch,
                                     const PoissonSrc* spec, hipStream_t st);
int poisson_solve_batch (Poisson* P, int nb, const double* const* src, long src_pitch, double* const* dst, long dst_pitch,
                         hipStream_t st)
{
    return poisson_solve_batch_impl(P, nb, src, src_pitch, dst, dst_pitch, nullptr, st);
}
// Device-side control of a caller's loop (the predictor-corrector iterations enqueued ahead of the host's knowledge of
// their number): every kernel of the solves enqueued from now on looks at *gate first and does nothing when it is 0.
// Only for back-ends whose every launch honours it (dense products, the symmetric own transform with blocked planes).
bool poisson_gateable (void* handle)
{
    Poisson* P = static_cast<Poisson*>(handle);
    return P->dense() || (P->own() && P->gate_ok);
}
void poisson_set_gate (void* handle, const int* gate) { static_cast<Poisson*>(handle)->gate = gate; }
bool poisson_sources_fusable (void* handle)
{
    Poisson* P = static_cast<Poisson*>(handle);
    return P->own() && (P->kx_src || P->kb_first_src) && !P->kcols && !P->dense();
}
int poisson_solve_batch_src (void* handle, int nb, const PoissonSrc* spec, long src_pitch, hps_slab dst, const int* dst_comps, hipStream_t st)
{
    Poisson* P = static_cast<Poisson*>(handle);
    HPS_REQUIRE(poisson_sources_fusable(P) && nb >= 1 && nb <= DST_MAXPLANES, "poisson_solve_batch_src: not available for this solver");
    double* d[DST_MAXPLANES];
    for (int b = 0; b < nb; ++b) d[b] = slab_cell00(dst, dst_comps[b]);
    return poisson_solve_batch_impl(P, nb, nullptr, src_pitch, d, dst.jstride, spec, st);
}
static void launch_tri (Poisson* P, int nb, hipStream_t st)
{
    TriArgs t{};
    for (int b = 0; b < nb; ++b) t.plane[b] = P->buf_a + (long)b*P->pa*P->ny;
    t.pitch = P->pa; t.cp = P->tri_cp; t.alpha = P->tri_alpha; t.nx = P->nx; t.ny = P->ny;
    t.nseg = ceil_div(P->ny, P->tri_M); t.gate = P->gate;
    hipLaunchKernelGGL(P->ktri, dim3(ceil_div(P->nx, P->tri_cols), nb), dim3(P->tri_cols*t.nseg), 0, st, t);
}
static int poisson_solve_batch_impl (Poisson* P, int nb, const double* const* src, long src_pitch, double* const* dst, long dst_pitch,
                                     const PoissonSrc* spec, hipStream_t st)
{
    if (nb > DST_MAXPLANES) { set_error("poisson_solve_batch: too many planes"); return HPS_ERR_ARG; }
    if (P->dense()) {
        const int nx = P->nx, ny = P->ny;
        const long plane = (long)nx*ny;
        const dim3 grid(ceil_div(nx, 32), ceil_div(ny, 32), nb), block(256);
        GemmArgs g{};
        g.gate = P->gate;
        g.M = ny; g.N = nx;
        // 1: A = src . S_x
        for (int b = 0; b < nb; ++b) { g.A[b] = src[b]; g.B[b] = P->S_x; g.C[b] = P->buf_a + b*plane; }
        g.lda = src_pitch; g.ldb = nx; g.ldc = nx; g.K = nx; g.scale = nullptr;
        hipLaunchKernelGGL(k_dense_product, grid, block, 0, st, g);
        if (P->tri()) {
            // 2: the y direction of every x mode as a tridiagonal solve, in place; 3: dst = A . S_x
            launch_tri(P, nb, st);
            for (int b = 0; b < nb; ++b) { g.A[b] = P->buf_a + b*plane; g.B[b] = P->S_x; g.C[b] = dst[b]; }
            g.lda = nx; g.ldb = nx; g.ldc = dst_pitch; g.K = nx; g.scale = nullptr;
            hipLaunchKernelGGL(k_dense_product, grid, block, 0, st, g);
            HPS_HIP_CHECK(hipGetLastError());
            return HPS_OK;
        }
        // 2: B = (S_y . A) * inverse eigenvalues (stored x-frequency major)
        for (int b = 0; b < nb; ++b) { g.A[b] = P->S_y; g.B[b] = P->buf_a + b*plane; g.C[b] = P->buf_b + b*plane; }
        g.lda = ny; g.ldb = nx; g.ldc = nx; g.K = ny; g.scale = P->eig; g.scale_r = 1; g.scale_c = ny;
        hipLaunchKernelGGL(k_dense_product, grid, block, 0, st, g);
        // 3: A = S_y . B
        for (int b = 0; b < nb; ++b) { g.A[b] = P->S_y; g.B[b] = P->buf_b + b*plane; g.C[b] = P->buf_a + b*plane; }
        g.scale = nullptr;
        hipLaunchKernelGGL(k_dense_product, grid, block, 0, st, g);
        // 4: dst = A . S_x
        for (int b = 0; b < nb; ++b) { g.A[b] = P->buf_a + b*plane; g.B[b] = P->S_x; g.C[b] = dst[b]; }
        g.lda = nx; g.ldb = nx; g.ldc = dst_pitch; g.K = nx;
        hipLaunchKernelGGL(k_dense_product, grid, block, 0, st, g);
        HPS_HIP_CHECK(hipGetLastError());
        return HPS_OK;
    }
    if (!P->own()) {
        for (int b = 0; b < nb; ++b) if (int e = solve_rocfft(P, src[b], src_pitch, dst[b], dst_pitch, st)) return e;
        return HPS_OK;
    }
    const int nx = P->nx, ny = P->ny;
    const long plane = (long)nx*ny;
    auto rows_grid = [] (int rows, int T) { return dim3(ceil_div(rows, 2*T)); };
    DstArgs a{};
    a.dbg = P->dbg;
    a.gate = P->gate;
    if (P->blocked()) {
        // three launches, the intermediate planes in blocks of B = 2T rows ("blocked layout", k_dst_rows_sym<.., LIN, LOUT>)
        const int B = 2*P->tx;
        const int ypad = ceil_div(ny, B)*B, xpad = ceil_div(nx, B)*B;
        // 1: DST along x of the sources -> blocked planes
        for (int b = 0; b < nb; ++b) { a.src[b] = spec ? nullptr : src[b]; a.dst[b] = P->buf_a + b*P->blk_plane; }
        a.src_pitch = src_pitch; a.dst_pitch = nx; a.scale = nullptr; a.fa = P->fa_x; a.fb = P->fb_x; a.tw = P->tw_x; a.isin4 = P->isin_x;
        a.rows_per_plane = ny; a.nplanes = nb; a.rows_pad = ypad; a.blk_cols = nx;
        if (spec) {
            for (int b = 0; b < nb; ++b) {
                a.npairs[b] = spec[b].npairs;
                for (int k = 0; k < 2; ++k) { a.sp[b][k] = spec[b].p[k]; a.sq[b][k] = spec[b].q[k]; a.sc[b][k] = spec[b].c[k]; }
            }
            hipLaunchKernelGGL(P->kb_first_src, dim3(nb*ypad/B), dim3(P->ntx), P->lds_x, st, a);
            for (int b = 0; b < nb; ++b) a.npairs[b] = 0;
        } else
        hipLaunchKernelGGL(P->kb_first, dim3(nb*ypad/B), dim3(P->ntx), P->lds_x, st, a);
        // 2: DST along y, inverse eigenvalues, DST along y -- in place, B columns of the blocked planes per workgroup
        for (int b = 0; b < nb; ++b) { a.src[b] = P->buf_a + b*P->blk_plane; a.dst[b] = P->buf_a + b*P->blk_plane; }
        a.scale = P->eig; a.fa = P->fa_y; a.fb = P->fb_y; a.tw = P->tw_y; a.isin4 = P->isin_y;
        a.rows_per_plane = nx; a.rows_pad = xpad; a.blk_cols = nx;
        hipLaunchKernelGGL(P->kb_twice, dim3(nb*xpad/B), dim3(P->nty), P->lds_y, st, a);
        // 3: DST along x -> destination planes
        for (int b = 0; b < nb; ++b) { a.src[b] = P->buf_a + b*P->blk_plane; a.dst[b] = dst[b]; }
        a.scale = nullptr; a.dst_pitch = dst_pitch; a.fa = P->fa_x; a.fb = P->fb_x; a.tw = P->tw_x; a.isin4 = P->isin_x;
        a.rows_per_plane = ny; a.rows_pad = ypad; a.blk_cols = nx;
        hipLaunchKernelGGL(P->kb_last, dim3(nb*ypad/B), dim3(P->ntx), P->lds_x, st, a);
        HPS_HIP_CHECK(hipGetLastError());
        return HPS_OK;
    }
    // 1: DST along x of the sources -> A
    const long pa = P->tri() ? P->pa : (long)nx, plane_a = P->tri() ? pa*ny : plane;
    for (int b = 0; b < nb; ++b) { a.src[b] = spec ? nullptr : src[b]; a.dst[b] = P->buf_a + b*plane_a; }
    a.src_pitch = src_pitch; a.dst_pitch = pa; a.scale = nullptr; a.fa = P->fa_x; a.fb = P->fb_x; a.tw = P->tw_x; a.isin4 = P->isin_x;
    a.ma = P->ma_x; a.mb = P->mb_x;
    a.rows_per_plane = ny; a.nplanes = nb;
    if (spec) {
        for (int b = 0; b < nb; ++b) {
            a.npairs[b] = spec[b].npairs;
            for (int k = 0; k < 2; ++k) { a.sp[b][k] = spec[b].p[k]; a.sq[b][k] = spec[b].q[k]; a.sc[b][k] = spec[b].c[k]; }
        }
        hipLaunchKernelGGL(P->kx_src, rows_grid(ny*nb, P->tx), dim3(P->ntx), P->lds_x, st, a);
        for (int b = 0; b < nb; ++b) a.npairs[b] = 0;
    } else
    hipLaunchKernelGGL(P->kx, rows_grid(ny*nb, P->tx), dim3(P->ntx), P->lds_x, st, a);
    if (P->tri()) {
        // 2: the y direction of every x mode as a tridiagonal solve, in place on A (no transposes, no y transforms)
        launch_tri(P, nb, st);
    } else
    if (P->kcols) {
        // 2-5: DST along y, inverse eigenvalues, DST along y -- in place on column blocks of A
        for (int b = 0; b < nb; ++b) { a.src[b] = P->buf_a + b*plane; a.dst[b] = P->buf_a + b*plane; }
        a.src_pitch = nx; a.dst_pitch = nx; a.scale = P->eig; a.fa = P->fa_y; a.fb = P->fb_y; a.tw = P->tw_y; a.isin4 = P->isin_y;
        a.rows_per_plane = ny;
        hipLaunchKernelGGL(P->kcols, dim3(nb*ceil_div(nx, 2*DSTC_T)), dim3(DSTS_NT), P->lds_cols, st, a, nx);
        a.scale = nullptr;
    } else {
        // 2: transpose -> B[k][j]
        hipLaunchKernelGGL(k_transpose, dim3(ceil_div(nx, 32), ceil_div(ny, 32), nb), dim3(256), 0, st, P->buf_a, P->buf_b, ny, nx, plane);
        // 3: DST along y, times the inverse eigenvalues -> A
        for (int b = 0; b < nb; ++b) { a.src[b] = P->buf_b + b*plane; a.dst[b] = P->buf_a + b*plane; }
        a.src_pitch = ny; a.dst_pitch = ny; a.scale = P->eig; a.fa = P->fa_y; a.fb = P->fb_y; a.tw = P->tw_y; a.isin4 = P->isin_y;
        a.ma = P->ma_y; a.mb = P->mb_y;
        a.rows_per_plane = nx;
        if (P->ky2) {
            // 3 + 4 in one launch: transform, inverse eigenvalues, transform again (in LDS) -> back into B
            for (int b = 0; b < nb; ++b) { a.src[b] = P->buf_b + b*plane; a.dst[b] = P->buf_b + b*plane; }
            hipLaunchKernelGGL(P->ky2, rows_grid(nx*nb, P->ty), dim3(P->nty), P->lds_y, st, a);
            a.scale = nullptr;
        } else {
        hipLaunchKernelGGL(P->ky, rows_grid(nx*nb, P->ty), dim3(P->nty), P->lds_y, st, a);
        // 4: DST along y again -> B
        for (int b = 0; b < nb; ++b) { a.src[b] = P->buf_a + b*plane; a.dst[b] = P->buf_b + b*plane; }
        a.scale = nullptr;
        hipLaunchKernelGGL(P->ky, rows_grid(nx*nb, P->ty), dim3(P->nty), P->lds_y, st, a);
        }
        // 5: transpose back -> A[j][k]
        hipLaunchKernelGGL(k_transpose, dim3(ceil_div(ny, 32), ceil_div(nx, 32), nb), dim3(256), 0, st, P->buf_b, P->buf_a, nx, ny, plane);
    }
    // 6: DST along x -> destination planes
    for (int b = 0; b < nb; ++b) { a.src[b] = P->buf_a + b*plane_a; a.dst[b] = dst[b]; }
    a.src_pitch = pa; a.dst_pitch = dst_pitch; a.fa = P->fa_x; a.fb = P->fb_x; a.tw = P->tw_x; a.isin4 = P->isin_x;
    a.ma = P->ma_x; a.mb = P->mb_x;
    a.rows_per_plane = ny;
    hipLaunchKernelGGL(P->kx, rows_grid(ny*nb, P->tx), dim3(P->ntx), P->lds_x, st, a);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

} // namespace hps

using namespace hps;

extern "C" int hps_poisson_create (int nx, int ny, double dx, double dy, void** handle)
{
    HPS_REQUIRE(nx >= 2 && ny >= 2 && handle, "hps_poisson_create: bad size");
    Poisson* P = nullptr;
    const char* env = getenv("HPS_POISSON_BACKEND");     // "rocfft" forces the library back-end
    const bool allow_own = !(env && std::string(env) == "rocfft");
    if (int e = poisson_create(nx, ny, dx, dy, allow_own, &P)) return e;
    *handle = P;
    return HPS_OK;
}


extern "C" int hps_poisson_solve (void* handle, const double* staging, hps_slab dst, int dst_comp, hps_stream stream)
{
    HPS_REQUIRE(handle && staging && dst.p, "hps_poisson_solve: null argument");
    Poisson* P = static_cast<Poisson*>(handle);
    HPS_REQUIRE(dst.nx == P->nx && dst.ny == P->ny, "hps_poisson_solve: slab size does not match the solver");
    HPS_REQUIRE(dst_comp >= 0 && dst_comp < dst.ncomp, "hps_poisson_solve: bad component");
    const double* s[1] = {staging};
    double* d[1] = {slab_cell00(dst, dst_comp)};
    return poisson_solve_batch(P, 1, s, P->nx, d, dst.jstride, (hipStream_t)stream);
}

extern "C" int hps_poisson_solve_batch (void* handle, int nbatch, const double* staging, hps_slab dst, const int* dst_comps,
                                        hps_stream stream)
{
    HPS_REQUIRE(handle && staging && dst.p && dst_comps, "hps_poisson_solve_batch: null argument");
    HPS_REQUIRE(nbatch >= 1 && nbatch <= DST_MAXPLANES, "hps_poisson_solve_batch: 1..4 solves per batch");
    Poisson* P = static_cast<Poisson*>(handle);
    HPS_REQUIRE(dst.nx == P->nx && dst.ny == P->ny, "hps_poisson_solve_batch: slab size does not match the solver");
    const double* s[DST_MAXPLANES]; double* d[DST_MAXPLANES];
    for (int b = 0; b < nbatch; ++b) {
        HPS_REQUIRE(dst_comps[b] >= 0 && dst_comps[b] < dst.ncomp, "hps_poisson_solve_batch: bad component");
        s[b] = staging + (long)b*P->nx*P->ny;
        d[b] = slab_cell00(dst, dst_comps[b]);
    }
    return poisson_solve_batch(P, nbatch, s, P->nx, d, dst.jstride, (hipStream_t)stream);
}

extern "C" int hps_poisson_debug_stamps (void* handle, long long* stamps6_host)      // [4 workgroups][6 stamps]
{
    Poisson* P = static_cast<Poisson*>(handle);
    if (!P->dbg) { HPS_HIP_CHECK(hipMalloc(&P->dbg, 24*sizeof(long long))); HPS_HIP_CHECK(hipMemset(P->dbg, 0, 24*sizeof(long long))); return HPS_OK; }
    HPS_HIP_CHECK(hipDeviceSynchronize());
    HPS_HIP_CHECK(hipMemcpy(stamps6_host, P->dbg, 24*sizeof(long long), hipMemcpyDeviceToHost));
    return HPS_OK;
}

extern "C" int hps_poisson_destroy (void* handle)
{
    delete static_cast<Poisson*>(handle);
    return HPS_OK;
}
