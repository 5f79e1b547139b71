// zslab.hip -- z-direction stage of the Poisson solve on z-slabs WITHOUT transposing the spectrum.
//
// After the per-plane 2-D transforms each (kx,ky) column satisfies the periodic three-point system
//     phi[k-1] + beta phi[k] + phi[k+1] = f[k],   beta = -(2 + (lx+ly) dz^2),  f = dz^2 b^      (k global, periodic)
// which is exactly what the reference's z-FFT + eigenvalue division solves (fft_based_poisson_solver.jl:93-111:
// lz are the eigenvalues of this very operator, poisson_eigenvalues.jl:8-11).  Its inverse is a convolution with
// the periodic Green's function C rho^|d| (rho + 1/rho = -beta, 0 < rho < 1, C = rho / (rho^2 - 1)), i.e. two
// first-order recursive sweeps along z.  On a slab each rank runs the sweeps on its own levels and only needs
// two numbers per column from every other rank (the weighted sums of their levels):
//     SP_r = sum_j rho^(n-1-j) f_j      SQ_r = sum_j rho^j f_j
// so the per-solve communication is ~1 MB to each peer instead of two all-to-all transposes of the whole
// half spectrum (2 x 118 MB per GPU at 256^3 per GPU).  xGMI links are point-to-point and per-link bound: this
// is the formulation that fits them.  The (0,0) column (rho = 1, singular) is gathered whole and solved with
// the zero-mean gauge of the reference (phi^[1,1,1] = 0).
#include "internal.h"

struct zc {
  double x, y;
};
#define ZS_MR 8   // segments up to this many levels are solved in registers (k_zslab_down_reg)

OCN_DEVFN double col_rho(double lam_dz2) {
  // roots of rho^2 + beta rho + 1 = 0 with -beta = 2 + s:  rho = 2 / (-beta + sqrt(beta^2 - 4))
  const double nb = 2.0 + lam_dz2;
  return 2.0 / (nb + sqrt(nb * nb - 4.0));
}

// Each column's levels are split into SZ segments of m = n / SZ levels handled by different threads (the same
// carry algebra that couples ranks couples segments), so a sweep has ncol * SZ independent recurrences.
//
// pass 1 (upward, per segment): P_i = rho P_{i-1} + f_i stored in place (zero carry-in);
//   segment sums  SP_s = P_{m-1},  SQ_s = sum_i rho^i f_i          -> segs[(s*2 + {0,1}) * ncol + col]
__global__ void k_zslab_up(zc* __restrict__ a, long ncol, int m, int SZ, const double* __restrict__ lxy, double dz2,
                           zc* __restrict__ segs, int store) {
  const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (col >= ncol) return;
  const double lam = lxy[col] * dz2;
  if (lam == 0.0) {   // the singular column is handled by k_zslab_mean
    segs[((size_t)s * 2 + 0) * ncol + col] = {0, 0};
    segs[((size_t)s * 2 + 1) * ncol + col] = {0, 0};
    return;
  }
  const double rho = col_rho(lam);
  zc P = {0, 0}, SQ = {0, 0};
  double pw = 1.0;
  zc* p = a + col + ncol * (size_t)s * m;
  // batches of 8 levels: all loads of a batch are issued before the (serial) recurrence consumes them
  for (int i0 = 0; i0 < m; i0 += 8) {
    zc f[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (i0 + q < m) f[q] = p[ncol * (size_t)(i0 + q)];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (i0 + q < m) {
        P.x = fma(rho, P.x, f[q].x);
        P.y = fma(rho, P.y, f[q].y);
        SQ.x = fma(pw, f[q].x, SQ.x);
        SQ.y = fma(pw, f[q].y, SQ.y);
        pw *= rho;
        f[q] = P;
      }
    if (store) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (i0 + q < m) p[ncol * (size_t)(i0 + q)] = f[q];
    }
  }
  segs[((size_t)s * 2 + 0) * ncol + col] = P;
  segs[((size_t)s * 2 + 1) * ncol + col] = SQ;
}

// rank sums from the segment sums: SP_r = sum_s rho^((SZ-1-s) m) SP_s,  SQ_r = sum_s rho^(s m) SQ_s
//
// `bel` (may be null): a source this rank holds for the level just BELOW its slab -- local index -1, i.e. the top level of the
// lower neighbour (round 3: the w* term of that level's divergence, transformed here instead of shipping the w* plane down
// before the right-hand side can be formed).  The convolution with C rho^|d| does not care who owns a source as long as it
// is counted once at its position: in this rank's message it weighs rho^n in SP (distance to the level above the slab's top)
// and rho^-1 in SQ; in this rank's own sweep it is the first term of the carry from below (k_zslab_down*).
__global__ void k_zslab_ranksums(long ncol, int m, int SZ, const double* __restrict__ lxy, double dz2,
                                 const zc* __restrict__ segs, zc* __restrict__ sums /* [2][ncol] */, const zc* __restrict__ bel) {
  const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncol) return;
  const double lam = lxy[col] * dz2;
  zc SP = {0, 0}, SQ = {0, 0};
  if (lam != 0.0) {
    const double rho1 = col_rho(lam);
    const double rm = exp(log(rho1) * m);
    double w = 1.0;
    for (int s = 0; s < SZ; ++s) {      // Horner for SP, running power for SQ
      zc sp = segs[((size_t)s * 2 + 0) * ncol + col], sq = segs[((size_t)s * 2 + 1) * ncol + col];
      SP.x = fma(rm, SP.x, sp.x);
      SP.y = fma(rm, SP.y, sp.y);
      SQ.x = fma(w, sq.x, SQ.x);
      SQ.y = fma(w, sq.y, SQ.y);
      w *= rm;
    }
    if (bel) {
      const zc d = bel[col];
      const double rn = exp(log(rho1) * ((double)m * SZ)), ri = 1.0 / rho1;
      SP.x = fma(rn, d.x, SP.x);
      SP.y = fma(rn, d.y, SP.y);
      SQ.x = fma(ri, d.x, SQ.x);
      SQ.y = fma(ri, d.y, SQ.y);
    }
  }
  sums[col] = SP;
  sums[ncol + col] = SQ;
}

// pass 2 (downward, per segment): x_i = scale C [P_i + rho Q_{i+1} + rho^(i+1) cP + rho^(m-i) cQ], i local to the
// segment; cP / cQ collect everything below / above the segment: the rank's other segments and, through the
// gathered rank sums [R][2][ncol], every other slab with all periodic images.
__global__ void k_zslab_down(zc* __restrict__ a, long ncol, int m, int SZ, int R, int rank, const double* __restrict__ lxy,
                             double dz2, double scale, const zc* __restrict__ segs, const zc* __restrict__ gathered,
                             size_t msg, const zc* __restrict__ bel) {
  const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (col >= ncol) return;
  const double lam = lxy[col] * dz2;
  if (lam == 0.0) return;
  const double rho = col_rho(lam);
  const double lnr = log(rho);
  const int n = m * SZ;
  const double rm = exp(lnr * m);                 // rho^m (may underflow to 0: then the images vanish, as they should)
  const double rn = exp(lnr * n);
  const double geo = 1.0 / (1.0 - exp(lnr * ((double)n * R)));
  // rank-level carries: slabs below (P) and above (Q), all periodic images included
  zc cP = {0, 0}, cQ = {0, 0};
  double w = 1.0;
  for (int mm = 1; mm <= R; ++mm) {
    const int rb = ((rank - mm) % R + R) % R, ra = (rank + mm) % R;
    zc sp = gathered[(size_t)rb * msg + col];
    zc sq = gathered[(size_t)ra * msg + ncol + col];
    cP.x = fma(w, sp.x, cP.x);
    cP.y = fma(w, sp.y, cP.y);
    cQ.x = fma(w, sq.x, cQ.x);
    cQ.y = fma(w, sq.y, cQ.y);
    w *= rn;
  }
  cP.x *= geo; cP.y *= geo; cQ.x *= geo; cQ.y *= geo;
  if (bel) { cP.x += bel[col].x; cP.y += bel[col].y; }   // this rank's source at index -1 (its periodic images came with the own-rank term)
  // segment-level: add the rank's own segments below / above this one
  {
    zc lo = {cP.x, cP.y};                         // carry entering segment 0 from below
    for (int t = 0; t < s; ++t) {
      zc sp = segs[((size_t)t * 2 + 0) * ncol + col];
      lo.x = fma(rm, lo.x, sp.x);
      lo.y = fma(rm, lo.y, sp.y);
    }
    zc hi = {cQ.x, cQ.y};                         // carry entering segment SZ-1 from above
    for (int t = SZ - 1; t > s; --t) {
      zc sq = segs[((size_t)t * 2 + 1) * ncol + col];
      hi.x = fma(rm, hi.x, sq.x);
      hi.y = fma(rm, hi.y, sq.y);
    }
    cP = lo;
    cQ = hi;
  }
  const double C = rho / (rho * rho - 1.0) * scale;
  zc* p = a + col + ncol * (size_t)s * m;
  zc Q = {0, 0};                                   // Q_{i+1}
  double pq = rho;                                 // rho^(m-i) at i = m-1
  // batches of 8 levels, top down; v[q] = P_{i1-q}, v[8] = P_{i1-8} (the next batch's first value)
  for (int i1 = m - 1; i1 >= 0; i1 -= 8) {
    zc v[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int i = i1 - q;
      v[q] = (i >= 0) ? p[ncol * (size_t)i] : zc{0, 0};
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i1 - q;
      if (i >= 0) {
        const zc Pi = v[q], Pm = v[q + 1];
        const double pp = exp(lnr * (i + 1));      // rho^(i+1)
        zc x;
        x.x = C * (Pi.x + rho * Q.x + pp * cP.x + pq * cQ.x);
        x.y = C * (Pi.y + rho * Q.y + pp * cP.y + pq * cQ.y);
        // f_i = P_i - rho P_{i-1};  Q_i = rho Q_{i+1} + f_i
        Q.x = fma(rho, Q.x, Pi.x - rho * Pm.x);
        Q.y = fma(rho, Q.y, Pi.y - rho * Pm.y);
        v[q] = x;
        pq *= rho;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i1 - q;
      if (i >= 0) p[ncol * (size_t)i] = v[q];
    }
  }
}

// pass 2 for short segments (m <= MR levels): the segment's right-hand sides fit in registers, so pass 1 only
// has to produce the sums (no in-place P array: 25 % less traffic over both passes) and the powers of rho are
// running products instead of one exp() per level.  Same formula as k_zslab_down with f used directly.
template <int MR>
__global__ void k_zslab_down_reg(zc* __restrict__ a, long ncol, int m, int SZ, int R, int rank, const double* __restrict__ lxy,
                                 double dz2, double scale, const zc* __restrict__ segs, const zc* __restrict__ gathered,
                                 size_t msg, const zc* __restrict__ bel) {
  const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (col >= ncol) return;
  const double lam = lxy[col] * dz2;
  if (lam == 0.0) return;
  zc* p = a + col + ncol * (size_t)s * m;
  zc f[MR];
#pragma unroll
  for (int q = 0; q < MR; ++q)
    if (q < m) f[q] = p[ncol * (size_t)q];
  const double rho = col_rho(lam);
  const double lnr = log(rho);
  const int n = m * SZ;
  const double rm = exp(lnr * m);
  const double rn = exp(lnr * n);
  const double geo = 1.0 / (1.0 - exp(lnr * ((double)n * R)));
  zc cP = {0, 0}, cQ = {0, 0};
  double w = 1.0;
  for (int mm = 1; mm <= R; ++mm) {
    const int rb = ((rank - mm) % R + R) % R, ra = (rank + mm) % R;
    zc sp = gathered[(size_t)rb * msg + col];
    zc sq = gathered[(size_t)ra * msg + ncol + col];
    cP.x = fma(w, sp.x, cP.x);
    cP.y = fma(w, sp.y, cP.y);
    cQ.x = fma(w, sq.x, cQ.x);
    cQ.y = fma(w, sq.y, cQ.y);
    w *= rn;
  }
  cP.x *= geo; cP.y *= geo; cQ.x *= geo; cQ.y *= geo;
  if (bel) { cP.x += bel[col].x; cP.y += bel[col].y; }
  for (int t = 0; t < s; ++t) {
    zc sp = segs[((size_t)t * 2 + 0) * ncol + col];
    cP.x = fma(rm, cP.x, sp.x);
    cP.y = fma(rm, cP.y, sp.y);
  }
  for (int t = SZ - 1; t > s; --t) {
    zc sq = segs[((size_t)t * 2 + 1) * ncol + col];
    cQ.x = fma(rm, cQ.x, sq.x);
    cQ.y = fma(rm, cQ.y, sq.y);
  }
  const double C = rho / (rho * rho - 1.0) * scale;
  // upward: P_i = rho P_{i-1} + f_i plus the carry from below, rho^(i+1) cP, folded in as P'_i = rho P'_{i-1} + f_i, P'_{-1} = cP
  zc P[MR];
  zc run = cP;
#pragma unroll
  for (int q = 0; q < MR; ++q)
    if (q < m) {
      run.x = fma(rho, run.x, f[q].x);
      run.y = fma(rho, run.y, f[q].y);
      P[q] = run;
    }
  // downward: Q'_i = rho Q'_{i+1} + f_i with Q'_m = cQ carries rho^(m-i) cQ;  x_i = C (P'_i + rho Q'_{i+1})
  zc Q = cQ;
#pragma unroll
  for (int q = MR - 1; q >= 0; --q)
    if (q < m) {
      zc x;
      x.x = C * fma(rho, Q.x, P[q].x);
      x.y = C * fma(rho, Q.y, P[q].y);
      Q.x = fma(rho, Q.x, f[q].x);
      Q.y = fma(rho, Q.y, f[q].y);
      p[ncol * (size_t)q] = x;
    }
}

// The solution ONE LEVEL BELOW the slab (local index -1), for d_z p at the slab's first level: x_{-1} = C (P_{-1} + rho Q_0) with
// P_{-1} = the carry from below (every slab below with its images, plus this rank's own source at -1) and Q_0 = this slab's
// own sum SQ plus rho^n times the carry from above -- everything is on this rank once the sums are gathered, so the pressure
// plane the lower neighbour used to send (fused_exchange_phi) is computed here instead (round 3).
__global__ void k_zslab_below(long ncol, int m, int SZ, int R, int rank, const double* __restrict__ lxy, double dz2, double scale,
                              const zc* __restrict__ segs, const zc* __restrict__ gathered, size_t msg, const zc* bel,
                              zc* out /* may alias bel */) {
  const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncol) return;
  const double lam = lxy[col] * dz2;
  if (lam == 0.0) return;                            // the singular column: k_zslab_mean
  const double rho = col_rho(lam);
  const double lnr = log(rho);
  const int n = m * SZ;
  const double rm = exp(lnr * m);
  const double rn = exp(lnr * n);
  const double geo = 1.0 / (1.0 - exp(lnr * ((double)n * R)));
  zc cP = {0, 0}, cQ = {0, 0};
  double w = 1.0;
  for (int mm = 1; mm <= R; ++mm) {
    const int rb = ((rank - mm) % R + R) % R, ra = (rank + mm) % R;
    zc sp = gathered[(size_t)rb * msg + col];
    zc sq = gathered[(size_t)ra * msg + ncol + col];
    cP.x = fma(w, sp.x, cP.x);
    cP.y = fma(w, sp.y, cP.y);
    cQ.x = fma(w, sq.x, cQ.x);
    cQ.y = fma(w, sq.y, cQ.y);
    w *= rn;
  }
  cP.x *= geo; cP.y *= geo; cQ.x *= geo; cQ.y *= geo;
  if (bel) { cP.x += bel[col].x; cP.y += bel[col].y; }
  zc SQ = {0, 0};                                    // this slab's own levels: sum_j rho^j f_j from the segment sums
  double ws = 1.0;
  for (int t = 0; t < SZ; ++t) {
    zc sq = segs[((size_t)t * 2 + 1) * ncol + col];
    SQ.x = fma(ws, sq.x, SQ.x);
    SQ.y = fma(ws, sq.y, SQ.y);
    ws *= rm;
  }
  const double C = rho / (rho * rho - 1.0) * scale;
  const zc Q0 = {fma(rn, cQ.x, SQ.x), fma(rn, cQ.y, SQ.y)};
  out[col] = {C * fma(rho, Q0.x, cP.x), C * fma(rho, Q0.y, cP.y)};
}

// the singular column (lx + ly = 0): second difference of x equals g = f - mean(f), zero-mean solution:
//   x_k - x_0 = k d_{-1} + sum_{j<k} c_j,  c = inclusive prefix sum of g,  d_{-1} = -mean(c),  x_0 from zero mean.
// One workgroup; every thread owns a contiguous chunk; the two prefix sums are chunk-local scans plus a
// Hillis-Steele scan of the chunk totals in LDS.  `gathered` holds every rank's message; the column's levels of
// rank q start at gathered[q * msg + 2 * ncol].
#define ZM_T 256
OCN_DEVFN zc zadd(zc a, zc b) { return {a.x + b.x, a.y + b.y}; }
OCN_DEVFN zc block_scan_excl(zc v, zc* sh /* [2][ZM_T] */, zc* total) {
  const int t = threadIdx.x;
  int cur = 0;
  sh[t] = v;
  __syncthreads();
  for (int off = 1; off < ZM_T; off <<= 1) {
    zc mine = sh[cur * ZM_T + t];
    if (t >= off) mine = zadd(mine, sh[cur * ZM_T + t - off]);
    sh[(1 - cur) * ZM_T + t] = mine;
    cur = 1 - cur;
    __syncthreads();
  }
  zc incl = sh[cur * ZM_T + t];
  *total = sh[cur * ZM_T + ZM_T - 1];
  __syncthreads();
  return {incl.x - v.x, incl.y - v.y};
}

__global__ void __launch_bounds__(ZM_T) k_zslab_mean(zc* __restrict__ a, long ncol, long col0, int n, int R, int rank,
                                                     const zc* __restrict__ gathered, size_t msg, double scale,
                                                     zc* __restrict__ work /* [R*n] */, zc* __restrict__ below /* may be null */) {
  OCN_SHARED zc sh[2 * ZM_T];
  const int t = threadIdx.x;
  const int N = n * R;
  const int chunk = (N + ZM_T - 1) / ZM_T;
  const int k0 = t * chunk, k1 = (k0 + chunk < N) ? k0 + chunk : N;
  auto F = [&](int k) {
    const int q = k / n, i = k - q * n;
    zc f = gathered[(size_t)q * msg + 2 * (size_t)ncol + i];
    if (i == n - 1) {   // the source rank q + 1 holds for the level below its slab = this level
      const zc d = gathered[(size_t)((q + 1) % R) * msg + 2 * (size_t)ncol + n];
      f = {f.x + d.x, f.y + d.y};
    }
    return f;
  };
  // mean of f
  zc part = {0, 0}, tot;
  for (int k = k0; k < k1; ++k) part = zadd(part, F(k));
  block_scan_excl(part, sh, &tot);
  const zc mean = {tot.x / N, tot.y / N};
  // c = inclusive prefix sum of g  (stored in work), and its total
  zc loc = {0, 0};
  for (int k = k0; k < k1; ++k) {
    zc f = F(k);
    loc = {loc.x + f.x - mean.x, loc.y + f.y - mean.y};
  }
  zc base = block_scan_excl(loc, sh, &tot);
  zc c = base, csum = {0, 0};
  for (int k = k0; k < k1; ++k) {
    zc f = F(k);
    c = {c.x + f.x - mean.x, c.y + f.y - mean.y};
    work[k] = c;
    csum = zadd(csum, c);
  }
  zc ctot;
  zc cbase = block_scan_excl(csum, sh, &ctot);      // sum of c over earlier chunks
  const zc dm1 = {-ctot.x / N, -ctot.y / N};
  // y_k = x_k - x_0 = k d_{-1} + sum_{j<k} c_j ; then the mean of y
  zc run = cbase, ysum = {0, 0};
  for (int k = k0; k < k1; ++k) {
    zc ck = work[k];
    zc y = {k * dm1.x + run.x, k * dm1.y + run.y};
    work[k] = y;
    ysum = zadd(ysum, y);
    run = zadd(run, ck);
  }
  zc ytot;
  block_scan_excl(ysum, sh, &ytot);
  const zc x0 = {-ytot.x / N, -ytot.y / N};
  __syncthreads();
  for (int i = t; i < n; i += ZM_T) {
    zc v = work[rank * n + i];
    a[col0 + ncol * (size_t)i] = {(v.x + x0.x) * scale, (v.y + x0.y) * scale};
  }
  if (below && t == 0) {                              // the level below this rank's slab: the whole column is here
    zc v = work[((rank * n - 1) % N + N) % N];
    below[col0] = {(v.x + x0.x) * scale, (v.y + x0.y) * scale};
  }
}

// copy column col0 of this rank's slab to a contiguous buffer (before pass 1 overwrites it)
__global__ void k_zslab_getcol(const zc* __restrict__ a, long ncol, long col0, int n, zc* __restrict__ out, const zc* __restrict__ bel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[col0 + ncol * (size_t)i];
  else if (i == n) out[n] = bel ? bel[col0] : zc{0, 0};   // entry n of the column message: the source below the slab
}

// ---- host side ----------------------------------------------------------------------------------------------------
struct ZSlab {
  long ncol = 0;
  int n = 0, R = 1, rank = 0;
  double* lxy = nullptr;
  zc* send = nullptr;      // [2][ncol] sums + [n] singular column + [1] its source below the slab
  zc* gathered = nullptr;  // [R] x that
  zc* gsums = nullptr;     // [R][2][ncol] (compact view used by pass 2)
  zc* fcol = nullptr;      // [R][n]
  zc* work = nullptr;
  zc* segs = nullptr;      // [SZ][2][ncol] segment sums
  int SZ = 1, m = 0;
  size_t msg = 0;          // elements per rank message
};

void* zslab_create(ocn_ctx* ctx, const std::vector<double>& lx_half, const std::vector<double>& ly, int n, int R, int rank) {
  ZSlab* z = new ZSlab;
  const size_t Nxh = lx_half.size(), Ny = ly.size();
  z->ncol = (long)(Nxh * Ny);
  z->n = n; z->R = R; z->rank = rank;
  std::vector<double> lxy(Nxh * Ny);
  for (size_t j = 0; j < Ny; ++j)
    for (size_t i = 0; i < Nxh; ++i) lxy[i + Nxh * j] = lx_half[i] + ly[j];
  z->SZ = 1;
  for (int cand : {64, 32, 16, 8, 4, 2})      // segments of 8 levels when possible (register kernel), never shorter
    if (n % cand == 0 && n / cand >= 8) { z->SZ = cand; break; }
  z->m = n / z->SZ;
  z->msg = 2 * (size_t)z->ncol + (size_t)n + 1;   // two sums per column, the singular column's levels, its source below the slab
  bool ok = hipMalloc((void**)&z->lxy, sizeof(double) * lxy.size()) == hipSuccess &&
            hipMalloc((void**)&z->send, sizeof(zc) * z->msg) == hipSuccess &&
            hipMalloc((void**)&z->gathered, sizeof(zc) * z->msg * R) == hipSuccess &&
            hipMalloc((void**)&z->gsums, sizeof(zc) * 2 * z->ncol * R) == hipSuccess &&
            hipMalloc((void**)&z->fcol, sizeof(zc) * (size_t)n * R) == hipSuccess &&
            hipMalloc((void**)&z->work, sizeof(zc) * (size_t)n * R) == hipSuccess &&
            hipMalloc((void**)&z->segs, sizeof(zc) * 2 * z->ncol * z->SZ) == hipSuccess;
  if (!ok) {
    ocn_set_error(ctx, "zslab: allocation failed");
    delete z;
    return nullptr;
  }
  hipMemcpy(z->lxy, lxy.data(), sizeof(double) * lxy.size(), hipMemcpyHostToDevice);
  return z;
}

void zslab_destroy(void* p) {
  ZSlab* z = (ZSlab*)p;
  if (!z) return;
  hipFree(z->lxy); hipFree(z->send); hipFree(z->gathered); hipFree(z->gsums); hipFree(z->fcol); hipFree(z->work); hipFree(z->segs);
  delete z;
}

// in place on this rank's (ncol, n) half spectrum.  dz2 = dz^2, scale = FFT normalisation 1/(Nx Ny).
int zslab_run(ocn_ctx* ctx, void* p, void* spec, double dz2, double scale, const void* below, void* phi_below) {
  ZSlab* z = (ZSlab*)p;
  hipStream_t st = ctx->stream;
  zc* a = (zc*)spec;
  const int TB = 64;
  dim3 b(TB, 1, 1), g((unsigned)((z->ncol + TB - 1) / TB), (unsigned)z->SZ, 1), g1((unsigned)((z->ncol + TB - 1) / TB), 1, 1);
  {
    ProfScope ps(ctx, "spectral_solve");
    const zc* bel = (const zc*)below;
    ocn_launch(k_zslab_getcol, dim3((z->n + 1 + 63) / 64), dim3(64), st, (const zc*)a, z->ncol, 0L, z->n, z->send + 2 * z->ncol, bel);
    ocn_launch(k_zslab_up, g, b, st, a, z->ncol, z->m, z->SZ, (const double*)z->lxy, dz2, z->segs, z->m > ZS_MR ? 1 : 0);
    ocn_launch(k_zslab_ranksums, g1, b, st, z->ncol, z->m, z->SZ, (const double*)z->lxy, dz2, (const zc*)z->segs, z->send, bel);
  }
  // all-gather (the same message to every peer)
  {
    ProfScope ps(ctx, "transpose");
    std::vector<CommOp> sends, recvs;
    const size_t bytes = z->msg * sizeof(zc);
    for (int q = 0; q < z->R; ++q) {
      sends.push_back({z->send, bytes, q, 2000});
      recvs.push_back({z->gathered + z->msg * q, bytes, q, 2000});
    }
    int rc = comm_exchange(ctx, sends, recvs);
    if (rc) return rc;
  }
  {
    ProfScope ps(ctx, "spectral_solve");
    if (z->m > ZS_MR)
      ocn_launch(k_zslab_down, g, b, st, a, z->ncol, z->m, z->SZ, z->R, z->rank, (const double*)z->lxy, dz2, scale * dz2,
                 (const zc*)z->segs, (const zc*)z->gathered, z->msg, (const zc*)below);
    else
      ocn_launch(k_zslab_down_reg<ZS_MR>, g, b, st, a, z->ncol, z->m, z->SZ, z->R, z->rank, (const double*)z->lxy, dz2,
                 scale * dz2, (const zc*)z->segs, (const zc*)z->gathered, z->msg, (const zc*)below);
    if (phi_below)   // after the sweeps: phi_below may be the very plane `below` lives in (each thread reads its entry, then writes it)
      ocn_launch(k_zslab_below, g1, b, st, z->ncol, z->m, z->SZ, z->R, z->rank, (const double*)z->lxy, dz2, scale * dz2,
                 (const zc*)z->segs, (const zc*)z->gathered, z->msg, (const zc*)below, (zc*)phi_below);
    ocn_launch_sync(k_zslab_mean, dim3(1), dim3(ZM_T), st, a, z->ncol, 0L, z->n, z->R, z->rank, (const zc*)z->gathered, z->msg,
                    scale * dz2, z->work, (zc*)phi_below);
  }
  return OCN_OK;
}
