// zfft.hip -- fused z-direction stage of the triply-periodic Poisson solve.
//
// Reference: Solvers/fft_based_poisson_solver.jl:93-120 applies forward transforms, divides by the
// eigenvalues, zeroes the mean mode and applies backward transforms as separate full-array passes.  Here the
// z-transform, the eigenvalue division and the inverse z-transform are ONE kernel: a workgroup owns 16
// consecutive (kx,ky) columns of the half spectrum (256 contiguous bytes per z-level, 256-B aligned because
// the column index is flattened), keeps every column's 256 z-points on chip, and touches HBM exactly once
// each way:  16 B/point read + 16 B/point written, instead of 5 passes (80 B/point) with library FFTs.
//
// 256-point FFT = four-step 16 x 16: each thread transforms 16 points in registers (radix-4 x radix-4),
// twiddles, one LDS transpose, second 16-point transform.  After the forward transform thread (col, k1) holds
// the frequencies kz = k1 + 16 k2, multiplies them by -norm / (lx + ly + lz[kz]) and runs the same network
// backwards, so the data never leaves registers/LDS in spectral space.  128 points: 16 x 8 (two 8-point transforms per
// thread in the second stage); 512 points: a radix-2 decimation-in-frequency step across two half-teams in front of the
// 256-point network.  The same three sizes serve the x pass (fused with the right-hand side) and the y passes.
#include "internal.h"

struct cd {
  double x, y;
};
OCN_DEVFN cd cadd(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
OCN_DEVFN cd csub(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
OCN_DEVFN cd cmul(cd a, cd b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// multiply by -i*S (S = +1 forward e^{-i..}, -1 inverse): forward: (x,y)*(-i) = (y,-x)
template <int S> OCN_DEVFN cd mul_mi(cd a) { return S > 0 ? cd{a.y, -a.x} : cd{-a.y, a.x}; }

// 4-point DFT of (a,b,c,d) with sign S: outputs X0..X3
template <int S> OCN_DEVFN void dft4(cd& a, cd& b, cd& c, cd& d) {
  cd s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = mul_mi<S>(csub(b, d));
  a = cadd(s0, s2);
  c = csub(s0, s2);
  b = cadd(s1, s3);
  d = csub(s1, s3);
}

// in-place 16-point DFT, natural order in and out.  X[k] = sum_n x[n] exp(-S 2 pi i n k / 16)
template <int S> OCN_DEVFN void dft16(cd* v) {
  // n = a + 4 b ; first DFT over b for each a  ->  y[a][c], c = 0..3 stored at v[a + 4 c]
#pragma unroll
  for (int a = 0; a < 4; ++a) dft4<S>(v[a], v[a + 4], v[a + 8], v[a + 12]);
  // twiddles W16^(a c)
  const double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, R = 0.70710678118654752440;
  auto tw = [&](cd z, double c, double s) { return cd{z.x * c + S * z.y * s, z.y * c - S * z.x * s}; };  // z * (c - i S s)
  v[1 + 4] = tw(v[1 + 4], C1, S1);    // a=1,c=1: W^1
  v[1 + 8] = tw(v[1 + 8], R, R);      // a=1,c=2: W^2
  v[1 + 12] = tw(v[1 + 12], S1, C1);  // a=1,c=3: W^3
  v[2 + 4] = tw(v[2 + 4], R, R);      // a=2,c=1: W^2
  v[2 + 8] = mul_mi<S>(v[2 + 8]);     // a=2,c=2: W^4 = -i
  v[2 + 12] = tw(v[2 + 12], -R, R);   // a=2,c=3: W^6
  v[3 + 4] = tw(v[3 + 4], S1, C1);    // a=3,c=1: W^3
  v[3 + 8] = tw(v[3 + 8], -R, R);     // a=3,c=2: W^6
  v[3 + 12] = tw(v[3 + 12], -C1, -S1);  // a=3,c=3: W^9
  // k = c + 4 d ; DFT over a for each c: inputs v[0+4c..3+4c] -> X[c + 4 d] stored at v[4c + d]
#pragma unroll
  for (int c = 0; c < 4; ++c) dft4<S>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
  // reorder so that v[k] = X[k]: currently X[c + 4 d] sits at v[4 c + d]  (a 4x4 transpose)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = c + 1; d < 4; ++d) {
      cd t = v[4 * c + d];
      v[4 * c + d] = v[4 * d + c];
      v[4 * d + c] = t;
    }
}

// in-place 8-point DFT, natural order in and out (radix 2 x 4)
template <int S> OCN_DEVFN void dft8(cd* v) {
  cd e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4<S>(e0, e1, e2, e3);
  dft4<S>(o0, o1, o2, o3);
  const double R = 0.70710678118654752440;
  auto tw = [&](cd z, double c, double s) { return cd{z.x * c + S * z.y * s, z.y * c - S * z.x * s}; };   // z * (c - i S s)
  o1 = tw(o1, R, R);
  o2 = mul_mi<S>(o2);
  o3 = tw(o3, -R, R);
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

// ---- N-point transforms (N = 128, 256, 512) by teams of threads of a 256-thread workgroup, 16 points per thread ---------
// `tw`: table of exp(-2 pi i m / 512), m = 0..511 (stride 512 / N gives the N-th roots).  LDS: SMN doubles.
//   N = 256: four-step 16 x 16, team of 16 threads (r = n2 on the way in, k1 on the way out): v[k2] = X[r + 16 k2]
//   N = 128: four-step 16 x 8, team of 8: the second stage is two 8-point transforms per thread:
//            v[k2 + 8 h] = X[(r + 8 h) + 16 k2]
//   N = 512: one radix-2 decimation-in-frequency step across two half-teams (h = 0: x[n] + x[n + 256], h = 1:
//            (x[n] - x[n + 256]) W512^n), then each half-team runs the 256-point network:
//            v[k2] = X[2 (r + 16 k2) + h].  The partner's 16 points travel through LDS.
// Columns per workgroup: 32 (N = 128), 16 (256), 8 (512); `c` is the column slot of the thread INCLUDING the half for
// N = 512 (c = col + 8 h), so the 256-point network below serves both.
#define SMN (16 * (256 + 32))
template <int N> struct FftGeo {
  static constexpr int M = N / 16;                   // threads per transform
  static constexpr int C = 256 / M;                  // columns per workgroup
  static constexpr int CS = N == 512 ? 16 : C;       // column slots of the LDS images (N = 512: column + 8 half)
  static constexpr int RS = N == 128 ? 8 : 16;       // threads per team in the LDS images
  static constexpr int K1S = RS * CS + CS;           // padded stride of the k1 index: rows of different r hit different banks
};

// all threads of the workgroup take part (barriers inside)
// XL: the x pass, whose lanes run over r first (the transformed direction is the contiguous one): image [k1][c][r] with an
// odd k1 stride, conflict-free for its writes (r consecutive) and its reads (k1 = r consecutive: stride 257 words)
template <int N, int S, bool XL = false> OCN_DEVFN void fft_stage2(double* sm, cd* v, int r, int c, const cd* tw) {
  typedef FftGeo<N> G;
  constexpr int KX = G::RS * G::CS + 1;
  constexpr int TS = 512 / (N == 512 ? 256 : N);     // table stride of the inner transform's roots
  // twiddles W^(r k1) of the inner transform (n2 = r)
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) {
    cd w = tw[((r * k1) * TS) & 511];
    if (S < 0) w.y = -w.y;
    v[k1] = cmul(v[k1], w);
  }
  // transpose: element (k1, n2 = r) -> the thread(s) that own k1
  auto wi = [&](int k1) { return XL ? k1 * KX + c * G::RS + r : k1 * G::K1S + r * G::CS + c; };
  if (N == 128) {
    // thread r takes k1 = r and r + 8: reads n2 = 0..7 of each
    auto ri = [&](int q) {
      return XL ? (r + 8 * (q >> 3)) * KX + c * G::RS + (q & 7) : (r + 8 * (q >> 3)) * G::K1S + (q & 7) * G::CS + c;
    };
#pragma unroll
    for (int q = 0; q < 16; ++q) sm[wi(q)] = v[q].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].x = sm[ri(q)];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) sm[wi(q)] = v[q].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].y = sm[ri(q)];
    dft8<S>(v);
    dft8<S>(v + 8);
  } else {
    auto ri = [&](int n2) { return XL ? r * KX + c * G::RS + n2 : r * G::K1S + n2 * G::CS + c; };
#pragma unroll
    for (int q = 0; q < 16; ++q) sm[wi(q)] = v[q].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].x = sm[ri(q)];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) sm[wi(q)] = v[q].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].y = sm[ri(q)];
    dft16<S>(v);
  }
}

// the 16 points of the partner thread (same column and r, other half) through LDS; barriers inside
OCN_DEVFN void fft512_partner(double* sm, const cd* v, cd* p, int r, int c) {
  const int me = r * 16 + c, other = r * 16 + (c ^ 8);          // [q][r][c]: consecutive lanes, consecutive words
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) sm[q * 256 + me] = v[q].x;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) p[q].x = sm[q * 256 + other];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) sm[q * 256 + me] = v[q].y;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) p[q].y = sm[q * 256 + other];
  __syncthreads();
}

// forward / backward N-point transform of the team's points.  In: v[n1] = x[r + M' n1 (+ 256 h)] with M' = 8 (N = 128)
// or 16; out: see the table above.  Unnormalised; S = +1: exp(-i ...), S = -1: exp(+i ...).
// XLANE: the x pass -- real input, and its thread layout puts the partner 16 lanes away in the same wave: the partner's
// 16 real values come by ds_bpermute instead of five barriers around an LDS image
template <int N, int S, bool XLANE = false> OCN_DEVFN void fft_fwd(double* sm, cd* v, int r, int c, const cd* tw) {
  if (N == 512) {
    cd p[16];
    if (XLANE) {
#pragma unroll
      for (int q = 0; q < 16; ++q) p[q] = cd{ocn_shfl_xor16(v[q].x), 0.0};
    } else fft512_partner(sm, v, p, r, c);
    const int h = (c >> 3) & 1;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      if (h == 0) v[n1] = cadd(v[n1], p[n1]);                    // x[n] + x[n + 256]
      else {
        cd w = tw[(r + 16 * n1) & 511];                          // W512^n, n = r + 16 n1
        if (S < 0) w.y = -w.y;
        v[n1] = cmul(csub(p[n1], v[n1]), w);                     // (x[n] - x[n + 256]) W^n: own = x[n + 256]
      }
    }
  }
  dft16<S>(v);
  fft_stage2<N, S, XLANE>(sm, v, r, c, tw);
}

// the same network run backwards: from the spectral layout of fft_fwd<N, +1> back to v[n1] = x[r + M' n1 (+ 256 h)],
// unnormalised (x N)
template <int N> OCN_DEVFN void fft_back(double* sm, cd* v, int r, int c, const cd* tw) {
  typedef FftGeo<N> G;
  constexpr int TS = 512 / (N == 512 ? 256 : N);
  if (N == 128) {
    dft8<-1>(v);
    dft8<-1>(v + 8);                                             // v[n2 + 8 h] = Z[k1 = r + 8 h][n2]
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      cd w = tw[(((r + 8 * (q >> 3)) * (q & 7)) * TS) & 511];    // conj W128^(k1 n2)
      w.y = -w.y;
      v[q] = cmul(v[q], w);
    }
    auto wi = [&](int q) { return (r + 8 * (q >> 3)) * G::K1S + (q & 7) * G::CS + c; };
    auto ri = [&](int k1) { return k1 * G::K1S + r * G::CS + c; };
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) sm[wi(q)] = v[q].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].x = sm[ri(q)];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) sm[wi(q)] = v[q].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].y = sm[ri(q)];
    dft16<-1>(v);
    return;
  }
  dft16<-1>(v);                                                  // v[n2] = Z[k1 = r][n2]
#pragma unroll
  for (int n2 = 1; n2 < 16; ++n2) {
    cd w = tw[((r * n2) * TS) & 511];
    w.y = -w.y;
    v[n2] = cmul(v[n2], w);
  }
  auto wi = [&](int n2) { return r * G::K1S + n2 * G::CS + c; };
  auto ri = [&](int k1) { return k1 * G::K1S + r * G::CS + c; };
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) sm[wi(q)] = v[q].x;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q].x = sm[ri(q)];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) sm[wi(q)] = v[q].y;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q].y = sm[ri(q)];
  dft16<-1>(v);                                                  // r = n2 again: v[n1] = u_h[r + 16 n1]
  if (N == 512) {
    // undo the decimation step: u0 = x[n] + x[n+256], u1 = (x[n] - x[n+256]) W^n  ->  2 x[n] = u0 + u1 conj(W^n), ...
    const int h = (c >> 3) & 1;
    if (h == 1) {
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        cd w = tw[(r + 16 * n1) & 511];
        w.y = -w.y;
        v[n1] = cmul(v[n1], w);
      }
    }
    cd p[16];
    fft512_partner(sm, v, p, r, c);
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = h == 0 ? cadd(v[n1], p[n1]) : csub(p[n1], v[n1]);   // 2 x[n], 2 x[n + 256]
  }
}

// thread -> (r, column slot c, element index of v[q] along the transformed direction) for data whose COLUMNS are adjacent
// in memory (y and z passes): t = c + CS * r
template <int N> OCN_DEVFN void team_of(int t, int& r, int& c, int& col) {
  typedef FftGeo<N> G;
  c = t % G::CS;
  r = t / G::CS;
  col = N == 512 ? (c & 7) : c;
}
// position along the transformed direction of v[n1] on the way in (physical side) ...
template <int N> OCN_DEVFN int pos_in(int r, int c, int n1) {
  return N == 128 ? r + 8 * n1 : N == 256 ? r + 16 * n1 : r + 16 * n1 + 256 * ((c >> 3) & 1);
}
// ... and of v[q] on the way out (spectral side)
template <int N> OCN_DEVFN int pos_out(int r, int c, int q) {
  return N == 128 ? (r + 8 * (q >> 3)) + 16 * (q & 7) : N == 256 ? r + 16 * q : 2 * (r + 16 * q) + ((c >> 3) & 1);
}

// ---- fused z stage: forward transform, eigenvalue division, backward transform in one pass ------------------------------
// `a`: (ncol, N) complex, element (col, z) at a[col + ncol * z].  lxy[col]: lx + ly of the column; lz[kz].
// zero_col: flattened column whose kz = 0 mode is set to 0 (or -1).  norm: 1 / (Nx Ny Nz).
template <int N>
__global__ void __launch_bounds__(256) k_zsolve(cd* __restrict__ a, long ncol, const double* __restrict__ lxy,
                                                const double* __restrict__ lz, const cd* __restrict__ tw, double norm, long zero_col) {
  OCN_SHARED double sm[SMN];
  int r, c, col;
  team_of<N>(threadIdx.x, r, c, col);
  const long gcol = (long)blockIdx.x * FftGeo<N>::C + col;
  const bool ok = gcol < ncol;
  cd v[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) v[n1] = ok ? a[gcol + ncol * pos_in<N>(r, c, n1)] : cd{0, 0};
  fft_fwd<N, 1>(sm, v, r, c, tw);
  // eigenvalue division (fft_based_poisson_solver.jl:106-111)
  const double lc = ok ? lxy[gcol] : 1.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int kz = pos_out<N>(r, c, q);
    double f = -norm / (lc + lz[kz]);
    if (gcol == zero_col && kz == 0) f = 0.0;
    v[q].x *= f;
    v[q].y *= f;
  }
  fft_back<N>(sm, v, r, c, tw);
  if (ok) {
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) a[gcol + ncol * pos_in<N>(r, c, n1)] = v[n1];
  }
}


// ---- fused z stage for Nz = 16 M with M = 20 or 24 (320 and 384 levels): mixed radix ----------------------------------------------
// Same structure as k_zsolve: a workgroup owns 16 consecutive columns (256 contiguous bytes per level), M threads per column hold
// 16 levels each (x[r + M n1]).  Stage 1: 16-point transform over n1 in registers, twiddle W_N^(r k1), transpose through LDS;
// stage 2: the 16 transforms of length M of a column are taken by its first 16 threads (k1 = r < 16; 4 of 20 / 8 of 24 threads idle for
// that stage), M points each in registers: 20 = 4 x 5 (five-point butterflies, twiddles, four-point butterflies), 24 = 8 x 3.  Eigenvalue
// division on X[k1 + 16 k2], then the network backwards.  `tw`: exp(-2 pi i m / N), m = 0..N-1.
template <int S> OCN_DEVFN void dft3(cd& a, cd& b, cd& c) {
  const double H3 = 0.86602540378443864676;                      // sin(2 pi / 3)
  const cd t1 = cadd(b, c), d = csub(b, c);
  const cd t2 = cd{a.x - 0.5 * t1.x, a.y - 0.5 * t1.y};
  const cd t3 = mul_mi<S>(cd{H3 * d.x, H3 * d.y});
  a = cadd(a, t1);
  b = cadd(t2, t3);
  c = csub(t2, t3);
}
template <int S> OCN_DEVFN void dft5(cd& x0, cd& x1, cd& x2, cd& x3, cd& x4) {
  const double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410;     // cos(2 pi / 5), cos(4 pi / 5)
  const double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;      // sin(2 pi / 5), sin(4 pi / 5)
  const cd t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
  const cd a1 = cd{x0.x + C1 * t1.x + C2 * t2.x, x0.y + C1 * t1.y + C2 * t2.y};
  const cd a2 = cd{x0.x + C2 * t1.x + C1 * t2.x, x0.y + C2 * t1.y + C1 * t2.y};
  const cd b1 = mul_mi<S>(cd{S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y});
  const cd b2 = mul_mi<S>(cd{S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y});
  x0 = cadd(x0, cadd(t1, t2));
  x1 = cadd(a1, b1);
  x4 = csub(a1, b1);
  x2 = cadd(a2, b2);
  x3 = csub(a2, b2);
}
// W_M^m with the sign of the transform, from the table of N-th roots (N = 16 M)
template <int M, int S> OCN_DEVFN cd rootM(const cd* tw, int m) {
  cd w = tw[(16 * m) % (16 * M)];
  if (S < 0) w.y = -w.y;
  return w;
}
// in-place M-point DFT, natural order in and out
template <int M, int S> OCN_DEVFN void dftM(cd* v, const cd* tw) {
  if (M == 20) {
    // n = a + 4 b, k = 5 c + d: five-point transforms over b, twiddles W20^(a d), four-point transforms over a
#pragma unroll
    for (int a = 0; a < 4; ++a) dft5<S>(v[a], v[a + 4], v[a + 8], v[a + 12], v[a + 16]);      // T[a][d] at v[a + 4 d]
#pragma unroll
    for (int a = 1; a < 4; ++a)
#pragma unroll
      for (int d = 1; d < 5; ++d) v[a + 4 * d] = cmul(v[a + 4 * d], rootM<M, S>(tw, a * d));
#pragma unroll
    for (int d = 0; d < 5; ++d) dft4<S>(v[4 * d], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3]);  // X[5 c + d] at v[c + 4 d]
    cd o[20];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int d = 0; d < 5; ++d) o[5 * c + d] = v[c + 4 * d];
#pragma unroll
    for (int q = 0; q < 20; ++q) v[q] = o[q];
  } else {
    // M = 24: n = a + 8 b, k = 3 c + d: three-point transforms over b, twiddles W24^(a d), eight-point transforms over a
#pragma unroll
    for (int a = 0; a < 8; ++a) dft3<S>(v[a], v[a + 8], v[a + 16]);                           // T[a][d] at v[a + 8 d]
#pragma unroll
    for (int a = 1; a < 8; ++a)
#pragma unroll
      for (int d = 1; d < 3; ++d) v[a + 8 * d] = cmul(v[a + 8 * d], rootM<M, S>(tw, a * d));
#pragma unroll
    for (int d = 0; d < 3; ++d) dft8<S>(v + 8 * d);                                           // X[3 c + d] at v[c + 8 d]
    cd o[24];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int d = 0; d < 3; ++d) o[3 * c + d] = v[c + 8 * d];
#pragma unroll
    for (int q = 0; q < 24; ++q) v[q] = o[q];
  }
}

template <int M>
__global__ void __launch_bounds__(16 * M) k_zsolve_mr(cd* __restrict__ a, long ncol, const double* __restrict__ lxy,
                                                      const double* __restrict__ lz, const cd* __restrict__ tw, double norm, long zero_col) {
  constexpr int N = 16 * M, K1S = 16 * M + 16;
  OCN_SHARED double sm[16 * K1S];
  const int c = threadIdx.x % 16, r = threadIdx.x / 16;            // r = n2 on the way in; the threads r < 16 take k1 = r in stage 2
  const long gcol = (long)blockIdx.x * 16 + c;
  const bool ok = gcol < ncol;
  cd v[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) v[n1] = ok ? a[gcol + ncol * (r + M * n1)] : cd{0, 0};
  dft16<1>(v);
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmul(v[k1], tw[(r * k1) % N]);
  // transpose (k1, n2 = r) -> thread k1: image [k1][n2][c], one component at a time
  cd z[M];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) sm[k1 * K1S + r * 16 + c] = v[k1].x;
  __syncthreads();
  if (r < 16) {
#pragma unroll
    for (int n2 = 0; n2 < M; ++n2) z[n2].x = sm[r * K1S + n2 * 16 + c];
  }
  __syncthreads();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) sm[k1 * K1S + r * 16 + c] = v[k1].y;
  __syncthreads();
  if (r < 16) {
#pragma unroll
    for (int n2 = 0; n2 < M; ++n2) z[n2].y = sm[r * K1S + n2 * 16 + c];
    dftM<M, 1>(z, tw);                                            // z[k2] = X[r + 16 k2]
    // eigenvalue division (fft_based_poisson_solver.jl:106-111)
    const double lc = ok ? lxy[gcol] : 1.0;
#pragma unroll
    for (int k2 = 0; k2 < M; ++k2) {
      const int kz = r + 16 * k2;
      double f = -norm / (lc + lz[kz]);
      if (gcol == zero_col && kz == 0) f = 0.0;
      z[k2].x *= f;
      z[k2].y *= f;
    }
    dftM<M, -1>(z, tw);                                           // z[n2] = sum over k2 (k1 = r fixed)
#pragma unroll
    for (int n2 = 1; n2 < M; ++n2) {
      cd w = tw[(r * n2) % N];
      w.y = -w.y;
      z[n2] = cmul(z[n2], w);
    }
  }
  __syncthreads();
  if (r < 16) {
#pragma unroll
    for (int n2 = 0; n2 < M; ++n2) sm[r * K1S + n2 * 16 + c] = z[n2].x;
  }
  __syncthreads();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) v[k1].x = sm[k1 * K1S + r * 16 + c];
  __syncthreads();
  if (r < 16) {
#pragma unroll
    for (int n2 = 0; n2 < M; ++n2) sm[r * K1S + n2 * 16 + c] = z[n2].y;
  }
  __syncthreads();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) v[k1].y = sm[k1 * K1S + r * 16 + c];
  dft16<-1>(v);                                                   // v[n1] = x[r + M n1], times N
  if (ok) {
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) a[gcol + ncol * (r + M * n1)] = v[n1];
  }
}

// ---- y-direction pass: in-place N-point FFT along ky of the half spectrum (Nxh, N, Nz) -------------------------------
// Tiles of C columns.  Main tiles: C consecutive kx of one z-plane (contiguous in memory for every ky).  The
// Nxh % C left-over kx columns are tiled over the flattened (kx_left, z) index.
template <int N, int S>
__global__ void __launch_bounds__(256) k_yfft(cd* __restrict__ a, int Nxh, int Nz, const cd* __restrict__ tw) {
  OCN_SHARED double sm[SMN];
  constexpr int C = FftGeo<N>::C;
  int r, c, col;
  team_of<N>(threadIdx.x, r, c, col);
  const int nfull = Nxh / C, left = Nxh - C * nfull;
  const long plane = (long)Nxh * N;
  const long nmain = (long)nfull * Nz;
  long base;
  bool ok = true;
  if ((long)blockIdx.x < nmain) {
    // A workgroup reads C consecutive kx (128 or 256 bytes) of every row, at 16-byte alignment: neighbouring kx tiles share
    // cache lines.  Workgroups b and b + 8 land on the same XCD (its own L2) right after each other, so the tiles are
    // handed out XCD-major: each L2 then fetches a shared line once instead of two XCDs fetching it each.
    long b = blockIdx.x;
    if (nmain % 8 == 0) b = (b % 8) * (nmain / 8) + b / 8;
    const int tile = (int)(b % nfull), zz = (int)(b / nfull);
    base = C * tile + col + plane * zz;
  } else {
    const long cc = ((long)blockIdx.x - nmain) * C + col;      // flattened (kx_left, z)
    ok = left > 0 && cc < (long)left * Nz;
    const long zz = ok ? cc / left : 0, kl = ok ? cc - zz * left : 0;
    base = C * nfull + kl + plane * zz;
  }
  cd v[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) v[n1] = ok ? a[base + (long)Nxh * pos_in<N>(r, c, n1)] : cd{0, 0};
  fft_fwd<N, S>(sm, v, r, c, tw);
  if (ok) {
#pragma unroll
    for (int q = 0; q < 16; ++q) a[base + (long)Nxh * pos_out<N>(r, c, q)] = v[q];
  }
}

// ---- x-direction forward pass fused with the Poisson right-hand side ----------------------------------------------
// rhs = div(U*) / dt (solve_for_pressure.jl:15-18) is formed on the fly from the predictor with periodic wrap
// indexing and transformed along x (N real points as a complex FFT with zero imaginary part); the half
// spectrum kx = 0..N/2 is written once.  One workgroup = C consecutive x-lines (flattened j + Ny k); here the
// TRANSFORMED direction is the contiguous one, so the team index r runs fastest over the lanes: t = r' + RT * line
// with r' = r (+ 16 h for N = 512).
template <int N, bool CO>
__global__ void __launch_bounds__(256) k_xfft_rhs(GridDev g, const double* __restrict__ us, const double* __restrict__ vs,
                                                  const double* __restrict__ ws, double rdt, int zwrap,
                                                  cd* __restrict__ spec, const cd* __restrict__ tw, int extra) {
  OCN_SHARED double sm[SMN];
  constexpr int M = FftGeo<N>::M, C = FftGeo<N>::C;
  const int t = threadIdx.x;
  const int rr = t % M, ln = t / M;                 // rr: position inside the team; ln: line inside the tile
  const int r = N == 512 ? rr & 15 : rr;
  const int c = N == 512 ? ln + 8 * (rr >> 4) : ln;  // column slot (line + 8 half)
  const long L = (long)blockIdx.x * C + ln;         // line index j + Ny k
  // extra != 0 (z-slab runs, round 3): Ny more lines behind the slab's own -- "plane Nz" of the spectrum is the x transform of
  // w*[level 0] / (dz dt), the term the divergence of the level BELOW this slab lacks (its owner forms it with w* above left at
  // zero); it enters the lower levels' solve through this rank's Green's-function sums (zslab.hip `bel`)
  const long nlines = (long)g.Ny * (g.Nz + (extra ? 1 : 0));
  const bool ok = L < nlines;
  const long sy = g.sy, sz = g.sz;
  const double rdz = 1.0 / g.dz;
  cd v[16];
  if (CO) {
    // The divergence is formed in the order of the memory: a wave reads 64 consecutive cells of a line (the team order
    // r + 16 n1 of the transform makes every load instruction touch four lines in 128-byte pieces, which held this kernel
    // at 3.6 TB/s where k_rhs + a separate transform ran at 5), lands in LDS, and the teams pick their points up there.
    constexpr int LP = N + 16;                       // line pitch in LDS: consecutive lines start on different bank halves
    static_assert(C * LP <= SMN, "divergence tile must fit the transform's LDS buffer");
    constexpr int PER = C * N / 256;                 // cells per thread
    const long L0 = (long)blockIdx.x * C;
    int j0 = (int)(L0 % g.Ny), k0 = (int)(L0 / g.Ny);
#pragma unroll
    for (int m = 0; m < PER; ++m) {
      const int e = t + 256 * m;
      const int l2 = e / N, i = e - l2 * N;          // N is a power of two: shifts; l2 is wave-uniform
      int j = j0 + l2, k = k0;
      while (j >= g.Ny) { j -= g.Ny; ++k; }
      double d = 0.0;
      if (extra && k == g.Nz) d = ws[j * sy + i] * rdz * rdt;      // the extra plane: w* of level 0 over dz dt
      else if (k < g.Nz) {
        const long row = j * sy + k * sz;
        const long rown = ((j + 1 == g.Ny) ? 0 : j + 1) * sy + k * sz;
        const long rowt = (zwrap && k + 1 == g.Nz) ? j * sy : j * sy + (k + 1) * sz;
        const int ie = (i + 1 == g.Nx) ? 0 : i + 1;
        if (g.zb) {   // Bounded z (Fourier-tridiagonal solver): k_rhs's expression, term for term -- div with / dz^c[k], times rdt, times dz^c[k]
          const double dzc = g_dzc(g, k);
          d = ((us[row + ie] - us[row + i]) * g.rdx + (vs[rown + i] - vs[row + i]) * g.rdy + (ws[rowt + i] - ws[row + i]) / dzc) * rdt * dzc;
        } else
        d = ((us[row + ie] - us[row + i]) * g.rdx + (vs[rown + i] - vs[row + i]) * g.rdy + (ws[rowt + i] - ws[row + i]) * rdz) * rdt;
      }
      sm[l2 * LP + i] = d;
    }
    __syncthreads();
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = {sm[ln * LP + pos_in<N>(r, c, n1)], 0.0};
    __syncthreads();                                 // the transform reuses the buffer
  } else {
    const int k = ok ? (int)(L / g.Ny) : 0, j = ok ? (int)(L - (long)k * g.Ny) : 0;
    const long row = j * sy + k * sz;
    const long rown = ((j + 1 == g.Ny) ? 0 : j + 1) * sy + k * sz;
    const long rowt = (zwrap && k + 1 == g.Nz) ? j * sy : j * sy + (k + 1) * sz;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int i = pos_in<N>(r, c, n1);
      const int ie = (i + 1 == g.Nx) ? 0 : i + 1;
      double d = 0.0;
      if (ok && extra && k == g.Nz) d = ws[j * sy + i] * rdz * rdt;
      else if (ok && g.zb) {
        const double dzc = g_dzc(g, k);
        d = ((us[row + ie] - us[row + i]) * g.rdx + (vs[rown + i] - vs[row + i]) * g.rdy + (ws[rowt + i] - ws[row + i]) / dzc) * rdt * dzc;
      } else if (ok) d = ((us[row + ie] - us[row + i]) * g.rdx + (vs[rown + i] - vs[row + i]) * g.rdy + (ws[rowt + i] - ws[row + i]) * rdz) * rdt;
      v[n1] = {d, 0.0};
    }
  }
  fft_fwd<N, 1, true>(sm, v, r, c, tw);
  if (ok) {
    cd* out = spec + L * (N / 2 + 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      // smallest wavenumber any thread holds in v[q]: registers that only hold kx > N/2 store nothing (folded at compile time)
      const int kmin = N == 128 ? 8 * (q >> 3) + 16 * (q & 7) : N == 256 ? 16 * q : 32 * q;
      if (kmin > N / 2) continue;
      const int kx = pos_out<N>(r, c, q);
      if (kx <= N / 2) out[kx] = v[q];
    }
  }
}

// host side ---------------------------------------------------------------------------------------------------
struct ZSolve {
  cd* twmr = nullptr;      // exp(-2 pi i m / N), N = 320 or 384: the mixed-radix z stage (created on first use)
  int twmr_n = 0;
  cd* tw = nullptr;        // exp(-2 pi i m / 512), m = 0..511
  double* lxy = nullptr;
  long ncol = 0;
};

bool fft_size_ok(int n) { return n == 128 || n == 256 || n == 512; }
// sizes the fused z stage serves: the three above and 16 x 20, 16 x 24 (mixed radix)
bool zsolve_size_ok(int n) { return fft_size_ok(n) || n == 320 || n == 384; }

void* zsolve_create(ocn_ctx* ctx, const std::vector<double>& lx_half, const std::vector<double>& ly_local) {
  // lxy[kx + Nxh * ky]
  ZSolve* z = new ZSolve;
  const size_t Nxh = lx_half.size(), Ny = ly_local.size();
  z->ncol = (long)(Nxh * Ny);
  std::vector<double> lxy(Nxh * Ny);
  for (size_t j = 0; j < Ny; ++j)
    for (size_t i = 0; i < Nxh; ++i) lxy[i + Nxh * j] = lx_half[i] + ly_local[j];
  std::vector<cd> tw(512);
  for (int m = 0; m < 512; ++m) tw[m] = {cos(2.0 * M_PI * m / 512.0), -sin(2.0 * M_PI * m / 512.0)};
  if (hipMalloc((void**)&z->tw, sizeof(cd) * 512) != hipSuccess ||
      hipMalloc((void**)&z->lxy, sizeof(double) * lxy.size()) != hipSuccess) {
    ocn_set_error(ctx, "zsolve: allocation failed");
    delete z;
    return nullptr;
  }
  hipMemcpy(z->tw, tw.data(), sizeof(cd) * 512, hipMemcpyHostToDevice);
  hipMemcpy(z->lxy, lxy.data(), sizeof(double) * lxy.size(), hipMemcpyHostToDevice);
  return z;
}

void zsolve_destroy(void* p) {
  ZSolve* z = (ZSolve*)p;
  if (!z) return;
  hipFree(z->twmr);
  hipFree(z->tw);
  hipFree(z->lxy);
  delete z;
}

#define FFT_BY_N(N, CALL128, CALL256, CALL512) \
  switch (N) { case 128: CALL128; break; case 256: CALL256; break; default: CALL512; break; }

// forward / inverse y pass (in place), and the fused rhs + x pass; all unnormalised
void yfft_run(ocn_ctx* ctx, void* p, void* spec, int Nxh, int Ny, int Nz, int inverse) {
  ZSolve* z = (ZSolve*)p;
  const int C = 256 / (Ny / 16);
  const int nfull = Nxh / C, left = Nxh - C * nfull;
  const long nblk = (long)nfull * Nz + ((long)left * Nz + C - 1) / C;
  dim3 b(256, 1, 1), g((unsigned)nblk, 1, 1);
  cd* a = (cd*)spec;
  const cd* tw = (const cd*)z->tw;
  hipStream_t s = ctx->stream;
  if (inverse) {
    FFT_BY_N(Ny, ocn_launch_sync(k_yfft<128, -1>, g, b, s, a, Nxh, Nz, tw), ocn_launch_sync(k_yfft<256, -1>, g, b, s, a, Nxh, Nz, tw),
             ocn_launch_sync(k_yfft<512, -1>, g, b, s, a, Nxh, Nz, tw))
  } else {
    FFT_BY_N(Ny, ocn_launch_sync(k_yfft<128, 1>, g, b, s, a, Nxh, Nz, tw), ocn_launch_sync(k_yfft<256, 1>, g, b, s, a, Nxh, Nz, tw),
             ocn_launch_sync(k_yfft<512, 1>, g, b, s, a, Nxh, Nz, tw))
  }
}

void xfft_rhs_run(ocn_model* m, void* p, void* spec, double dt, int extra_plane) {
  ZSolve* z = (ZSolve*)p;
  const GridDev& g = m->gd;
  const long nlines = (long)g.Ny * (g.Nz + (extra_plane ? 1 : 0));
  const int ex = extra_plane;
  const int C = 256 / (g.Nx / 16);
  dim3 b(256, 1, 1), gr((unsigned)((nlines + C - 1) / C), 1, 1);
  // the predictor: us / vs / ws on the all-in-one path and between update and projection of the tiled Bounded-z path, else u, v, w
  const double* us = (m->fast_path ? m->us : pred_u(m)).interior();
  const double* vs = (m->fast_path ? m->vs : pred_v(m)).interior();
  const double* ws = (m->fast_path ? m->ws : pred_w(m)).interior();
  const int zw = (m->g->dist || m->g->topo[2] != OCN_PERIODIC) ? 0 : 1;
  hipStream_t s = m->ctx->stream;
  cd* sp = (cd*)spec;
  const cd* tw = (const cd*)z->tw;
  if (m->knob_xfft_team) {
    FFT_BY_N(g.Nx, ocn_launch_sync(k_xfft_rhs<128, false>, gr, b, s, g, us, vs, ws, 1.0 / dt, zw, sp, tw, ex),
             ocn_launch_sync(k_xfft_rhs<256, false>, gr, b, s, g, us, vs, ws, 1.0 / dt, zw, sp, tw, ex),
             ocn_launch_sync(k_xfft_rhs<512, false>, gr, b, s, g, us, vs, ws, 1.0 / dt, zw, sp, tw, ex))
  } else {
    FFT_BY_N(g.Nx, ocn_launch_sync(k_xfft_rhs<128, true>, gr, b, s, g, us, vs, ws, 1.0 / dt, zw, sp, tw, ex),
             ocn_launch_sync(k_xfft_rhs<256, true>, gr, b, s, g, us, vs, ws, 1.0 / dt, zw, sp, tw, ex),
             ocn_launch_sync(k_xfft_rhs<512, true>, gr, b, s, g, us, vs, ws, 1.0 / dt, zw, sp, tw, ex))
  }
}

// in place on the (ncol, Nz) spectrum; `zero_col` < 0 when this rank does not own the mean mode
void zsolve_run(ocn_ctx* ctx, void* p, void* spec, int Nz, const double* lz, double norm, long zero_col) {
  ZSolve* z = (ZSolve*)p;
  if (Nz == 320 || Nz == 384) {
    if (z->twmr_n != Nz) {
      std::vector<cd> t(Nz);
      for (int m = 0; m < Nz; ++m) t[m] = {cos(2.0 * M_PI * m / Nz), -sin(2.0 * M_PI * m / Nz)};
      hipStreamSynchronize(ctx->stream);
      hipFree(z->twmr);
      z->twmr = nullptr;
      z->twmr_n = 0;
      if (hipMalloc((void**)&z->twmr, sizeof(cd) * Nz) != hipSuccess) {
        ocn_set_error(ctx, "zsolve: allocation failed");
        return;
      }
      hipMemcpy(z->twmr, t.data(), sizeof(cd) * Nz, hipMemcpyHostToDevice);
      z->twmr_n = Nz;
    }
    dim3 g((unsigned)((z->ncol + 15) / 16), 1, 1);
    cd* a = (cd*)spec;
    if (Nz == 320)
      ocn_launch_sync(k_zsolve_mr<20>, g, dim3(320, 1, 1), ctx->stream, a, z->ncol, (const double*)z->lxy, lz, (const cd*)z->twmr, norm, zero_col);
    else
      ocn_launch_sync(k_zsolve_mr<24>, g, dim3(384, 1, 1), ctx->stream, a, z->ncol, (const double*)z->lxy, lz, (const cd*)z->twmr, norm, zero_col);
    return;
  }
  const int C = 256 / (Nz / 16);
  dim3 b(256, 1, 1), g((unsigned)((z->ncol + C - 1) / C), 1, 1);
  cd* a = (cd*)spec;
  const cd* tw = (const cd*)z->tw;
  hipStream_t s = ctx->stream;
  FFT_BY_N(Nz, ocn_launch_sync(k_zsolve<128>, g, b, s, a, z->ncol, (const double*)z->lxy, lz, tw, norm, zero_col),
           ocn_launch_sync(k_zsolve<256>, g, b, s, a, z->ncol, (const double*)z->lxy, lz, tw, norm, zero_col),
           ocn_launch_sync(k_zsolve<512>, g, b, s, a, z->ncol, (const double*)z->lxy, lz, tw, norm, zero_col))
}
