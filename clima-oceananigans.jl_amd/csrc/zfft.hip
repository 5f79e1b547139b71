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
// backwards, so the data never leaves registers/LDS in spectral space.
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

// 16 x 16 x 16 transpose through LDS in two halves (real parts, then imaginary parts): 32 KB per workgroup
// instead of 64, so four workgroups fit a CU and twice as many loads are in flight (these passes are
// HBM-latency bound).  wi(q) / ri(q): element index written / read for register q.
template <class WI, class RI>
OCN_DEVFN void transpose_halves(double* sm, cd* v, WI wi, RI ri) {
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
}

// `a`: (ncol, 256) complex, element (col, z) at a[col + ncol * z].  lxy[col]: lx + ly of the column.
// lz[kz]; tw256[m] = exp(-2 pi i m / 256).  zero_col: flattened column whose kz = 0 mode is set to 0 (or -1).
__global__ void __launch_bounds__(256) k_zsolve256(cd* __restrict__ a, long ncol, const double* __restrict__ lxy,
                                                   const double* __restrict__ lz, const cd* __restrict__ tw256,
                                                   double norm, long zero_col) {
  OCN_SHARED cd sm[16 * 16 * 16];           // [p][q][col], column fastest: conflict-free b128 writes and reads
  const int t = threadIdx.x;
  const int col = t & 15, r = t >> 4;       // r plays n2 (loads / stores) and k1 (spectral side)
  const long gcol = (long)blockIdx.x * 16 + col;
  const bool ok = gcol < ncol;
  cd v[16];
  // ---- load x[n2 + 16 n1], n1 = 0..15 (each wave instruction: four 256-byte rows) ----
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) v[n1] = ok ? a[gcol + ncol * (r + 16 * n1)] : cd{0, 0};
  // ---- forward: DFT over n1 -> Y[n2][k1]; twiddle W256^(n2 k1); transpose; DFT over n2 ----
  dft16<1>(v);
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmul(v[k1], tw256[(r * k1) & 255]);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) sm[(k1 * 16 + r) * 16 + col] = v[k1];
  __syncthreads();
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) v[n2] = sm[(r * 16 + n2) * 16 + col];   // now r = k1
  dft16<1>(v);                                                            // v[k2] = X[k1 + 16 k2]
  // ---- eigenvalue division (fft_based_poisson_solver.jl:106-111) ----
  const double lc = ok ? lxy[gcol] : 1.0;
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) {
    const int kz = r + 16 * k2;
    double f = -norm / (lc + lz[kz]);
    if (gcol == zero_col && kz == 0) f = 0.0;
    v[k2].x *= f;
    v[k2].y *= f;
  }
  // ---- inverse: DFT(+) over k2 -> Z[k1][n2]; twiddle conj W256^(n2 k1); transpose; DFT(+) over k1 ----
  dft16<-1>(v);
#pragma unroll
  for (int n2 = 1; n2 < 16; ++n2) {
    cd w = tw256[(r * n2) & 255];
    w.y = -w.y;
    v[n2] = cmul(v[n2], w);
  }
  __syncthreads();                           // everyone finished reading the forward transpose
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) sm[(n2 * 16 + r) * 16 + col] = v[n2];   // r = k1
  __syncthreads();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) v[k1] = sm[(r * 16 + k1) * 16 + col];   // now r = n2
  dft16<-1>(v);                                                           // v[n1] = x[n2 + 16 n1]
  if (ok) {
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) a[gcol + ncol * (r + 16 * n1)] = v[n1];
  }
}

// ---- y-direction pass: in-place 256-point FFT along ky of the half spectrum (Nxh, 256, Nz) -------------------
// Tiles of 16 columns.  Main tiles: 16 consecutive kx of one z-plane (256 contiguous bytes per ky).  The
// Nxh % 16 left-over kx columns are tiled over the flattened (kx_left, z) index.
template <int S>
__global__ void __launch_bounds__(256) k_yfft256(cd* __restrict__ a, int Nxh, int Nz, const cd* __restrict__ tw256) {
  OCN_SHARED double sm[16 * 16 * 16];
  const int t = threadIdx.x;
  const int col = t & 15, r = t >> 4;
  const int nfull = Nxh / 16, left = Nxh - 16 * nfull;
  const long plane = (long)Nxh * 256;
  const long nmain = (long)nfull * Nz;
  long base;
  bool ok = true;
  if ((long)blockIdx.x < nmain) {
    const int tile = blockIdx.x % nfull, zz = blockIdx.x / nfull;
    base = 16 * tile + col + plane * zz;
  } else {
    const long c = ((long)blockIdx.x - nmain) * 16 + col;      // flattened (kx_left, z)
    ok = left > 0 && c < (long)left * Nz;
    const long zz = ok ? c / left : 0, kl = ok ? c - zz * left : 0;
    base = 16 * nfull + kl + plane * zz;
  }
  cd v[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) v[n1] = ok ? a[base + (long)Nxh * (r + 16 * n1)] : cd{0, 0};
  dft16<S>(v);
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) {
    cd w = tw256[(r * k1) & 255];
    if (S < 0) w.y = -w.y;
    v[k1] = cmul(v[k1], w);
  }
  transpose_halves(sm, v, [&](int k1) { return (k1 * 16 + r) * 16 + col; }, [&](int n2) { return (r * 16 + n2) * 16 + col; });
  dft16<S>(v);                                                            // r = k1 now; v[k2] = X[k1 + 16 k2]
  if (ok) {
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) a[base + (long)Nxh * (r + 16 * k2)] = v[k2];
  }
}

// ---- x-direction forward pass fused with the Poisson right-hand side ----------------------------------------------
// rhs = div(U*) / dt (solve_for_pressure.jl:15-18) is formed on the fly from the predictor with periodic wrap
// indexing and transformed along x (256 real points as a complex FFT with zero imaginary part); the half
// spectrum kx = 0..128 is written once.  One workgroup = 16 consecutive x-lines (flattened j + Ny k).
__global__ void __launch_bounds__(256) k_xfft_rhs256(GridDev g, const double* __restrict__ us, const double* __restrict__ vs,
                                                     const double* __restrict__ ws, double rdt, int zwrap,
                                                     cd* __restrict__ spec, const cd* __restrict__ tw256) {
  OCN_SHARED double sm[16 * 16 * 16];
  const int t = threadIdx.x;
  const int r = t & 15, ln = t >> 4;              // r = x mod 16 (loads) / k1 (stores); ln = line inside the tile
  const long L = (long)blockIdx.x * 16 + ln;      // line index j + Ny k
  const long nlines = (long)g.Ny * g.Nz;
  const bool ok = L < nlines;
  const int k = ok ? (int)(L / g.Ny) : 0, j = ok ? (int)(L - (long)k * g.Ny) : 0;
  const long sy = g.sy, sz = g.sz;
  const long row = j * sy + k * sz;
  const long rown = ((j + 1 == g.Ny) ? 0 : j + 1) * sy + k * sz;
  const long rowt = (zwrap && k + 1 == g.Nz) ? j * sy : j * sy + (k + 1) * sz;
  const double rdz = 1.0 / g.dz;
  cd v[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    const int i = r + 16 * n1;
    const int ie = (i + 1 == g.Nx) ? 0 : i + 1;
    double d = 0.0;
    if (ok) d = ((us[row + ie] - us[row + i]) * g.rdx + (vs[rown + i] - vs[row + i]) * g.rdy + (ws[rowt + i] - ws[row + i]) * rdz) * rdt;
    v[n1] = {d, 0.0};
  }
  dft16<1>(v);
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmul(v[k1], tw256[(r * k1) & 255]);
  transpose_halves(sm, v, [&](int k1) { return (k1 * 16 + r) * 16 + ln; }, [&](int n2) { return (r * 16 + n2) * 16 + ln; });
  dft16<1>(v);                                    // v[k2] = X[k1 + 16 k2], k1 = r
  if (ok) {
    cd* out = spec + L * 129;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) out[r + 16 * k2] = v[k2];
    if (r == 0) out[128] = v[8];
  }
}

// host side ---------------------------------------------------------------------------------------------------
struct ZSolve {
  cd* tw = nullptr;
  double* lxy = nullptr;
  long ncol = 0;
};

void* zsolve_create(ocn_ctx* ctx, const std::vector<double>& lx_half, const std::vector<double>& ly_local) {
  // lxy[kx + Nxh * ky]
  ZSolve* z = new ZSolve;
  const size_t Nxh = lx_half.size(), Ny = ly_local.size();
  z->ncol = (long)(Nxh * Ny);
  std::vector<double> lxy(Nxh * Ny);
  for (size_t j = 0; j < Ny; ++j)
    for (size_t i = 0; i < Nxh; ++i) lxy[i + Nxh * j] = lx_half[i] + ly_local[j];
  std::vector<cd> tw(256);
  for (int m = 0; m < 256; ++m) tw[m] = {cos(2.0 * M_PI * m / 256.0), -sin(2.0 * M_PI * m / 256.0)};
  if (hipMalloc((void**)&z->tw, sizeof(cd) * 256) != hipSuccess ||
      hipMalloc((void**)&z->lxy, sizeof(double) * lxy.size()) != hipSuccess) {
    ocn_set_error(ctx, "zsolve: allocation failed");
    delete z;
    return nullptr;
  }
  hipMemcpy(z->tw, tw.data(), sizeof(cd) * 256, hipMemcpyHostToDevice);
  hipMemcpy(z->lxy, lxy.data(), sizeof(double) * lxy.size(), hipMemcpyHostToDevice);
  return z;
}

void zsolve_destroy(void* p) {
  ZSolve* z = (ZSolve*)p;
  if (!z) return;
  hipFree(z->tw);
  hipFree(z->lxy);
  delete z;
}

// forward / inverse y pass (in place), and the fused rhs + x pass; all unnormalised
void yfft256_run(ocn_ctx* ctx, void* p, void* spec, int Nxh, int Nz, int inverse) {
  ZSolve* z = (ZSolve*)p;
  const int nfull = Nxh / 16, left = Nxh - 16 * nfull;
  const long nblk = (long)nfull * Nz + ((long)left * Nz + 15) / 16;
  dim3 b(256, 1, 1), g((unsigned)nblk, 1, 1);
  if (inverse) ocn_launch_sync(k_yfft256<-1>, g, b, ctx->stream, (cd*)spec, Nxh, Nz, (const cd*)z->tw);
  else ocn_launch_sync(k_yfft256<1>, g, b, ctx->stream, (cd*)spec, Nxh, Nz, (const cd*)z->tw);
}

void xfft_rhs256_run(ocn_model* m, void* p, void* spec, double dt) {
  ZSolve* z = (ZSolve*)p;
  const GridDev& g = m->gd;
  const long nlines = (long)g.Ny * g.Nz;
  dim3 b(256, 1, 1), gr((unsigned)((nlines + 15) / 16), 1, 1);
  ocn_launch_sync(k_xfft_rhs256, gr, b, m->ctx->stream, g, (const double*)m->us.interior(), (const double*)m->vs.interior(),
                  (const double*)m->ws.interior(), 1.0 / dt, m->g->dist ? 0 : 1, (cd*)spec, (const cd*)z->tw);
}

// in place on the (ncol, 256) spectrum; `zero_col` < 0 when this rank does not own the mean mode
void zsolve_run(ocn_ctx* ctx, void* p, void* spec, const double* lz, double norm, long zero_col) {
  ZSolve* z = (ZSolve*)p;
  dim3 b(256, 1, 1), g((unsigned)((z->ncol + 15) / 16), 1, 1);
  ocn_launch_sync(k_zsolve256, g, b, ctx->stream, (cd*)spec, z->ncol, (const double*)z->lxy, lz, (const cd*)z->tw, norm,
                  zero_col);
}
