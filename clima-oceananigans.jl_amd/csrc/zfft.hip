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

// in place on the (ncol, 256) spectrum; `zero_col` < 0 when this rank does not own the mean mode
void zsolve_run(ocn_ctx* ctx, void* p, void* spec, const double* lz, double norm, long zero_col) {
  ZSolve* z = (ZSolve*)p;
  dim3 b(256, 1, 1), g((unsigned)((z->ncol + 15) / 16), 1, 1);
  ocn_launch_sync(k_zsolve256, g, b, ctx->stream, (cd*)spec, z->ncol, (const double*)z->lxy, lz, (const cd*)z->tw, norm,
                  zero_col);
}
