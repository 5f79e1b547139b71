// micro_checks.hip -- hardware facts the tendency kernel relies on, checked on the GPU box (not part of the product):
//   (1) accuracy of v_rcp_f64 followed by 0 / 1 / 2 Newton steps
//   (2) lane direction of v_mov_b32_dpp wave_shl:1 and of ds_bpermute with address (lane + 1) % 64
//   (3) global_load_lds_dwordx4 lands lane l's 16 bytes at base + 16 l, honouring EXEC
// build: hipcc -O3 --offload-arch=gfx950 tools/micro_checks.hip -o tools/micro_checks
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__global__ void k_rcp(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  r1[i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  r2[i] = r;
}

__global__ void k_shift(const double* in, double* dpp, double* bperm) {
  const int tid = threadIdx.x;
  union { double d; int i[2]; } a, b, c;
  a.d = in[tid];
  c.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x130, 0xf, 0xf, false);
  c.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x130, 0xf, 0xf, false);
  const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int addr = ((lane + 1) & 63) << 2;
  b.i[0] = __builtin_amdgcn_ds_bpermute(addr, a.i[0]);
  b.i[1] = __builtin_amdgcn_ds_bpermute(addr, a.i[1]);
  dpp[tid] = c.d;
  bperm[tid] = b.d;
}

__global__ void k_glds(const double* src, double* dst, int npieces) {
  __shared__ double buf[2048];
  const int tid = threadIdx.x;
  buf[tid] = -1.0; buf[tid + 256] = -1.0; buf[tid + 512] = -1.0; buf[tid + 768] = -1.0;
  __syncthreads();
  const int wave = tid >> 6;
  if (tid < npieces)   // partial last wave: EXEC-masked lanes must neither load nor write
    __builtin_amdgcn_global_load_lds((glb_void*)(src + 2 * tid), (lds_void*)(buf + wave * 128), 16, 0, 0);
  __syncthreads();
  for (int q = tid; q < 1024; q += 256) dst[q] = buf[q];
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> ex(-40.0, 40.0), mant(1.0, 2.0);
  for (int i = 0; i < n; ++i) x[i] = mant(g) * std::pow(2.0, std::floor(ex(g)));
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k_rcp<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  std::vector<double> r0(n), r1(n), r2(n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)x[i];
    e0 = std::fmax(e0, (double)fabsl(((long double)r0[i] - t) / t));
    e1 = std::fmax(e1, (double)fabsl(((long double)r1[i] - t) / t));
    e2 = std::fmax(e2, (double)fabsl(((long double)r2[i] - t) / t));
  }
  printf("rcp_f64 max relative error: raw %.3e, 1 Newton step %.3e, 2 Newton steps %.3e (eps = 1.1e-16)\n", e0, e1, e2);

  std::vector<double> in(256), a(256), b(256);
  for (int i = 0; i < 256; ++i) in[i] = i;
  double *din, *da, *db;
  hipMalloc(&din, 2048 * 8); hipMalloc(&da, 2048 * 8); hipMalloc(&db, 2048 * 8);
  hipMemcpy(din, in.data(), 256 * 8, hipMemcpyHostToDevice);
  k_shift<<<1, 256>>>(din, da, db);
  hipMemcpy(a.data(), da, 256 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), db, 256 * 8, hipMemcpyDeviceToHost);
  printf("dpp wave_shl:1  lanes 0,1,62,63,64: %g %g %g %g %g\n", a[0], a[1], a[62], a[63], a[64]);
  printf("bpermute(l+1)   lanes 0,1,62,63,64: %g %g %g %g %g\n", b[0], b[1], b[62], b[63], b[64]);
  int ok_dpp = 1, ok_bp = 1;
  for (int i = 0; i < 256; ++i) {
    if ((i & 63) != 63 && a[i] != i + 1) ok_dpp = 0;
    if (b[i] != (i & ~63) + ((i + 1) & 63)) ok_bp = 0;
  }
  printf("dpp wave_shl:1 gives lane+1 on lanes 0..62: %s; bpermute rotates within the wave: %s\n", ok_dpp ? "yes" : "NO", ok_bp ? "yes" : "NO");

  std::vector<double> s(2048), o(1024);
  for (int i = 0; i < 2048; ++i) s[i] = 1000 + i;
  hipMemcpy(din, s.data(), 2048 * 8, hipMemcpyHostToDevice);
  const int npieces = 200;   // 3 full waves + 8 lanes
  k_glds<<<1, 256>>>(din, da, npieces);
  hipMemcpy(o.data(), da, 1024 * 8, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int q = 0; q < 1024; ++q) {
    const double want = q < 2 * npieces ? 1000 + q : -1.0;
    if (o[q] != want) { ok = 0; printf("glds mismatch at %d: %g (want %g)\n", q, o[q], want); break; }
  }
  printf("global_load_lds_dwordx4 lane-linear + EXEC-masked tail: %s\n", ok ? "yes" : "NO");
  return 0;
}
