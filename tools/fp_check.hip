// Are the round-to-nearest f64 intrinsics of this toolchain correctly rounded on gfx950?  (device vs host, 1M samples each)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const double* a, const double* b, double* o_sqrt, double* o_div, double* o_mul, double* o_add, double* o_rcp, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  o_sqrt[i] = __dsqrt_rn(a[i]); o_div[i] = __ddiv_rn(a[i], b[i]); o_mul[i] = __dmul_rn(a[i], b[i]); o_add[i] = __dadd_rn(a[i], b[i]); o_rcp[i] = __ddiv_rn(1.0, a[i]);
}
int main() {
  const int n = 1 << 20; std::vector<double> a(n), b(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  for (int i = 0; i < n; i++) { a[i] = rnd() * ((i & 1) ? 1e-3 : 1.0) + 1e-9; b[i] = rnd() + 1e-6; }
  double *da, *db, *o[5]; hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); for (auto& p : o) hipMalloc(&p, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, o[0], o[1], o[2], o[3], o[4], n);
  std::vector<double> r(n); const char* nm[5] = { "sqrt", "div", "mul", "add", "rcp" };
  for (int j = 0; j < 5; j++) {
    hipMemcpy(r.data(), o[j], n * 8, hipMemcpyDeviceToHost); int bad = 0;
    for (int i = 0; i < n; i++) { volatile double h = j == 0 ? std::sqrt(a[i]) : j == 1 ? a[i] / b[i] : j == 2 ? a[i] * b[i] : j == 3 ? a[i] + b[i] : 1.0 / a[i]; if (h != r[i]) bad++; }
    std::printf("%s: %d of %d differ from the host\n", nm[j], bad, n);
  }
  return 0;
}
