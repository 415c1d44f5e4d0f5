// scatter_roofline — what does an MI355X deliver for SCATTERED small accesses?  The tick kernels of this library read and write
// 4..64 bytes per lane at addresses that share no cache line (~4.4 G receivers/s whatever the arrangement, 1-15 % of
// the streaming 8 TB/s).  Before the next kernel is rewritten the ceiling for that access pattern should be a measured number:
// this tool sweeps working set (L2 / Infinity Cache / HBM resident), bytes per access (4 / 16 / 64, the last as four lanes
// per line), lanes in flight (waves per SIMD by LDS padding) and dependence (independent loads vs a pointer chase), for
// loads, stores and returning atomics, and prints useful GB/s, accesses per second and time per dependent access.
//   hipcc --offload-arch=gfx950 -O3 -o scatter_roofline tools/scatter_roofline.hip && ./scatter_roofline [--quick]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// every lane: `n` independent accesses of 4 or 16 bytes at hashed slots of a table of `slots` 16-byte entries
template <int BYTES>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ tab, uint32_t mask, uint32_t n, uint32_t salt, uint32_t* out) {
  extern __shared__ uint32_t pad[];
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0, h = mix(gid ^ salt);
  for (uint32_t i = 0; i < n; i++) {
    h = mix(h + 0x9e3779b9u);
    if (BYTES == 4) acc += ((const uint32_t*)tab)[(size_t)(h & mask) * 4];
    else { const uint4 v = tab[h & mask]; acc += v.x ^ v.w; }
  }
  if (acc == 0x12345u) out[0] = acc + pad[0];
}
// 64 bytes per access, the way a kernel would have to fetch a node's line: four adjacent lanes take 16 bytes each of one line
__global__ void __launch_bounds__(256) k_gather_quad(const uint4* __restrict__ tab, uint32_t mask, uint32_t n, uint32_t salt, uint32_t* out) {
  extern __shared__ uint32_t pad[];
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0, h = mix((gid >> 2) ^ salt);
  for (uint32_t i = 0; i < n; i++) {
    h = mix(h + 0x9e3779b9u);
    const uint4 v = tab[(size_t)((h & mask) & ~3u) + (gid & 3u)];
    acc += v.x ^ v.w;
  }
  if (acc == 0x12345u) out[0] = acc + pad[0];
}
// a pointer chase: every access depends on the one before (the shape of a lane's merge: line -> header -> queue -> view ...)
__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ tab, uint32_t mask, uint32_t n, uint32_t salt, uint32_t* out) {
  extern __shared__ uint32_t pad[];
  uint32_t at = mix((blockIdx.x * 256 + threadIdx.x) ^ salt) & mask, acc = 0;
  for (uint32_t i = 0; i < n; i++) { const uint4 v = tab[at]; acc += v.y; at = (v.x ^ acc) & mask; }
  if (acc == 0x12345u) out[0] = acc + pad[0];
}
__global__ void __launch_bounds__(256) k_scatter_store(uint4* tab, uint32_t mask, uint32_t n, uint32_t salt) {
  extern __shared__ uint32_t pad[];
  uint32_t h = mix((blockIdx.x * 256 + threadIdx.x) ^ salt);
  for (uint32_t i = 0; i < n; i++) { h = mix(h + 0x9e3779b9u); tab[h & mask] = make_uint4(h, i, salt, pad[0]); }
}
__global__ void __launch_bounds__(256) k_scatter_atomic(uint32_t* tab, uint32_t mask, uint32_t n, uint32_t salt, uint32_t* out) {
  extern __shared__ uint32_t pad[];
  uint32_t h = mix((blockIdx.x * 256 + threadIdx.x) ^ salt), acc = 0;
  for (uint32_t i = 0; i < n; i++) { h = mix(h + 0x9e3779b9u); acc += atomicAdd(&tab[(size_t)(h & mask) * 4], 1u); }   // returning: the inbox reservation
  if (acc == 0x12345u) out[0] = acc + pad[0];
}
__global__ void k_fill(uint4* tab, size_t slots) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) tab[i] = make_uint4(mix((uint32_t)i), (uint32_t)i, 0, 0);
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !std::strcmp(argv[1], "--quick");
  int dev = 0, cus = 0; CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  std::printf("device %s, %d CUs\n", prop.gcnArchName, cus);
  const size_t max_bytes = quick ? (size_t)1 << 30 : (size_t)8 << 30;
  uint4* tab; uint32_t* out; CK(hipMalloc(&tab, max_bytes)); CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, tab, max_bytes / 16); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // more than 64 KB of dynamic LDS per workgroup has to be asked for (it is only there to limit occupancy)
  const int lds_max = 160 * 1024;
  (void)hipFuncSetAttribute((const void*)k_gather<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  (void)hipFuncSetAttribute((const void*)k_gather<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  (void)hipFuncSetAttribute((const void*)k_gather_quad, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  (void)hipFuncSetAttribute((const void*)k_chase, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  (void)hipFuncSetAttribute((const void*)k_scatter_store, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  (void)hipFuncSetAttribute((const void*)k_scatter_atomic, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  const size_t sets_mb[] = { 16, 128, 1024, 8192 };            // L2-resident (8 x 4 MB), Infinity-Cache-resident (256 MB), HBM, HBM
  const int waves_per_simd[] = { 1, 2, 4, 8 };
  std::printf("%-14s %8s %6s %10s %12s %12s %10s\n", "kernel", "set MB", "w/SIMD", "lanes", "useful GB/s", "G access/s", "ns/dep.acc");
  for (size_t smb : sets_mb) {
    if (smb * ((size_t)1 << 20) > max_bytes) continue;
    const uint32_t mask = (uint32_t)(smb * ((size_t)1 << 20) / 16 - 1);
    for (int w : waves_per_simd) {
      // occupancy by LDS: a CU has 160 KB; a workgroup of 4 waves that takes 160/w KB leaves room for w workgroups = w waves per SIMD
      const size_t lds = w >= 8 ? 0 : (size_t)(160 * 1024 / w) - 1024;
      const uint32_t blocks = (uint32_t)cus * (uint32_t)w * 4u;   // four rounds of resident workgroups
      const uint32_t n = quick ? 64 : 256;
      for (int kind = 0; kind < 6; kind++) {
        if (kind == 5 && smb > 1024) continue;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
          CK(hipEventRecord(e0, 0));
          switch (kind) {
            case 0: hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), lds, 0, tab, mask, n, (uint32_t)rep, out); break;
            case 1: hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(256), lds, 0, tab, mask, n, (uint32_t)rep, out); break;
            case 2: hipLaunchKernelGGL(k_gather_quad, dim3(blocks), dim3(256), lds, 0, tab, mask, n, (uint32_t)rep, out); break;
            case 3: hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(256), lds, 0, tab, mask, n, (uint32_t)rep, out); break;
            case 4: hipLaunchKernelGGL(k_scatter_store, dim3(blocks), dim3(256), lds, 0, tab, mask, n, (uint32_t)rep); break;
            default: hipLaunchKernelGGL(k_scatter_atomic, dim3(blocks), dim3(256), lds, 0, (uint32_t*)tab, mask, n, (uint32_t)rep, out); break;
          }
          { hipError_t le = hipGetLastError(); if (le != hipSuccess) { std::printf("launch failed (%zu MB, %d w/SIMD, lds %zu): %s\n", smb, w, lds, hipGetErrorString(le)); best = -1; break; } }
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        if (best < 0) continue;
        static const char* const names[6] = { "gather 4 B", "gather 16 B", "gather 64 B/4", "chase 16 B", "store 16 B", "atomic 4 B" };
        static const int bytes[6] = { 4, 16, 16, 16, 16, 4 };
        const double acc = (double)blocks * 256 * n, sec = best * 1e-3;
        // a resident lane's dependent-access time: lanes in flight / accesses per second (meaningful for the chase; a bound for the rest)
        const double lanes = (double)cus * w * 4 * 64;
        std::printf("%-14s %8zu %6d %10.0f %12.1f %12.2f %10.0f\n", names[kind], smb, w, lanes, acc * bytes[kind] / sec / 1e9, acc / sec / 1e9, lanes / (acc / sec) * 1e9);
      }
    }
  }
  return 0;
}
