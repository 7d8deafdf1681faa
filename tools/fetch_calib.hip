// FETCH_SIZE / WRITE_SIZE calibration for THIS repository's access patterns (MI355X_MICROARCH.md, HBM: "calibrate on a
// known byte count in your own access pattern before trusting an absolute"). Known-size kernels over a 2 GiB array
// (past the 256 MiB Infinity Cache); run each under rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) and divide:
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o _exp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- _exp/fetch_calib
//   calib_stream16      16 B per lane, coalesced (the guide's case: counter = 1/2 of the bytes)
//   calib_stream4       4 B per lane, coalesced (K6's per-Gaussian scalars)
//   calib_gather64x48   random 64-byte records, 48 B of each read as 3 x 16 B by one lane (compositing kernels: SplatRec)
//   calib_gather64x64   random 64-byte records read whole (4 x 16 B)
//   calib_gather12      random 12-byte rows (means3D-style gathers)
//   calib_scatter8      random 8-byte writes (gsr_scatter's keys)        -> WRITE_SIZE
//   calib_stream_w16    16 B per lane coalesced writes                   -> WRITE_SIZE
// The program prints the exact byte count of each kernel (unique cache lines touched for the gathers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

extern "C" __global__ void calib_stream16(const float4* __restrict__ a, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 1.2345e-30f) out[0] = s;
}
extern "C" __global__ void calib_stream4(const float* __restrict__ a, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 1.2345e-30f) out[0] = s;
}
// nrec records of 64 B; gather g (< ngather) reads record perm(g): a bijection on [0, nrec) when ngather <= nrec, so
// every record is touched at most once and the byte count is exact
__device__ __forceinline__ uint32_t perm(uint32_t g, uint32_t mask) { return (hash32(g) ^ (g * 0x9e3779b1u)) & mask; }
extern "C" __global__ void calib_gather64(const float4* __restrict__ a, uint32_t mask, uint32_t ngather, int parts, float* out) {
    float s = 0.f;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngather; g += gridDim.x * blockDim.x) {
        const float4* p = a + (size_t)(hash32(g * 2654435761u + 12345u) & mask) * 4;
        for (int q = 0; q < parts; ++q) { const float4 v = p[q]; s += v.x + v.w; }
    }
    if (s == 1.2345e-30f) out[0] = s;
}
extern "C" __global__ void calib_gather12(const float* __restrict__ a, uint32_t mask, uint32_t ngather, float* out) {
    float s = 0.f;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngather; g += gridDim.x * blockDim.x) {
        const float* p = a + (size_t)(hash32(g * 2654435761u + 777u) & mask) * 3;
        s += p[0] + p[1] + p[2];
    }
    if (s == 1.2345e-30f) out[0] = s;
}
extern "C" __global__ void calib_scatter8(unsigned long long* __restrict__ a, uint32_t mask, uint32_t nwrite) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nwrite; g += gridDim.x * blockDim.x)
        a[hash32(g * 2654435761u + 99u) & mask] = g;
}
extern "C" __global__ void calib_stream_w16(float4* __restrict__ a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    void* buf = nullptr; float* out = nullptr;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc((void**)&out, 256));
    CHECK(hipMemset(buf, 0x3c, bytes));
    const dim3 grid(256 * 16), block(256);
    const uint32_t rec_mask = (uint32_t)(bytes / 64) - 1u;           // 32 Mi records of 64 B
    const uint32_t ngather = 4u << 20;                                 // 4 Mi random records: collisions ~6 % (counted below on the host)
    // exact unique-line counts of the hashed gathers (host replay)
    auto uniq = [&](uint32_t mult, uint32_t add, uint32_t mask, uint32_t n, uint32_t line_of_shift, uint32_t bytes_per_item) {
        (void)bytes_per_item;
        uint8_t* seen = (uint8_t*)calloc(((size_t)mask + 1) >> line_of_shift, 1);
        size_t u = 0;
        for (uint32_t g = 0; g < n; ++g) {
            uint32_t x = g * mult + add; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
            const size_t line = (size_t)(x & mask) >> line_of_shift;
            if (!seen[line]) { seen[line] = 1; ++u; }
        }
        free(seen);
        return u;
    };
    for (int rep = 0; rep < 2; ++rep) {                                // second round = the measured one (clocks, TLBs warm)
        hipLaunchKernelGGL(calib_stream16, grid, block, 0, 0, (const float4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(calib_stream4, grid, block, 0, 0, (const float*)buf, bytes / 4, out);
        hipLaunchKernelGGL(calib_gather64, grid, block, 0, 0, (const float4*)buf, rec_mask, ngather, 3, out);
        hipLaunchKernelGGL(calib_gather64, grid, block, 0, 0, (const float4*)buf, rec_mask, ngather, 4, out);
        hipLaunchKernelGGL(calib_gather12, grid, block, 0, 0, (const float*)buf, (uint32_t)(bytes / 16) - 1u, ngather, out);
        hipLaunchKernelGGL(calib_scatter8, grid, block, 0, 0, (unsigned long long*)buf, (uint32_t)(bytes / 8) - 1u, ngather);
        hipLaunchKernelGGL(calib_stream_w16, grid, block, 0, 0, (float4*)buf, bytes / 16);
        CHECK(hipDeviceSynchronize());
    }
    const size_t u64 = uniq(2654435761u, 12345u, rec_mask, ngather, 0, 64);
    const size_t u8 = uniq(2654435761u, 99u, (uint32_t)(bytes / 8) - 1u, ngather, 0, 8);
    printf("calib_stream16   reads  %zu bytes\n", bytes);
    printf("calib_stream4    reads  %zu bytes\n", bytes);
    printf("calib_gather64   x48:   %zu unique 64-B records = %zu bytes touched as lines, %zu bytes requested (launch 1 of each pair: parts=3, launch 2: parts=4 -> %zu)\n",
           u64, u64 * 64, (size_t)ngather * 48, (size_t)ngather * 64);
    printf("calib_gather12   reads  %zu bytes requested (rows of 12 B, %u gathers, a 64-B line each unless shared)\n", (size_t)ngather * 12, ngather);
    printf("calib_scatter8   writes %zu bytes requested, %zu unique 8-B slots\n", (size_t)ngather * 8, u8);
    printf("calib_stream_w16 writes %zu bytes\n", bytes);
    return 0;
}
