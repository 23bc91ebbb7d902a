// Calibration of rocprofv3's FETCH_SIZE on gfx950 against KNOWN byte / sector counts in the access patterns the scan kernels use
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//
//   k_stream16   every lane reads 16 B, coalesced: the bitmap / doc-id streams              -> bytes = N * 16
//   k_gather2    every lane reads ONE u16 from its own 64-byte sector (sectors are a random permutation of a buffer far
//                larger than the 256 MiB Infinity Cache, each touched exactly once): the per-survivor score gathers
//                                                                                             -> sectors = N, useful bytes = N * 2
//   k_gather2_pair  as k_gather2, but lanes 2i and 2i+1 read the two 64-byte halves of one 128-byte line (is a line fetched once?)
//
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv`; tools/fetch_calib_report.py prints, per kernel,
// FETCH_SIZE * 1024 / known bytes.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            std::fprintf(stderr, "HIP error %s at %s\n", hipGetErrorString(e), #x); \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_stream16(const u32x4* __restrict__ p, uint64_t n_vec, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * blockDim.x) {
        const u32x4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;  // never true for the fill pattern: keeps the loads alive
}

// sector of gather i: (i * odd) mod 2^k — a bijection on [0, 2^k)
__global__ __launch_bounds__(256) void k_gather2(const uint16_t* __restrict__ p, uint32_t log2_sectors, uint64_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    const uint64_t mask = (1ull << log2_sectors) - 1ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t sector = (i * 0x9E3779B97F4A7C15ull) & mask;
        acc += p[sector * 32ull + (i & 31ull)];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_gather2_pair(const uint16_t* __restrict__ p, uint32_t log2_lines, uint64_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    const uint64_t mask = (1ull << log2_lines) - 1ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t line = ((i >> 1) * 0x9E3779B97F4A7C15ull) & mask;  // 128-byte line shared by lanes 2j, 2j+1
        acc += p[line * 64ull + (i & 1ull) * 32ull + ((i >> 1) & 31ull)];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint32_t log2_sectors = 26;                          // 2^26 sectors * 64 B = 4 GiB: 16x the Infinity Cache
    const uint64_t bytes = (1ull << log2_sectors) * 64ull;
    void* buf = nullptr;
    uint32_t* out = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 256));
    CK(hipMemset(buf, 0x5A, bytes));
    CK(hipMemset(out, 0, 256));
    CK(hipDeviceSynchronize());
    const uint64_t n_vec = bytes / 16;
    const uint64_t n_g = 1ull << log2_sectors;  // every sector exactly once
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, uint64_t known_bytes, auto launch) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("%s rep %d: %.3f ms, known bytes %llu -> %.1f GB/s\n", name, rep, ms, (unsigned long long)known_bytes, known_bytes / (ms * 1e-3) / 1e9);
        }
    };
    timed("k_stream16 (4 GiB, 16 B/lane)", bytes, [&] { hipLaunchKernelGGL(k_stream16, dim3(8192), dim3(256), 0, 0, (const u32x4*)buf, n_vec, out); });
    timed("k_gather2 (2^26 gathers, one per 64-B sector)", n_g * 64ull, [&] { hipLaunchKernelGGL(k_gather2, dim3(8192), dim3(256), 0, 0, (const uint16_t*)buf, log2_sectors, n_g, out); });
    timed("k_gather2_pair (2^26 gathers, two per 128-B line)", n_g * 64ull, [&] { hipLaunchKernelGGL(k_gather2_pair, dim3(8192), dim3(256), 0, 0, (const uint16_t*)buf, log2_sectors - 1, n_g, out); });
    CK(hipDeviceSynchronize());
    std::printf("KNOWN k_stream16 %llu\nKNOWN k_gather2 %llu\nKNOWN k_gather2_pair %llu\n", (unsigned long long)bytes, (unsigned long long)(n_g * 64ull),
                (unsigned long long)(n_g * 64ull));
    return 0;
}
