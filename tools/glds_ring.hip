// LDS-DMA (global_load_lds_*) on gfx950: what the instruction does with its operands (part A), and what a loader-wave / consumer-wave
// ring per CU streams (part B) — the skeleton of k_scan_ring (veloci_amd/csrc/scan_ring.hip), measured before that kernel was written.
//
//   hipcc --offload-arch=gfx950 -O3 tools/glds_ring.hip -o tools/glds_ring && tools/glds_ring
//
// Part B: one workgroup per CU, NL loader waves + C consumer waves.  A consumer posts requests (a global pointer per tile) into its own
// ring of S slots in LDS and "processes" landed tiles (reads `work` words of the slot per lane, or verifies every word); a loader serves
// the request queues round-robin with KT 1-KiB DMA pieces per tile and publishes a tile behind a counted s_waitcnt vmcnt.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

// ------------------------------------------------------------------------------------------------ part A
__global__ void k_sem(const uint32_t* g, uint32_t* out, int mode) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) lds[i] = 0xDEAD0000u | i;
    __syncthreads();
    uint32_t keep;
    const uint32_t ldsb = 512u;  // LDS byte address of the destination
    const unsigned long long base = (unsigned long long)g;
    if (mode == 0) {  // all lanes, 16 B each, instruction offset 1024
        const uint32_t voff = lane * 16u;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(ldsb), "s"(base) : "memory");
    } else if (mode == 1) {  // lanes 0-31 only
        const uint32_t voff = lane * 16u;
        if (lane < 32) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(ldsb), "s"(base) : "memory");
    } else if (mode == 2) {  // 4 B per lane
        const uint32_t voff = lane * 4u;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3 offset:256\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(ldsb), "s"(base) : "memory");
    } else if (mode == 3) {  // per-lane source addresses (clamped: the upper lanes repeat lane 40's vector)
        const uint32_t voff = (lane < 40u ? lane : 40u) * 16u;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(ldsb), "s"(base) : "memory");
    } else if (mode == 4) {  // odd lanes only
        const uint32_t voff = lane * 16u;
        if (lane & 1) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(ldsb), "s"(base) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (uint32_t i = lane; i < 4096; i += 64) out[i] = lds[i];
}

// ------------------------------------------------------------------------------------------------ part B
constexpr int kCtlWords = 64;
constexpr int kReqWords = 4;
__host__ __device__ constexpr uint32_t mix32(uint32_t i) { return i * 2654435761u + 12345u; }

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <int N>
__device__ __forceinline__ void vmcnt_imm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int KT, int M>
__device__ __forceinline__ void wait_all_but_tiles(int m) {  // all DMAs but those of the m (<= M) youngest tiles have landed
    static_assert(KT * M <= 63, "vmcnt is a 6-bit field");
    if (M >= 7 && m >= 7) vmcnt_imm<KT * (M >= 7 ? 7 : 0)>();
    else if (M >= 6 && m == 6) vmcnt_imm<KT * (M >= 6 ? 6 : 0)>();
    else if (M >= 5 && m == 5) vmcnt_imm<KT * (M >= 5 ? 5 : 0)>();
    else if (M >= 4 && m == 4) vmcnt_imm<KT * (M >= 4 ? 4 : 0)>();
    else if (M >= 3 && m == 3) vmcnt_imm<KT * (M >= 3 ? 3 : 0)>();
    else if (M >= 2 && m == 2) vmcnt_imm<KT * (M >= 2 ? 2 : 0)>();
    else if (m == 1) vmcnt_imm<KT>();
    else vmcnt_imm<0>();
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;  // (a volatile access through a GENERIC pointer is a flat_load sc0 sc1 + s_waitcnt vmcnt(0): it would drain every DMA in flight)
__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) { return *(const volatile lds_u32*)p; }
__device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) { *(volatile lds_u32*)p = v; }
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }  // LDS byte address
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int NL, int C, int S, int KT, int M>
__global__ __launch_bounds__((NL + C) * 64) void k_ring(const uint8_t* __restrict__ buf, unsigned long long bytes_per_consumer, uint32_t tiles_per_consumer, int work,
                                                        unsigned long long* __restrict__ sink, uint32_t* __restrict__ err) {
    extern __shared__ __attribute__((aligned(1024))) uint32_t lds[];
    constexpr uint32_t kSlotWords = KT * 256;
    uint32_t* ctl = lds;                                // [0, C): req_count, [16, 16 + C): full_count, [32, 32 + C): done
    uint32_t* reqs = lds + kCtlWords;                            // [C][S][kReqWords]
    uint32_t* slots = lds + 1024;                                // [C][S][kSlotWords]   (1 KiB aligned: a DMA piece never straddles anything)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < kCtlWords) lds[threadIdx.x] = 0u;
    __syncthreads();
    constexpr uint32_t kSpin = 1u << 22;

    if (wave < NL) {  // ---------------------------------------------------------------- loader
        __builtin_amdgcn_s_setprio(3);
        uint32_t issued = 0;                    // lane c: tiles issued for consumer c
        // FIFO of issued, unpublished tiles: lane i of `fifo` holds entry (head + i) % 8 as consumer << 24 | its count after the tile
        uint32_t fifo = 0, head = 0, npend = 0;
        uint32_t rr = 0, idle = 0;
        const uint32_t mine = (lane < C && (lane % NL) == wave) ? 1u : 0u;
        const uint32_t voff = lane * 16u;
        while (true) {
            const uint32_t rc = lane < C ? lds_ld(ctl + lane) : 0u;
            const uint32_t dn = lane < C ? lds_ld(ctl + 32 + lane) : 1u;
            lds_fence();
            const unsigned long long want = __builtin_amdgcn_ballot_w64(mine && rc != issued);
            if (want) {
                idle = 0;
                // round-robin: the first wanting consumer at or behind rr
                const unsigned long long hi = want & ~((1ull << rr) - 1ull);
                const uint32_t c = (uint32_t)__builtin_ctzll(hi ? hi : want);
                rr = (c + 1u) % C;
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)issued, (int)c);
                const uint32_t s = n % S;
                const uint32_t* rq = reqs + (c * S + s) * kReqWords;
                const uint32_t r0 = lds_ld(rq), r1 = lds_ld(rq + 1);
                lds_fence();
                const uint32_t plo = (uint32_t)__builtin_amdgcn_readfirstlane((int)r0), phi = (uint32_t)__builtin_amdgcn_readfirstlane((int)r1);
                const unsigned long long base = ((unsigned long long)phi << 32) | plo;
                const uint32_t ldsb = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(slots + (c * S + s) * kSlotWords));
                uint32_t keep;
#pragma unroll
                for (int g4 = 0; g4 < KT / 4; ++g4) {
                    const uint32_t lb = ldsb + g4 * 4096u;
                    const unsigned long long gb = base + g4 * 4096ull;
                    asm volatile(
                        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                        "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                        "global_load_lds_dwordx4 %1, %3 offset:2048\n\tglobal_load_lds_dwordx4 %1, %3 offset:3072\n\ts_mov_b32 m0, %0"
                        : "=&s"(keep)
                        : "v"(voff), "s"(lb), "s"(gb)
                        : "memory");
                }
                if (lane == c) issued = n + 1u;
                // publish the oldest pending tile once 3 are behind it
                if (npend == (uint32_t)M) {
                    wait_all_but_tiles<KT, M>(M);
                    const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)fifo, (int)(head & 7u));
                    if (lane == 0) lds_st(ctl + 16 + (e >> 24), e & 0xFFFFFFu);
                    ++head;
                    --npend;
                }
                if (lane == ((head + npend) & 7u)) fifo = (c << 24) | (n + 1u);
                ++npend;
            } else if (npend) {  // nothing to issue: publish what is in flight, oldest first
                wait_all_but_tiles<KT, M>((int)npend - 1);
                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)fifo, (int)(head & 7u));
                if (lane == 0) lds_st(ctl + 16 + (e >> 24), e & 0xFFFFFFu);
                ++head;
                --npend;
            } else {
                if (!__builtin_amdgcn_ballot_w64(mine && !dn)) break;  // every consumer of mine is done and has nothing outstanding
                __builtin_amdgcn_s_sleep(1);
                if (++idle > kSpin) {
                    if (lane == 0) atomicAdd(err, 1u);
                    break;
                }
            }
        }
    } else {  // ---------------------------------------------------------------------- consumer
        const uint32_t c = wave - NL;
        const uint8_t* region = buf + (unsigned long long)(blockIdx.x * C + c) * bytes_per_consumer;
        uint32_t nreq = 0;
        unsigned long long acc = 0;
        auto post = [&]() {
            const unsigned long long p = (unsigned long long)(uintptr_t)(region + (unsigned long long)nreq * (KT * 1024ull));
            uint32_t* rq = reqs + (c * S + (nreq % S)) * kReqWords;
            if (lane == 0) {
                lds_st(rq, (uint32_t)p);
                lds_st(rq + 1, (uint32_t)(p >> 32));
            }
            lds_fence();
            ++nreq;
            if (lane == 0) lds_st(ctl + c, nreq);
        };
        for (uint32_t i = 0; i < (uint32_t)S && nreq < tiles_per_consumer; ++i) post();
        bool bad = false;
        for (uint32_t n = 0; n < tiles_per_consumer && !bad; ++n) {
            uint32_t spin = 0;
            while (lds_ld(ctl + 16 + c) <= n) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > kSpin) {
                    bad = true;
                    break;
                }
            }
            if (bad) break;
            lds_fence();
            const uint32_t* sl = slots + (c * S + (n % S)) * kSlotWords;
            if (work < 0) {  // verify every word
                const uint32_t w0 = (uint32_t)(((unsigned long long)(blockIdx.x * C + c) * bytes_per_consumer + (unsigned long long)n * (KT * 1024ull)) / 4ull);
                for (uint32_t i = lane; i < kSlotWords; i += 64)
                    if (sl[i] != mix32(w0 + i)) acc += 1ull;
            } else {
                for (int i = 0; i < work; ++i) acc += sl[(lane * 17u + (uint32_t)i * 131u) % kSlotWords];
            }
            lds_fence();
            if (nreq < tiles_per_consumer) post();
        }
        if (bad && lane == 0) atomicAdd(err, 1u);
        if (lane == 0) lds_st(ctl + 32 + c, 1u);
        if (acc) atomicAdd(sink, acc);
    }
}

// the same bytes read the plain way: one wave per workgroup, 16 B per lane, 4 loads in flight
__global__ __launch_bounds__(64) void k_plain(const uint4* __restrict__ buf, unsigned long long vec_per_wave, unsigned long long* __restrict__ sink) {
    const uint4* p = buf + (unsigned long long)blockIdx.x * vec_per_wave;
    uint32_t acc = 0;
    for (unsigned long long i = threadIdx.x; i + 192 < vec_per_wave; i += 256) {
        const uint4 a = p[i], b = p[i + 64], c = p[i + 128], d = p[i + 192];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345u) atomicAdd(sink, 1ull);
}

__global__ void k_fill(uint32_t* w, unsigned long long n) {
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) w[i] = mix32((uint32_t)i);
}

template <int NL, int C, int S, int KT, int M>
static void run_ring(const uint8_t* buf, unsigned long long total_bytes, int cus, int work, unsigned long long* sink, uint32_t* err, const char* label) {
    const unsigned long long per = (total_bytes / ((unsigned long long)cus * C)) & ~((unsigned long long)KT * 1024 - 1);
    const uint32_t tiles = (uint32_t)(per / (KT * 1024ull));
    const size_t lds = 4096 + (size_t)C * S * KT * 1024;
    if (lds > 160 * 1024) {
        std::printf("%-44s skipped: %zu B of LDS\n", label, lds);
        return;
    }
    CK(hipFuncSetAttribute((const void*)k_ring<NL, C, S, KT, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipMemset(sink, 0, 8));
    CK(hipMemset(err, 0, 4));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_ring<NL, C, S, KT, M>), dim3(cus), dim3((NL + C) * 64), lds, 0, buf, per, tiles, work, sink, err);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep) best = ms < best ? ms : best;
    }
    unsigned long long hs;
    uint32_t he;
    CK(hipMemcpy(&hs, sink, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
    const double bytes = (double)tiles * KT * 1024.0 * cus * C;
    std::printf("%-44s %8.3f ms  %7.1f GB/s  (%.2f GB, lds %zu, err %u%s)\n", label, best, bytes / best * 1e-6, bytes * 1e-9, lds, he,
                work < 0 ? (hs ? ", MISMATCHES" : ", verified") : "");
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    std::printf("device %s, %d CUs\n", prop.name, cus);

    // ---- part A
    {
        std::vector<uint32_t> hg(8192);
        for (size_t i = 0; i < hg.size(); ++i) hg[i] = 0x10000u + (uint32_t)i;
        uint32_t *g, *out;
        CK(hipMalloc(&g, hg.size() * 4));
        CK(hipMalloc(&out, 4096 * 4));
        CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 5; ++mode) {
            hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 4096 * 4, 0, g, out, mode);
            CK(hipDeviceSynchronize());
            std::vector<uint32_t> ho(4096);
            CK(hipMemcpy(ho.data(), out, 4096 * 4, hipMemcpyDeviceToHost));
            int first = -1, last = -1, count = 0;
            for (int i = 0; i < 4096; ++i)
                if ((ho[i] >> 16) != 0xDEADu) {
                    if (first < 0) first = i;
                    last = i;
                    ++count;
                }
            std::printf("mode %d: %d LDS words written, first word %d (byte %d) = g[%u], last word %d = g[%u]", mode, count, first, first * 4, first >= 0 ? ho[first] - 0x10000u : 0, last,
                        last >= 0 ? ho[last] - 0x10000u : 0);
            if (mode == 3 && first >= 0) std::printf("; word of lane 41 = g[%u], lane 63 = g[%u]", ho[first + 41 * 4] - 0x10000u, ho[first + 63 * 4] - 0x10000u);
            if ((mode == 1 || mode == 4) && first >= 0) {
                std::printf("; words at +0,+4,+8,+128: g[%u] g[%u] g[%u] g[%u]", ho[first] - 0x10000u, ho[first + 4] - 0x10000u, ho[first + 8] - 0x10000u, ho[first + 128] - 0x10000u);
            }
            std::printf("\n");
        }
        CK(hipFree(g));
        CK(hipFree(out));
    }

    // ---- part B
    const unsigned long long total = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 24ull) << 30;
    uint8_t* buf;
    CK(hipMalloc(&buf, total));
    hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, 0, (uint32_t*)buf, total / 4);
    CK(hipDeviceSynchronize());
    unsigned long long* sink;
    uint32_t* err;
    CK(hipMalloc(&sink, 8));
    CK(hipMalloc(&err, 4));
    {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        for (int waves_per_cu : {8, 12, 16, 32}) {
            const int blocks = cus * waves_per_cu * 4;
            const unsigned long long vec_per_wave = total / 16 / blocks;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(64), 0, 0, (const uint4*)buf, vec_per_wave, sink);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                if (rep) best = ms < best ? ms : best;
            }
            std::printf("plain 16 B/lane loads, %d blocks                %8.3f ms  %7.1f GB/s\n", blocks, best, (double)total / best * 1e-6);
        }
    }
    run_ring<1, 4, 3, 12, 3>(buf, total, cus, -1, sink, err, "ring NL1 C4 S3 12K M3 verify");
    run_ring<2, 4, 3, 12, 5>(buf, total, cus, -1, sink, err, "ring NL2 C4 S3 12K M5 verify");
    run_ring<1, 4, 3, 4, 7>(buf, total, cus, -1, sink, err, "ring NL1 C4 S3 4K M7 verify");
    for (int work : {0, 32, 128}) {
        char l[96];
#define RUN(NL, C, S, KT, M)                                                                   \
    std::snprintf(l, sizeof l, "ring NL%d C%d S%d %dK M%d work %d", NL, C, S, KT, M, work);    \
    run_ring<NL, C, S, KT, M>(buf, total, cus, work, sink, err, l);
        RUN(1, 4, 3, 12, 1)
        RUN(1, 4, 3, 12, 2)
        RUN(1, 4, 3, 12, 3)
        RUN(1, 4, 3, 12, 5)
        RUN(2, 4, 3, 12, 2)
        RUN(2, 4, 3, 12, 3)
        RUN(2, 4, 3, 12, 5)
        RUN(1, 4, 8, 4, 3)
        RUN(1, 4, 8, 4, 7)
        RUN(2, 4, 8, 4, 7)
        RUN(3, 3, 4, 12, 5)
        RUN(4, 4, 3, 12, 5)
        RUN(2, 6, 2, 12, 5)
        RUN(2, 5, 2, 12, 5)
    }
    return 0;
}
