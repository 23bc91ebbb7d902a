// TEST INFRASTRUCTURE — the device layer stubbed for the CPU sanitizer build (`make -C veloci_amd/csrc asan`, tests/test_host_asan.py): the HIP
// runtime calls the host side makes are mapped to host memory, every kernel launcher throws (one exception, opt-in: with VQ_STUB_DICT_SCAN=1
// exact / prefix dictionary probes are answered by a plain loop, see launch_dict_scan below).  What this build can run is everything in front of
// the first launch: index staging (index.cpp), request parsing, query compilation (compile.cpp), the C ABI's argument handling — under
// AddressSanitizer + UndefinedBehaviorSanitizer.  Never linked into the product library.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "../../veloci_amd/csrc/engine.hpp"

extern "C" {
hipError_t hipMalloc(void** p, size_t n) {
    *p = std::malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    std::memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    std::memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) {
    std::memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    std::memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) {
    *s = reinterpret_cast<hipStream_t>(std::malloc(8));
    return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned int, int) {
    *s = reinterpret_cast<hipStream_t>(std::malloc(8));
    return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) {
    *lo = 0;
    *hi = -1;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    std::free(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
    *e = reinterpret_cast<hipEvent_t>(std::malloc(8));
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned int) { return hipEventCreate(e); }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) {
    std::free(e);
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
    *ms = 0.0f;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stubbed HIP runtime"; }
hipError_t hipGetDeviceCount(int* n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
    *v = 256;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
void vq_stub_fail_launches_after(long k);
}

namespace vq {
// VQ_STUB_NOOP_LAUNCH=1 (tools/host_step_profile.py only): launches do nothing instead of throwing, so that the host side of a whole step
// (compile, pack, launch calls, result assembly — over garbage "results") can be timed on a machine without a GPU
// vq_stub_fail_launches_after(k) (tests/native/gloo_step_driver.py): the k+1-th launch from now and every later one fail — a device error in the
// middle of a step, on one rank only; k < 0 switches it off again
static std::atomic<long> g_fail_after{-1};
void stub_set_fail_after(long k) { g_fail_after.store(k); }
static void no_device(const char* what) {
    static const bool noop = std::getenv("VQ_STUB_NOOP_LAUNCH") != nullptr;
    if (g_fail_after.load() >= 0 && g_fail_after.fetch_sub(1) <= 0) {
        g_fail_after.store(0);
        throw vqreq::VelociError(vqreq::ERR_DEVICE, std::string("device layer stubbed: injected failure of ") + what);
    }
    if (noop) return;
    throw vqreq::VelociError(vqreq::ERR_DEVICE, std::string("device layer stubbed: ") + what);
}
size_t tile_scan_lds_bytes(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, bool, uint32_t) { return 0; }
size_t scan_simple_lds_bytes(uint32_t, uint32_t, uint32_t, bool) { return 0; }
size_t scan_wide_lds_bytes(uint32_t, uint32_t, uint32_t) { return 0; }
int debug_facet_select(const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t*) { return -1; }
uint32_t debug_div100_mismatches() {
    no_device("debug_div100_mismatches");
    return 0;
}
void launch_tile_scan(hipStream_t, uint32_t, size_t, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t, unsigned long long*,
                      unsigned long long*, uint32_t*, bool, uint32_t, bool) { no_device("k_tile_scan"); }
void launch_scan_leaf_f32(hipStream_t, uint32_t, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, unsigned long long*, unsigned long long*, uint32_t*) {
    no_device("k_scan_leaf_f32");
}
void launch_scan_simple(hipStream_t, bool, uint32_t, uint32_t, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, unsigned long long*, unsigned long long*,
                        uint32_t*, bool) { no_device("k_scan_simple"); }
void launch_scan_wide(hipStream_t, uint32_t, uint32_t, uint32_t, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, unsigned long long*,
                      unsigned long long*) { no_device("k_scan_wide"); }
void launch_scan_probe(hipStream_t, uint32_t, uint32_t, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, unsigned long long*, unsigned long long*, bool, bool) {
    no_device("k_scan_probe");
}
size_t scan_probe_lds_bytes(uint32_t, uint32_t) { return 0; }
void launch_scan_ring(hipStream_t, uint32_t, uint32_t, const uint8_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, const uint32_t*, uint32_t, uint32_t*, unsigned long long*,
                      unsigned long long*) {
    no_device("k_scan_ring");
}
uint32_t scan_ring_consumers(uint32_t) { return 10; }
uint32_t scan_ring_slots(uint32_t, uint32_t) { return 3; }
size_t scan_ring_lds_bytes(uint32_t, uint32_t, uint32_t) { return 0; }
void launch_merge_spans(hipStream_t, uint32_t, const uint8_t*, const uint32_t*, const unsigned long long*, unsigned long long*) { no_device("k_merge_spans"); }
void launch_finalize(hipStream_t, uint32_t, const uint8_t*, const uint32_t*, const uint8_t*, uint32_t, size_t, const PartialLayout&, uint32_t*, float*, uint32_t*, unsigned long long*) {
    no_device("k_finalize");
}
void launch_facet_select(hipStream_t, uint32_t, const FacetJob*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*) { no_device("k_facet_select"); }
void launch_range_hits(hipStream_t, uint32_t, uint32_t, const UList*, const RangeJobD*, const uint32_t*, unsigned long long*) { no_device("k_range_hits"); }
void launch_union(hipStream_t, bool, uint32_t, const UList*, const UTask*, const uint32_t*, uint32_t*, const uint64_t*, uint32_t*, float*, uint32_t*) { no_device("k_union"); }
void launch_scan_union(hipStream_t, bool, uint32_t, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, unsigned long long*, unsigned long long*) {
    no_device("k_scan_union");
}
// VQ_STUB_DICT_SCAN=1 (tools/host_profile.py only): exact / prefix probes answered by a plain loop, so that the host compiler can be timed on this
// machine on requests with prefix leaves.  The sanitizer test leaves it off: there every launcher throws.
void launch_dict_scan(hipStream_t, const DictProbe* probes, uint32_t probe_base, uint32_t n_probes, const uint32_t* off, const uint16_t* chars, const uint16_t* low_chars,
                      uint32_t num_terms, uint32_t* out_count, uint32_t out_cap, DictMatch* out) {
    if (!std::getenv("VQ_STUB_DICT_SCAN")) no_device("k_dict_scan");
    for (uint32_t p = 0; p < n_probes; ++p) {
        const DictProbe& P = probes[p];
        if (P.max_d != 0) no_device("k_dict_scan (the stub's loop answers distance 0 only)");
        for (uint32_t t = 0; t < num_terms; ++t) {
            const uint32_t n = off[t + 1] - off[t];
            if ((P.flags & 2u) ? n < P.m : n != P.m) continue;
            bool eq = true;
            for (uint32_t i = 0; i < P.m && eq; ++i) eq = chars[off[t] + i] == P.query[i];
            if (!eq) continue;
            uint32_t info = 0;
            if (P.lm != 0xFFFFFFFFu) {  // lower-cased hit against the lower-cased term: a prefix match is n - lm insertions away
                bool starts = n >= P.lm;
                for (uint32_t i = 0; i < P.lm && starts; ++i) starts = low_chars[off[t] + i] == P.lquery[i];
                const uint32_t d = starts ? std::min<uint32_t>(n - P.lm, 255u) : 255u;
                info = d | (d << 8) | (uint32_t(starts) << 16);
            }
            const uint32_t pos = (*out_count)++;
            if (pos < out_cap) out[pos] = DictMatch{probe_base + p, t, info};
        }
    }
}
void launch_loc_gather(hipStream_t, const LocRow*, uint32_t, const uint32_t*, uint32_t*) { no_device("k_loc_gather"); }
void launch_loc_expand(hipStream_t, bool, const LocJob*, uint32_t, const uint32_t*, uint32_t, uint32_t*, unsigned long long*) { no_device("k_loc_expand"); }
void launch_loc_compact(hipStream_t, const LocJob*, uint32_t, const unsigned long long*, uint32_t*, float*, uint32_t*) { no_device("k_loc_compact"); }
size_t seg_sort_u32(void*, size_t, const uint32_t*, uint32_t*, uint32_t, uint32_t, const uint32_t*, const uint32_t*, hipStream_t) {
    no_device("seg_sort_u32");
    return 0;
}
size_t seg_sort_u64(void*, size_t, const unsigned long long*, unsigned long long*, uint32_t, uint32_t, const uint32_t*, const uint32_t*, hipStream_t) {
    no_device("seg_sort_u64");
    return 0;
}
void launch_b1n_map(hipStream_t, const B1nJob*, uint32_t, const uint32_t*, uint32_t*, float*, B1nResult*) { no_device("k_b1n_map"); }
void launch_explain(hipStream_t, uint32_t, const ExQuery*, const uint32_t*, const uint32_t*, const ExOp*, const uint16_t*, const ExList*, const DColBoost*, uint32_t*) { no_device("k_explain"); }
}  // namespace vq

extern "C" void vq_stub_fail_launches_after(long k) { vq::stub_set_fail_after(k); }
