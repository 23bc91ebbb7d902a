// Batch executor: compile n requests, pack their device programs, launch the kernels, assemble results.
// One batch == one k_tile_scan launch over all (query, span) pairs (SURVEY.md §7 "design for batches").
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "engine.hpp"

#include "text.hpp"

namespace vq {

using namespace vqreq;

void DevBuf::ensure(size_t n) {
    if (n <= bytes) return;
    size_t want = std::max(n, bytes + bytes / 2);
    alloc(want);
}
PinnedBuf::~PinnedBuf() {
    if (p) (void)hipHostFree(p);
}
void PinnedBuf::ensure(size_t n) {
    if (n <= bytes) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    size_t want = std::max(n, bytes + bytes / 2);
    VQ_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    bytes = want;
}

extern std::atomic<uint64_t> g_compile_ns[16];
static bool timing_enabled() {
    static const bool on = std::getenv("VQ_TIMING") != nullptr;
    return on;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

const char* const kKernelNames[K_COUNT_] = {"k_dict_scan", "k_union<count>", "k_union<write>", "k_range_hits", "k_tile_scan<count pre-pass>", "k_scan_leaf_f32",
                                            "k_scan_simple<2,rich>", "k_scan_ring (AND)", "k_scan_probe (AND / OR)", "k_scan_simple<2> (AND)", "k_scan_simple<2>", "k_scan_union", "k_scan_wide", "k_tile_scan",
                                            "k_merge_spans", "k_finalize", "k_facet_select", "k_locality", "k_boost1n"};

LaunchTimer::LaunchTimer(bool on, Workspace& w, hipStream_t s, int kernel, uint64_t layout_bytes, uint64_t algorithmic_bytes, uint64_t queries) {
    if (!on) return;
    if (w.ev_pool.empty()) {
        w.ev_pool.resize(96, nullptr);
        for (auto& e : w.ev_pool) VQ_HIP(hipEventCreate(&e));
    }
    if (w.ev_used + 2 > w.ev_pool.size()) return;  // (more launches than events: the rest of the batch goes untimed)
    ws = &w;
    st = s;
    slot = w.timed.size();
    w.timed.push_back(TimedLaunch{kernel, w.ev_used, w.ev_used + 1, layout_bytes, algorithmic_bytes, queries});
    w.ev_used += 2;
    VQ_HIP(hipEventRecord(w.ev_pool[w.timed[slot].ev_begin], st));
}
LaunchTimer::~LaunchTimer() {
    if (ws) (void)hipEventRecord(ws->ev_pool[ws->timed[slot].ev_end], st);
}

// host threads per index for request compilation (the caller counts as one): VQ_HOST_THREADS, else the machine's, at most 16 (a GPU's share of
// the host on an 8-GPU node)
static size_t host_threads() {
    static const size_t n = [] {
        const char* e = std::getenv("VQ_HOST_THREADS");
        size_t hw = std::max(1u, std::thread::hardware_concurrency());
        if (const char* lws = std::getenv("LOCAL_WORLD_SIZE"); lws && std::atoi(lws) > 1)  // one process per GPU (torch.distributed.run): this rank's share of the node
            hw = std::max<size_t>(hw / size_t(std::atoi(lws)), 2);
        size_t v = e ? size_t(std::atoi(e)) : std::min<size_t>(16, hw);
        return std::min<size_t>(std::max<size_t>(v, 1), 64);
    }();
    return n;
}
// [begin, end) parts of n items for `threads` workers, shrinking: each part is 1 / (2 x threads) of what is left, at least one item
// `singles`: that many leading items are parts of their own (items sorted heaviest first: the heavy ones must not queue up behind each other)
static std::vector<std::pair<size_t, size_t>> guided_ranges(size_t n, size_t threads, size_t singles = 0) {
    std::vector<std::pair<size_t, size_t>> out;
    if (threads <= 1) {
        if (n) out.push_back({0, n});
        return out;
    }
    singles = std::min(singles, n);
    for (size_t b = 0; b < singles; ++b) out.push_back({b, b + 1});
    if (singles) {
        for (auto& r : guided_ranges(n - singles, threads)) out.push_back({r.first + singles, r.second + singles});
        return out;
    }
    static const size_t fixed = std::getenv("VQ_COMPILE_PART") ? size_t(std::atoi(std::getenv("VQ_COMPILE_PART"))) : 0;  // (experiments: parts of a fixed size)
    for (size_t b = 0; b < n;) {
        const size_t len = fixed ? std::min(fixed, n - b) : std::max<size_t>((n - b) / (2 * threads), 1);
        out.push_back({b, b + len});
        b += len;
    }
    return out;
}
static HostPool& host_pool(const Index& idx) {
    std::lock_guard<std::mutex> g(idx.pool_mu);
    if (!idx.pool) idx.pool = std::make_unique<HostPool>(host_threads() - 1);
    return *idx.pool;
}

// serialise one compiled query into `dst` (host), whose device address will be `dev`
static size_t pack_blob(const CompiledQuery& cq, const Index& idx, uint8_t* dst, const uint8_t* dev, uint32_t keys_base, uint32_t part_keys_off,
                        const std::vector<uint32_t>& hist_off, const std::vector<uint32_t>& fac_out_off, size_t* desc_bytes_out = nullptr, uint32_t stat_off = 0) {
    size_t off = align_up(sizeof(QHeader), 16);
    QHeader h{};
    auto section = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 16);
        return o;
    };
    h.n_lists = uint32_t(cq.lists.size());
    h.n_ops = uint32_t(cq.ops.size());
    h.n_fops = uint32_t(cq.fops.size());
    h.n_groups = uint32_t(cq.groups.size());
    h.n_tboost = uint32_t(cq.tboosts.size());
    h.n_col = cq.n_top_cols;
    h.n_locf = uint32_t(cq.locf.size());
    h.n_facets = uint32_t(cq.facets.size());
    h.off_lists = uint32_t(section(cq.lists.size() * sizeof(DList)));
    h.off_ops = uint32_t(section(cq.ops.size() * sizeof(DOp)));
    h.off_fops = uint32_t(section(cq.fops.size() * sizeof(DOp)));
    h.off_groups = uint32_t(section(cq.groups.size() * sizeof(DGroup)));
    h.off_tboost = uint32_t(section(cq.tboosts.size() * sizeof(DTermBoost)));
    h.off_col = uint32_t(section(cq.cols.size() * sizeof(DColBoost)));
    h.off_locf = uint32_t(section(cq.locf.size() * sizeof(DLocField)));
    h.off_facets = uint32_t(section(cq.facets.size() * sizeof(DFacet)));
    h.n_pres = uint32_t(cq.pres.size());
    h.off_pres = uint32_t(section(cq.pres.size() * sizeof(DPresOp)));
    h.off_pres_in = uint32_t(section(cq.pres_in.size() * sizeof(uint16_t)));
    h.off_loc_idx = uint32_t(section(cq.loc_idx.size() * sizeof(uint16_t)));
    h.off_simple2 = uint32_t(section((cq.simple_flags >> 18) & 1u ? sizeof(DSimple2) : (cq.simple_flags >> 24) & 1u ? sizeof(DWide) : ((cq.simple_flags >> 25) & 1u) || ((cq.simple_flags >> 28) & 1u) ? sizeof(DProbe) : 0));
    const bool pool = ((cq.simple_flags >> 25) & 1u) && cq.top_k <= kPoolMaxK;
    h.off_pool = pool ? uint32_t(section(sizeof(DPool) + 8 * size_t(cq.top_k))) : 0u;
    h.n_temps = cq.n_temps;
    h.n_counts = cq.n_counts;
    h.prune_n = cq.prune_n;
    h.seq_tiles = cq.seq_tiles;
    h.key_upper = cq.key_upper;
    h.prune_mask = cq.prune_mask;
    std::memcpy(h.prune_gbits, cq.prune_gbits, sizeof h.prune_gbits);
    h.simple_n = cq.simple_n;
    h.bitmap_base = idx.bitmap_base;
    h.simple_flags = cq.simple_flags;
    h.desc_bytes = uint32_t(off);
    if (desc_bytes_out) *desc_bytes_out = off;
    std::vector<size_t> inline_off(cq.inline_lists.size());
    for (size_t i = 0; i < cq.inline_lists.size(); ++i) inline_off[i] = section(align_up(cq.inline_lists[i].size(), 4) * 4);
    std::vector<size_t> inline_val_off(cq.inline_vals.size());
    for (size_t i = 0; i < cq.inline_vals.size(); ++i) inline_val_off[i] = section(align_up(cq.inline_vals[i].size(), 4) * 4);
    h.top_k = cq.top_k;
    h.tile_words = cq.tile_words;
    h.n_spans = cq.n_spans;
    h.keys_base = keys_base;
    h.doc_lo = idx.doc_lo;
    h.doc_hi = idx.doc_hi;
    h.part_keys_off = part_keys_off;
    h.stat_off = stat_off;
    h.blob_bytes = uint32_t(off);
    if (!dst) return off;

    std::memcpy(dst, &h, sizeof h);
    if (h.off_pool) std::memset(dst + h.off_pool, 0, sizeof(DPool) + 8 * size_t(cq.top_k));
    DList* dl = reinterpret_cast<DList*>(dst + h.off_lists);
    for (size_t i = 0; i < cq.lists.size(); ++i) {
        const HList& l = cq.lists[i];
        DList d{};
        d.docs = l.inline_idx >= 0 ? reinterpret_cast<const uint32_t*>(dev + inline_off[l.inline_idx]) : l.d_docs;
        d.scores = l.inline_val_idx >= 0 ? reinterpret_cast<const uint16_t*>(dev + inline_val_off[l.inline_val_idx]) : l.d_scores;
        d.len = l.len;
        d.flags = l.flags;
        d.term_score = l.term_score;
        d.max_raw = l.max_raw;
        d.bitmap = l.d_bitmap;
        d.rank_dir = l.d_rank_dir;
        d.tile_dir = l.d_tile_dir;
        dl[i] = d;
    }
    if (!cq.ops.empty()) std::memcpy(dst + h.off_ops, cq.ops.data(), cq.ops.size() * sizeof(DOp));
    if (!cq.fops.empty()) std::memcpy(dst + h.off_fops, cq.fops.data(), cq.fops.size() * sizeof(DOp));
    if (!cq.groups.empty()) std::memcpy(dst + h.off_groups, cq.groups.data(), cq.groups.size() * sizeof(DGroup));
    if (!cq.tboosts.empty()) std::memcpy(dst + h.off_tboost, cq.tboosts.data(), cq.tboosts.size() * sizeof(DTermBoost));
    if (!cq.cols.empty()) std::memcpy(dst + h.off_col, cq.cols.data(), cq.cols.size() * sizeof(DColBoost));
    if (!cq.locf.empty()) std::memcpy(dst + h.off_locf, cq.locf.data(), cq.locf.size() * sizeof(DLocField));
    if (!cq.loc_idx.empty()) std::memcpy(dst + h.off_loc_idx, cq.loc_idx.data(), cq.loc_idx.size() * sizeof(uint16_t));
    if ((cq.simple_flags >> 18) & 1u) std::memcpy(dst + h.off_simple2, &cq.simple2, sizeof(DSimple2));
    if ((cq.simple_flags >> 24) & 1u) std::memcpy(dst + h.off_simple2, &cq.wide, sizeof(DWide));
    if (((cq.simple_flags >> 25) & 1u) || ((cq.simple_flags >> 28) & 1u)) std::memcpy(dst + h.off_simple2, &cq.probe, sizeof(DProbe));
    if (!cq.pres.empty()) std::memcpy(dst + h.off_pres, cq.pres.data(), cq.pres.size() * sizeof(DPresOp));
    if (!cq.pres_in.empty()) std::memcpy(dst + h.off_pres_in, cq.pres_in.data(), cq.pres_in.size() * sizeof(uint16_t));
    DFacet* df = reinterpret_cast<DFacet*>(dst + h.off_facets);
    for (size_t i = 0; i < cq.facets.size(); ++i) {
        DFacet f = cq.facets[i];
        f.hist_off = i < hist_off.size() ? hist_off[i] : 0u;  // (the count pre-pass packs queries without facet outputs)
        f.out_off = i < fac_out_off.size() ? fac_out_off[i] : 0u;
        df[i] = f;
    }
    for (size_t i = 0; i < cq.inline_vals.size(); ++i)
        if (!cq.inline_vals[i].empty()) std::memcpy(dst + inline_val_off[i], cq.inline_vals[i].data(), cq.inline_vals[i].size() * 4);
    for (size_t i = 0; i < cq.inline_lists.size(); ++i) {
        uint32_t* p = reinterpret_cast<uint32_t*>(dst + inline_off[i]);
        const auto& v = cq.inline_lists[i];
        std::memcpy(p, v.data(), v.size() * 4);
        for (size_t k = v.size(); k < align_up(v.size(), 4); ++k) p[k] = 0xFFFFFFFFu;
    }
    return off;
}

// Answer every dictionary scan of a batch with k_dict_scan launches (grid.y = probe), then bring the match
// sets back sorted ascending (== FST stream order, which is what the reference's callback order is).
void run_fuzzy_probes(const Index& idx, Workspace& ws, FuzzyTable& table, hipStream_t st) {
    std::vector<FuzzyProbe*> todo;
    for (auto& kv : table)
        if (kv.second.status == 0) todo.push_back(&kv.second);
    if (todo.empty()) return;
    auto image_of = [&](const FuzzyProbe& fp) {  // the dictionary image a probe scans
        const Dictionary& d = idx.dict.at(fp.path);
        return fp.ci ? d.d_low.as<uint16_t>() : d.d_raw.as<uint16_t>();
    };
    // probes of one image next to each other: a launch scans ONE image for a run of probes (blocks answer 16 probes per pass over their terms)
    std::stable_sort(todo.begin(), todo.end(), [&](const FuzzyProbe* a, const FuzzyProbe* b) { return image_of(*a) < image_of(*b); });
    std::vector<DictProbe> probes(todo.size());
    std::vector<uint8_t> host_scored(todo.size(), 0);
    for (size_t i = 0; i < todo.size(); ++i) {
        const FuzzyProbe& fp = *todo[i];
        DictProbe& P = probes[i];
        std::memset(&P, 0, sizeof P);
        P.m = uint32_t(fp.query.size());
        P.max_d = fp.max_d;
        P.flags = (fp.transposition ? 1u : 0u) | (fp.prefix ? 2u : 0u);
        for (size_t j = 0; j < fp.query.size(); ++j) P.query[j] = fp.query[j];
        const auto lcps = vqtext::decode_utf8(fp.lower_term);  // scoring side: the lower-cased term as a whole (search_field.rs:298-300)
        bool bmp = lcps.size() <= 64 && idx.dict.at(fp.path).low_exact;
        for (uint32_t cp : lcps) bmp = bmp && cp <= 0xFFFFu;
        if (bmp) {
            P.lm = uint32_t(lcps.size());
            for (size_t j = 0; j < lcps.size(); ++j) P.lquery[j] = uint16_t(lcps[j]);
            // a case-insensitive scan matches with the very string it scores with, over the very image: the kernel then scores a hit from the
            // tables it already holds in LDS instead of re-reading the probe from HBM character by character
            bool same = fp.ci && lcps.size() == fp.query.size();
            for (size_t j = 0; same && j < lcps.size(); ++j) same = lcps[j] == fp.query[j];
            if (same) P.flags |= 4u;
        } else {
            P.lm = 0xFFFFFFFFu;
            host_scored[i] = 1;
        }
    }
    DevBuf &d_probes = ws.d_probe_desc, &d_count = ws.d_probe_counts, &d_out = ws.d_probe_ids;
    d_probes.ensure(probes.size() * sizeof(DictProbe));
    d_count.ensure(64);
    VQ_HIP(hipMemcpyAsync(d_probes.p, probes.data(), probes.size() * sizeof(DictProbe), hipMemcpyHostToDevice, st));
    uint32_t cap = uint32_t(std::max<size_t>(64 * todo.size(), 1u << 16));  // matches of the whole batch share one output array
    std::vector<DictMatch> recs;
    for (int pass = 0; pass < 2; ++pass) {  // pass 1 only when the matches outgrew the first guess (the count is exact then)
        d_out.ensure(size_t(cap) * sizeof(DictMatch) + 16);
        VQ_HIP(hipMemsetAsync(d_count.p, 0, 4, st));
        {
            uint64_t dict_bytes = 0, layout = 0;  // every probe reads its dictionary once (offsets + code points) ...
            for (FuzzyProbe* fp : todo) {
                const Dictionary& d = idx.dict.at(fp->path);
                dict_bytes += d.d_off.bytes + d.d_low.bytes;
            }
            LaunchTimer timer(idx.profile.enabled, ws, st, K_DICT_SCAN, 0, dict_bytes, todo.size());
            for (size_t g0 = 0; g0 < todo.size();) {  // one launch per run of probes over the same image
                size_t g1 = g0 + 1;
                while (g1 < todo.size() && image_of(*todo[g1]) == image_of(*todo[g0])) ++g1;
                const Dictionary& d = idx.dict.at(todo[g0]->path);
                launch_dict_scan(st, d_probes.as<DictProbe>() + g0, uint32_t(g0), uint32_t(g1 - g0), d.d_off.as<uint32_t>(), image_of(*todo[g0]), d.d_low.as<uint16_t>(),
                                 uint32_t(d.terms.size()), d_count.as<uint32_t>(), cap, d_out.as<DictMatch>());
                layout += (d.d_off.bytes + d.d_low.bytes) * ((g1 - g0 + 15) / 16);  // ... this layout: once per 16 probes of one image
                g0 = g1;
            }
            if (!ws.timed.empty() && idx.profile.enabled) ws.timed.back().layout_bytes = layout;
        }
        VQ_HIP(hipGetLastError());
        uint32_t count = 0;
        VQ_HIP(hipMemcpyAsync(&count, d_count.p, 4, hipMemcpyDeviceToHost, st));
        VQ_HIP(hipStreamSynchronize(st));
        if (count > cap) {
            if (pass == 1) throw VelociError(ERR_DEVICE, "dictionary scan: match count changed between passes");
            cap = count;
            continue;
        }
        if (std::getenv("VQ_TIMING")) {
            std::fprintf(stderr, "[vq timing] dictionary scan: %zu probes, %u matches\n", todo.size(), count);
        }
        recs.resize(count);
        if (count) {
            VQ_HIP(hipMemcpyAsync(recs.data(), d_out.p, size_t(count) * sizeof(DictMatch), hipMemcpyDeviceToHost, st));
            VQ_HIP(hipStreamSynchronize(st));
        }
        break;
    }
    if (std::getenv("VQ_TIMING")) {  // the distribution of matches over the probes (a few short probes can hold most of them)
        std::vector<uint32_t> per(todo.size(), 0);
        for (auto& r : recs) per[r.probe]++;
        std::vector<uint32_t> sorted = per;
        std::sort(sorted.begin(), sorted.end(), std::greater<uint32_t>());
        std::string top;
        for (size_t i = 0; i < sorted.size() && i < 8; ++i) top += " " + std::to_string(sorted[i]);
        std::fprintf(stderr, "[vq timing] dictionary scan: most matches per probe:%s\n", top.c_str());
    }
    // bucket by probe, ascending term ids (== FST stream order, which is the reference's callback order)
    std::sort(recs.begin(), recs.end(), [](const DictMatch& a, const DictMatch& b) { return a.probe != b.probe ? a.probe < b.probe : a.term < b.term; });
    size_t r = 0;
    for (size_t i = 0; i < todo.size(); ++i) {
        FuzzyProbe& fp = *todo[i];
        fp.matches.clear();
        fp.scores.clear();
        const Dictionary& dict = idx.dict.at(fp.path);
        for (; r < recs.size() && recs[r].probe == i; ++r) {
            fp.matches.push_back(recs[r].term);
            if (host_scored[i]) continue;
            const uint32_t osa = recs[r].info & 0xFFu, lev = (recs[r].info >> 8) & 0xFFu;
            const bool starts = (recs[r].info >> 16) & 1u;
            uint32_t d;
            if (osa <= fp.lev && osa < 255u) d = osa;  // the scoring automaton's answer (search_field.rs:691-702)
            else {  // its fallback, plain Levenshtein in u8 — 255 for strings of 255 bytes or more (:705-732)
                const bool long_strings = fp.lower_term.size() >= 255 || (dict.terms[recs[r].term].size() >= 200 && vqtext::to_lower_utf8(dict.terms[recs[r].term]).size() >= 255);
                d = long_strings ? 255u : lev;
                if (!long_strings && (osa == 255u || lev == 255u)) {  // capped on the device: exact on the host (never for dictionary-sized terms)
                    host_scored[i] = 2;
                    break;
                }
            }
            fp.scores.push_back(default_score_for_distance_host(uint8_t(d), fp.check_prefix && starts));
        }
        if (host_scored[i] == 2) {  // (rare) finish the bucket, then score all of it on the host
            fp.matches.clear();
            size_t r0 = r;
            while (r0 > 0 && recs[r0 - 1].probe == i) --r0;
            for (r = r0; r < recs.size() && recs[r].probe == i; ++r) fp.matches.push_back(recs[r].term);
        }
        if (host_scored[i]) score_fuzzy_probe(idx, fp);
    }
}

// K2: run the union jobs of a batch.  Level 1 merges groups of <= 64 posting lists (one lane per list); a job with
// more lists gets a level-2 task over the level-1 outputs.  Each level: count pass -> host prefix sums -> write pass.
namespace {
struct UnionTaskH {
    std::vector<UList> lists;
    UnionJob* job = nullptr;   // level 1: set when this task IS the job's result (<= 64 lists); level 2: always
    size_t parent = SIZE_MAX;  // level 1: index of the level-2 task that consumes this output
    uint64_t out_off = 0;
    uint32_t len = 0;
    float max_value = std::numeric_limits<float>::infinity();
};

void run_union_level(bool timed, Workspace& ws, std::vector<UnionTaskH>& tasks, DevBuf& docs, DevBuf& vals, DevBuf& maxes, DevBuf& meta, hipStream_t st) {
    if (tasks.empty()) return;
    std::vector<UList> ulists;
    std::vector<UTask> utasks;
    std::vector<uint32_t> span_task;
    for (size_t t = 0; t < tasks.size(); ++t) {
        UTask u{};
        u.list_begin = uint32_t(ulists.size());
        u.n_lists = uint32_t(tasks[t].lists.size());
        uint64_t total = 0;
        uint32_t piv = 0;
        for (uint32_t i = 0; i < u.n_lists; ++i) {
            total += tasks[t].lists[i].len;
            if (tasks[t].lists[i].len > tasks[t].lists[piv].len) piv = i;
            ulists.push_back(tasks[t].lists[i]);
        }
        u.pivot = u.list_begin + piv;
        static const uint64_t span_len = std::getenv("VQ_UNION_SPAN") ? std::max(1, std::atoi(std::getenv("VQ_UNION_SPAN"))) : 128;  // (bench_jmdict shape: 512 -> 36.6 k, 256 -> 43.8 k, 128 -> 44.2 k, 64 -> 45.3 k requests/s)
        uint64_t spans = std::min<uint64_t>(std::max<uint64_t>(total / span_len, 1), 4096);  // (short spans: a span is one serial merge loop, its length is the pass's latency)
        spans = std::min<uint64_t>(spans, std::max<uint32_t>(tasks[t].lists[piv].len, 1u));
        u.span_begin = uint32_t(span_task.size());
        u.n_spans = uint32_t(spans);
        for (uint32_t k = 0; k < u.n_spans; ++k) span_task.push_back(uint32_t(t));
        utasks.push_back(u);
    }
    const size_t n_spans = span_task.size();
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_lists = 0, o_tasks = al(ulists.size() * sizeof(UList)), o_st = o_tasks + al(utasks.size() * sizeof(UTask)),
                 o_cnt = o_st + al(n_spans * 4), o_off = o_cnt + al(n_spans * 4), bytes = o_off + al(n_spans * 8);
    meta.ensure(bytes);
    uint8_t* m = meta.as<uint8_t>();
    VQ_HIP(hipMemcpyAsync(m + o_lists, ulists.data(), ulists.size() * sizeof(UList), hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(m + o_tasks, utasks.data(), utasks.size() * sizeof(UTask), hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(m + o_st, span_task.data(), n_spans * 4, hipMemcpyHostToDevice, st));
    uint64_t in_bytes = 0;
    for (auto& u : ulists) in_bytes += uint64_t(u.len) * ((u.flags & 1u) ? 8u : 6u);
    uint64_t out_bytes = 0;
    auto launch = [&](bool write) {
        LaunchTimer timer(timed, ws, st, write ? K_UNION_WRITE : K_UNION_COUNT, in_bytes + (write ? out_bytes : 0), in_bytes + (write ? out_bytes : 0), tasks.size());
        launch_union(st, write, uint32_t(n_spans), reinterpret_cast<const UList*>(m + o_lists), reinterpret_cast<const UTask*>(m + o_tasks),
                     reinterpret_cast<const uint32_t*>(m + o_st), reinterpret_cast<uint32_t*>(m + o_cnt), reinterpret_cast<const uint64_t*>(m + o_off),
                     docs.as<uint32_t>(), vals.as<float>(), maxes.as<uint32_t>());
        VQ_HIP(hipGetLastError());
    };
    launch(false);
    std::vector<uint32_t> cnt(n_spans);
    VQ_HIP(hipMemcpyAsync(cnt.data(), m + o_cnt, n_spans * 4, hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    std::vector<uint64_t> off(n_spans);
    uint64_t cursor = 0;
    for (size_t t = 0; t < tasks.size(); ++t) {
        tasks[t].out_off = cursor;
        uint64_t len = 0;
        for (uint32_t k = 0; k < utasks[t].n_spans; ++k) {
            off[utasks[t].span_begin + k] = cursor + len;
            len += cnt[utasks[t].span_begin + k];
        }
        if (len > 0xFFFFFFF0ull) throw VelociError(ERR_UNSUPPORTED, "materialised leaf longer than 2^32 entries");
        tasks[t].len = uint32_t(len);
        cursor += (len + 8 + 3) / 4 * 4;  // 8 sentinel entries behind every list, starts stay 16-byte aligned
    }
    out_bytes = cursor * 8;
    docs.ensure(cursor * 4 + 64);
    vals.ensure(cursor * 4 + 64);
    maxes.ensure(tasks.size() * 4 + 64);
    VQ_HIP(hipMemsetAsync(maxes.p, 0xFF, tasks.size() * 4, st));  // min(~order(value)) per task, written by the write pass
    VQ_HIP(hipMemcpyAsync(m + o_off, off.data(), n_spans * 8, hipMemcpyHostToDevice, st));
    launch(true);
    // the largest value of every merged list: the compiler's score bounds need it (top-k pruning of the scan)
    std::vector<uint32_t> mins(tasks.size());
    VQ_HIP(hipMemcpyAsync(mins.data(), maxes.p, tasks.size() * 4, hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    for (size_t t = 0; t < tasks.size(); ++t)
        if (mins[t] != 0xFFFFFFFFu) {
            const uint32_t bits = unorder_f32(~mins[t]);
            std::memcpy(&tasks[t].max_value, &bits, 4);
        } else tasks[t].max_value = 0.0f;
}
}  // namespace

// Range jobs (k_range_hits): the leaf's postings at and between the entry anchors of its 1:n boost list; summed over the shards.
void run_range_jobs(const Index& idx, Workspace& ws, RangeTable& table, const UnionTable& unions, hipStream_t st) {
    std::vector<UList> ulists;
    std::vector<RangeJobD> jobs;
    size_t n_anchors = 0;
    uint32_t n_blocks = 0;
    for (auto& kv : table) n_anchors += kv.second.anchors ? kv.second.anchors->size() : 0;
    if (n_anchors > 0x7FFFFFF0ull) throw VelociError(ERR_UNSUPPORTED, "1:n field boosts of one batch with more than 2^31 boosted anchors");
    std::vector<uint32_t> anchors;
    anchors.reserve(n_anchors);
    for (auto& kv : table) {
        RangeJob& job = kv.second;
        const PostingStore& ps = idx.postings.at(job.store_path);
        const uint32_t lb = uint32_t(ulists.size());
        auto uit = job.union_key.empty() ? unions.end() : unions.find(job.union_key);
        if (uit != unions.end()) {  // the leaf was materialised: its merged list holds exactly the leaf's hits
            if (uit->second.len) {
                UList u{};
                u.docs = uit->second.d_docs;
                u.len = uit->second.len;
                ulists.push_back(u);
            }
        } else
        for (uint32_t tid : job.tokens) {
            if (tid >= ps.len.size() || !ps.len[tid]) continue;
            UList u{};
            u.docs = ps.docs.as<uint32_t>() + ps.start[tid];
            u.len = ps.len[tid];
            ulists.push_back(u);
        }
        RangeJobD d{};
        d.list_begin = lb;
        d.n_lists = uint32_t(ulists.size()) - lb;
        d.anchor_begin = uint32_t(anchors.size());
        d.n_anchors = job.anchors ? uint32_t(job.anchors->size()) : 0u;
        d.block_begin = n_blocks;
        if (job.anchors) anchors.insert(anchors.end(), job.anchors->begin(), job.anchors->end());
        if (d.n_lists == 0 || d.n_anchors == 0) continue;  // nothing to count: the job's counts stay 0
        n_blocks += d.n_lists == 1 ? (d.n_anchors + 63u) / 64u : d.n_anchors;
        jobs.push_back(d);
    }
    std::vector<uint64_t> counts(2 * n_anchors, 0);
    if (!jobs.empty()) {
        auto al = [](size_t x) { return (x + 255) / 256 * 256; };
        const size_t o_jobs = al(ulists.size() * sizeof(UList)), o_anchors = o_jobs + al(jobs.size() * sizeof(RangeJobD)), o_cnt = o_anchors + al(anchors.size() * 4),
                     bytes = o_cnt + al(counts.size() * 8);
        ws.d_union_meta.ensure(bytes);
        uint8_t* m = ws.d_union_meta.as<uint8_t>();
        VQ_HIP(hipMemcpyAsync(m, ulists.data(), ulists.size() * sizeof(UList), hipMemcpyHostToDevice, st));
        VQ_HIP(hipMemcpyAsync(m + o_jobs, jobs.data(), jobs.size() * sizeof(RangeJobD), hipMemcpyHostToDevice, st));
        VQ_HIP(hipMemcpyAsync(m + o_anchors, anchors.data(), anchors.size() * 4, hipMemcpyHostToDevice, st));
        VQ_HIP(hipMemsetAsync(m + o_cnt, 0, counts.size() * 8, st));
        {
            LaunchTimer timer(idx.profile.enabled, ws, st, K_RANGE_HITS, 0, 0, jobs.size());
            launch_range_hits(st, n_blocks, uint32_t(jobs.size()), reinterpret_cast<const UList*>(m), reinterpret_cast<const RangeJobD*>(m + o_jobs),
                              reinterpret_cast<const uint32_t*>(m + o_anchors), reinterpret_cast<unsigned long long*>(m + o_cnt));
        }
        VQ_HIP(hipGetLastError());
        VQ_HIP(hipMemcpyAsync(counts.data(), m + o_cnt, counts.size() * 8, hipMemcpyDeviceToHost, st));
        VQ_HIP(hipStreamSynchronize(st));
    }
    if (idx.sharded()) idx.sum_over_shards(counts);
    size_t k = 0;
    for (auto& kv : table) {
        const size_t n = kv.second.anchors ? 2 * kv.second.anchors->size() : 0;
        kv.second.counts.assign(counts.begin() + k, counts.begin() + k + n);
        k += n;
    }
}

void run_union_jobs(const Index& idx, Workspace& ws, UnionTable& table, hipStream_t st) {
    std::vector<UnionTaskH> l1, l2;
    for (auto& kv : table) {
        UnionJob& job = kv.second;
        std::vector<UList> raw;
        for (auto& t : job.terms) {
            const PostingStore& ps = *t.store;
            UList u{};
            u.docs = ps.docs.as<uint32_t>() + ps.start[t.token];
            u.scores = ps.scores.as<uint16_t>() + ps.start[t.token];
            u.len = ps.len[t.token];
            u.term_score = t.score;
            raw.push_back(u);
        }
        if (raw.size() > 64 * 64) throw VelociError(ERR_UNSUPPORTED, "leaf expansion with more than 4096 posting lists");
        if (raw.size() <= 64) {
            UnionTaskH t;
            t.lists = std::move(raw);
            t.job = &job;
            l1.push_back(std::move(t));
        } else {
            UnionTaskH top;
            top.job = &job;
            for (size_t b = 0; b < raw.size(); b += 64) {
                UnionTaskH t;
                t.lists.assign(raw.begin() + b, raw.begin() + std::min(raw.size(), b + 64));
                t.parent = l2.size();
                l1.push_back(std::move(t));
            }
            l2.push_back(std::move(top));
        }
    }
    run_union_level(idx.profile.enabled, ws, l1, ws.d_union_docs[0], ws.d_union_vals[0], ws.d_union_max, ws.d_union_meta, st);
    for (auto& t : l1) {
        const uint32_t* d = ws.d_union_docs[0].as<uint32_t>() + t.out_off;
        const float* v = ws.d_union_vals[0].as<float>() + t.out_off;
        if (t.job) {
            t.job->d_docs = d;
            t.job->d_vals = v;
            t.job->max_value = t.max_value;
            t.job->len = t.len;
        } else {
            UList u{};
            u.docs = d;
            u.scores = v;
            u.len = t.len;
            u.term_score = 1.0f;
            u.flags = 1u;
            l2[t.parent].lists.push_back(u);
        }
    }
    if (!l2.empty()) {
        run_union_level(idx.profile.enabled, ws, l2, ws.d_union_docs[1], ws.d_union_vals[1], ws.d_union_max, ws.d_union_meta, st);  // (level 1 has synchronised)
        for (auto& t : l2) {
            t.job->max_value = t.max_value;
            t.job->d_docs = ws.d_union_docs[1].as<uint32_t>() + t.out_off;
            t.job->d_vals = ws.d_union_vals[1].as<float>() + t.out_off;
            t.job->len = t.len;
        }
    }
}

// K7: text locality of fields whose text ids are not anchors, for every (request, field) job of the batch (see kernels.hip, k_loc_*).
void run_locality_jobs(const Index& idx, Workspace& ws, LocalityTable& table, hipStream_t st) {
    std::vector<LocalityJob*> jobs;
    for (auto& kv : table) jobs.push_back(&kv.second);
    if (jobs.empty()) return;
    // all jobs of one table next to each other: one gather launch per tokens_to_text_id table
    std::stable_sort(jobs.begin(), jobs.end(), [](const LocalityJob* a, const LocalityJob* b) { return a->t2t_path < b->t2t_path; });
    const size_t nj = jobs.size();
    std::vector<LocJob> dj(nj);
    std::vector<LocRow> rows;
    std::vector<std::pair<size_t, size_t>> table_rows;  // per run of jobs over one table: [first row, end row)
    uint64_t cursor = 0;
    for (size_t j = 0; j < nj; ++j) {
        const KVStore& t2t = idx.kv.at(jobs[j]->t2t_path);
        const KVStore& t2a = idx.kv.at(jobs[j]->t2a_path);
        if (j == 0 || jobs[j]->t2t_path != jobs[j - 1]->t2t_path) table_rows.push_back({rows.size(), rows.size()});
        LocJob& J = dj[j];
        std::memset(&J, 0, sizeof J);
        J.t2a_vals = t2a.values.as<uint32_t>();
        J.t2a_start = t2a.d_row_start.as<uint64_t>();
        J.t2a_len = t2a.d_row_len.as<uint32_t>();
        J.t2a_key_base = t2a.key_base;
        J.t2a_num_keys = t2a.num_keys;
        J.seg_begin = uint32_t(cursor);
        for (uint32_t id : jobs[j]->tokens) {
            if (id < t2t.key_base || id - t2t.key_base >= t2t.num_keys) continue;
            const uint32_t r = id - t2t.key_base;
            uint64_t src = t2t.host_off[r], left = t2t.host_off[r + 1] - t2t.host_off[r];
            while (left) {  // long rows in pieces: one workgroup copies one piece
                const uint32_t piece = uint32_t(std::min<uint64_t>(left, 65536));
                rows.push_back(LocRow{src, cursor, piece, 0u});
                src += piece;
                cursor += piece;
                left -= piece;
            }
        }
        if (cursor > 0xFFFFFFF0ull) throw VelociError(ERR_UNSUPPORTED, "text_locality: more than 2^32 token->text entries in one batch");
        J.seg_end = uint32_t(cursor);
        table_rows.back().second = rows.size();
    }
    const uint32_t E = uint32_t(cursor);
    for (auto* j : jobs) {
        j->d_docs = nullptr;
        j->d_vals = nullptr;
        j->len = 0;
    }
    if (!E) return;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    // meta: [jobs][rows][seg_begin u32][seg_end u32][counters u32][pair_begin u32][pair_end u32][out_len u32]
    const size_t o_jobs = 0, o_rows = al(nj * sizeof(LocJob)), o_sb = o_rows + al(rows.size() * sizeof(LocRow)), o_se = o_sb + al(nj * 4), o_cnt = o_se + al(nj * 4),
                 o_pb = o_cnt + al(nj * 4), o_pe = o_pb + al(nj * 4), o_len = o_pe + al(nj * 4), meta_bytes = o_len + al(nj * 4);
    ws.d_loc_meta.ensure(meta_bytes);
    uint8_t* m = ws.d_loc_meta.as<uint8_t>();
    std::vector<uint32_t> sb(nj), se(nj);
    for (size_t j = 0; j < nj; ++j) {
        sb[j] = dj[j].seg_begin;
        se[j] = dj[j].seg_end;
    }
    ws.d_loc_a.ensure(size_t(E) * 4 + 16);
    ws.d_loc_b.ensure(size_t(E) * 4 + 16);
    LaunchTimer timer(idx.profile.enabled, ws, st, K_LOCALITY, size_t(E) * 16, size_t(E) * 4, nj);
    VQ_HIP(hipMemcpyAsync(m + o_jobs, dj.data(), nj * sizeof(LocJob), hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(m + o_rows, rows.data(), rows.size() * sizeof(LocRow), hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(m + o_sb, sb.data(), nj * 4, hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(m + o_se, se.data(), nj * 4, hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemsetAsync(m + o_cnt, 0, nj * 4, st));
    {
        size_t tr = 0;
        for (size_t j = 0; j < nj; ++j)
            if (j == 0 || jobs[j]->t2t_path != jobs[j - 1]->t2t_path) {
                const KVStore& t2t = idx.kv.at(jobs[j]->t2t_path);
                const auto [r0, r1] = table_rows[tr++];
                launch_loc_gather(st, reinterpret_cast<const LocRow*>(m + o_rows) + r0, uint32_t(r1 - r0), t2t.d_text_vals.as<uint32_t>(), ws.d_loc_a.as<uint32_t>());
            }
    }
    VQ_HIP(hipGetLastError());
    auto sort32 = [&](const uint32_t* in, uint32_t* out, uint32_t n, const uint32_t* b, const uint32_t* e) {
        const size_t need = seg_sort_u32(nullptr, 0, in, out, n, uint32_t(nj), b, e, st);
        if (need == size_t(-1)) throw VelociError(ERR_DEVICE, "segmented radix sort failed (size query)");
        ws.d_loc_tmp.ensure(need + 256);
        if (seg_sort_u32(ws.d_loc_tmp.p, need, in, out, n, uint32_t(nj), b, e, st) == size_t(-1)) throw VelociError(ERR_DEVICE, "segmented radix sort failed");
    };
    sort32(ws.d_loc_a.as<uint32_t>(), ws.d_loc_b.as<uint32_t>(), E, reinterpret_cast<const uint32_t*>(m + o_sb), reinterpret_cast<const uint32_t*>(m + o_se));
    // count pass: (anchor, boost) pairs per job
    launch_loc_expand(st, false, reinterpret_cast<const LocJob*>(m + o_jobs), uint32_t(nj), ws.d_loc_b.as<uint32_t>(), E, reinterpret_cast<uint32_t*>(m + o_cnt), nullptr);
    VQ_HIP(hipGetLastError());
    std::vector<uint32_t> totals(nj);
    VQ_HIP(hipMemcpyAsync(totals.data(), m + o_cnt, nj * 4, hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    uint64_t P = 0, out_cursor = 0;
    std::vector<uint32_t> pb(nj), pe(nj);
    for (size_t j = 0; j < nj; ++j) {
        dj[j].pair_begin = pb[j] = uint32_t(P);
        P += totals[j];
        if (P > 0xFFFFFFF0ull) throw VelociError(ERR_UNSUPPORTED, "text_locality: more than 2^32 (anchor, boost) pairs in one batch");
        dj[j].pair_end = pe[j] = uint32_t(P);
        dj[j].out_off = uint32_t(out_cursor);
        out_cursor += (uint64_t(totals[j]) + 8 + 3) / 4 * 4;  // 8 sentinel entries behind every list, starts stay 16-byte aligned
    }
    ws.d_loc_docs.ensure(out_cursor * 4 + 64);
    ws.d_loc_vals.ensure(out_cursor * 4 + 64);
    if (P) {
        ws.d_loc_pairs_a.ensure(P * 8 + 16);
        ws.d_loc_pairs_b.ensure(P * 8 + 16);
        VQ_HIP(hipMemcpyAsync(m + o_jobs, dj.data(), nj * sizeof(LocJob), hipMemcpyHostToDevice, st));
        VQ_HIP(hipMemcpyAsync(m + o_pb, pb.data(), nj * 4, hipMemcpyHostToDevice, st));
        VQ_HIP(hipMemcpyAsync(m + o_pe, pe.data(), nj * 4, hipMemcpyHostToDevice, st));
        VQ_HIP(hipMemsetAsync(m + o_cnt, 0, nj * 4, st));
        launch_loc_expand(st, true, reinterpret_cast<const LocJob*>(m + o_jobs), uint32_t(nj), ws.d_loc_b.as<uint32_t>(), E, reinterpret_cast<uint32_t*>(m + o_cnt),
                          ws.d_loc_pairs_a.as<unsigned long long>());
        VQ_HIP(hipGetLastError());
        const size_t need = seg_sort_u64(nullptr, 0, ws.d_loc_pairs_a.as<unsigned long long>(), ws.d_loc_pairs_b.as<unsigned long long>(), uint32_t(P), uint32_t(nj),
                                         reinterpret_cast<const uint32_t*>(m + o_pb), reinterpret_cast<const uint32_t*>(m + o_pe), st);
        if (need == size_t(-1)) throw VelociError(ERR_DEVICE, "segmented radix sort failed (size query)");
        ws.d_loc_tmp.ensure(need + 256);
        if (seg_sort_u64(ws.d_loc_tmp.p, need, ws.d_loc_pairs_a.as<unsigned long long>(), ws.d_loc_pairs_b.as<unsigned long long>(), uint32_t(P), uint32_t(nj),
                         reinterpret_cast<const uint32_t*>(m + o_pb), reinterpret_cast<const uint32_t*>(m + o_pe), st) == size_t(-1))
            throw VelociError(ERR_DEVICE, "segmented radix sort failed");
    } else VQ_HIP(hipMemcpyAsync(m + o_jobs, dj.data(), nj * sizeof(LocJob), hipMemcpyHostToDevice, st));
    launch_loc_compact(st, reinterpret_cast<const LocJob*>(m + o_jobs), uint32_t(nj), ws.d_loc_pairs_b.as<unsigned long long>(), ws.d_loc_docs.as<uint32_t>(),
                       ws.d_loc_vals.as<float>(), reinterpret_cast<uint32_t*>(m + o_len));
    VQ_HIP(hipGetLastError());
    std::vector<uint32_t> lens(nj);
    VQ_HIP(hipMemcpyAsync(lens.data(), m + o_len, nj * 4, hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    for (size_t j = 0; j < nj; ++j) {
        jobs[j]->d_docs = ws.d_loc_docs.as<uint32_t>() + dj[j].out_off;
        jobs[j]->d_vals = ws.d_loc_vals.as<float>() + dj[j].out_off;
        jobs[j]->len = lens[j];
    }
}

// K10: the 1:n boost lists of a batch (boost.rs:432-468).  Host: one gather descriptor per text id (its value_id_to_parent row).  Device: gather
// the value ids, sort them per job (value-id order is the order the reference applies the boosts in), k_b1n_map.
void run_boost1n_jobs(const Index& idx, Workspace& ws, Boost1nTable& table, hipStream_t st) {
    const double t_begin = now_ms();
    std::vector<Boost1nJob*> jobs;
    for (auto& kv : table) jobs.push_back(&kv.second);
    if (jobs.empty()) return;
    std::stable_sort(jobs.begin(), jobs.end(), [](const Boost1nJob* a, const Boost1nJob* b) { return a->to_parent_path < b->to_parent_path; });
    const size_t nj = jobs.size();
    std::vector<B1nJob> dj(nj);
    std::vector<LocRow> rows;
    std::vector<std::pair<size_t, size_t>> table_rows;  // per run of jobs over one value_id_to_parent table: [first row, end row)
    uint64_t cursor = 0, out_cursor = 0;
    for (size_t j = 0; j < nj; ++j) {
        const KVStore& to_parent = idx.kv.at(jobs[j]->to_parent_path);
        const KVStore& to_anchor = idx.kv.at(jobs[j]->to_anchor_path);
        const BoostColumn& col = idx.boost.at(jobs[j]->boost_path);
        if (j == 0 || jobs[j]->to_parent_path != jobs[j - 1]->to_parent_path) table_rows.push_back({rows.size(), rows.size()});
        B1nJob& J = dj[j];
        std::memset(&J, 0, sizeof J);
        J.boost_present = col.has_present ? col.present.as<uint32_t>() : nullptr;
        J.boost_values = col.values.as<float>();
        J.boost_key_base = col.key_base;
        J.boost_num_keys = col.num_keys;
        J.to_anchor_off = to_anchor.d_csr_off.as<uint64_t>();
        J.to_anchor_vals = to_anchor.d_text_vals.as<uint32_t>();
        J.to_anchor_key_base = to_anchor.key_base;
        J.to_anchor_num_keys = to_anchor.num_keys;
        J.doc_lo = idx.doc_lo;
        J.doc_hi = idx.doc_hi;
        J.seg_begin = uint32_t(cursor);
        for (uint32_t id : jobs[j]->text_ids) {
            if (id < to_parent.key_base || id - to_parent.key_base >= to_parent.num_keys) continue;
            const uint32_t r = id - to_parent.key_base;
            uint64_t src = to_parent.host_off[r], left = to_parent.host_off[r + 1] - to_parent.host_off[r];
            while (left) {
                const uint32_t piece = uint32_t(std::min<uint64_t>(left, 65536));
                rows.push_back(LocRow{src, cursor, piece, 0u});
                src += piece;
                cursor += piece;
                left -= piece;
            }
        }
        if (cursor > 0xFFFFFFF0ull) throw VelociError(ERR_UNSUPPORTED, "1:n field boost: more than 2^32 value ids in one batch");
        J.seg_end = uint32_t(cursor);
        J.out_off = uint32_t(out_cursor);
        out_cursor += (uint64_t(J.seg_end - J.seg_begin) + 8 + 3) / 4 * 4;  // 8 sentinel entries behind every list, starts stay 16-byte aligned
        if (out_cursor > 0xFFFFFFF0ull) throw VelociError(ERR_UNSUPPORTED, "1:n field boost: more than 2^32 value ids in one batch");
        table_rows.back().second = rows.size();
    }
    const uint32_t E = uint32_t(cursor);
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_jobs = 0, o_rows = al(nj * sizeof(B1nJob)), o_sb = o_rows + al(rows.size() * sizeof(LocRow)), o_se = o_sb + al(nj * 4), o_res = o_se + al(nj * 4),
                 meta_bytes = o_res + al(nj * sizeof(B1nResult));
    ws.d_b1n_meta.ensure(meta_bytes);
    ws.d_b1n_a.ensure(size_t(E) * 4 + 16);
    ws.d_b1n_b.ensure(size_t(E) * 4 + 16);
    ws.d_b1n_docs.ensure(out_cursor * 4 + 64);
    ws.d_b1n_vals.ensure(out_cursor * 4 + 64);
    uint8_t* m = ws.d_b1n_meta.as<uint8_t>();
    std::vector<uint32_t> sb(nj), se(nj);
    for (size_t j = 0; j < nj; ++j) {
        sb[j] = dj[j].seg_begin;
        se[j] = dj[j].seg_end;
    }
    const double t_rows = now_ms();
    LaunchTimer timer(idx.profile.enabled, ws, st, K_BOOST1N, size_t(E) * 24, size_t(E) * 12, nj);
    VQ_HIP(hipMemcpyAsync(m + o_jobs, dj.data(), nj * sizeof(B1nJob), hipMemcpyHostToDevice, st));
    if (!rows.empty()) VQ_HIP(hipMemcpyAsync(m + o_rows, rows.data(), rows.size() * sizeof(LocRow), hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(m + o_sb, sb.data(), nj * 4, hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(m + o_se, se.data(), nj * 4, hipMemcpyHostToDevice, st));
    if (E) {
        size_t tr = 0;
        for (size_t j = 0; j < nj; ++j)
            if (j == 0 || jobs[j]->to_parent_path != jobs[j - 1]->to_parent_path) {
                const KVStore& to_parent = idx.kv.at(jobs[j]->to_parent_path);
                const auto [r0, r1] = table_rows[tr++];
                launch_loc_gather(st, reinterpret_cast<const LocRow*>(m + o_rows) + r0, uint32_t(r1 - r0), to_parent.d_text_vals.as<uint32_t>(), ws.d_b1n_a.as<uint32_t>());
            }
        VQ_HIP(hipGetLastError());
        const size_t need = seg_sort_u32(nullptr, 0, ws.d_b1n_a.as<uint32_t>(), ws.d_b1n_b.as<uint32_t>(), E, uint32_t(nj), reinterpret_cast<const uint32_t*>(m + o_sb),
                                         reinterpret_cast<const uint32_t*>(m + o_se), st);
        if (need == size_t(-1)) throw VelociError(ERR_DEVICE, "segmented radix sort failed (size query)");
        ws.d_b1n_tmp.ensure(need + 256);
        if (seg_sort_u32(ws.d_b1n_tmp.p, need, ws.d_b1n_a.as<uint32_t>(), ws.d_b1n_b.as<uint32_t>(), E, uint32_t(nj), reinterpret_cast<const uint32_t*>(m + o_sb),
                         reinterpret_cast<const uint32_t*>(m + o_se), st) == size_t(-1))
            throw VelociError(ERR_DEVICE, "segmented radix sort failed");
    }
    double t_sorted = 0;
    if (timing_enabled()) {
        VQ_HIP(hipStreamSynchronize(st));
        t_sorted = now_ms();
    }
    launch_b1n_map(st, reinterpret_cast<const B1nJob*>(m + o_jobs), uint32_t(nj), ws.d_b1n_b.as<uint32_t>(), ws.d_b1n_docs.as<uint32_t>(), ws.d_b1n_vals.as<float>(),
                   reinterpret_cast<B1nResult*>(m + o_res));
    VQ_HIP(hipGetLastError());
    std::vector<B1nResult> res(nj);
    VQ_HIP(hipMemcpyAsync(res.data(), m + o_res, nj * sizeof(B1nResult), hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    for (size_t j = 0; j < nj; ++j) {
        jobs[j]->d_docs = ws.d_b1n_docs.as<uint32_t>() + dj[j].out_off;
        jobs[j]->d_vals = ws.d_b1n_vals.as<float>() + dj[j].out_off;
        jobs[j]->len = res[j].len;
        jobs[j]->total = res[j].total;
        jobs[j]->ascending = !(res[j].flags & 1u);
        jobs[j]->several = (res[j].flags & 2u) != 0;
        jobs[j]->done = true;
    }
    if (timing_enabled()) {
        uint32_t longest = 0, n_several = 0;
        for (size_t j = 0; j < nj; ++j) {
            longest = std::max(longest, dj[j].seg_end - dj[j].seg_begin);
            n_several += jobs[j]->several ? 1u : 0u;
        }
        std::fprintf(stderr, "[vq timing] 1:n boost lists (K10): %zu jobs, %u value ids (longest list %u, %u with several values per anchor), %zu gather rows; host rows %.3f ms, gather + sort %.3f ms, map %.3f ms\n",
                     nj, E, longest, n_several, rows.size(), t_rows - t_begin, t_sorted - t_rows, now_ms() - t_sorted);
    }
}

// Count pre-pass: launches k_tile_scan in count mode for queries whose AND operands' result sizes the compiler needs
// (set_op.rs:388-393,439) and returns them per query.  Presence only: no scores are read.
static void run_count_queries(const Index& idx, Workspace& ws, const std::vector<CompiledQuery*>& cqs, std::vector<QueryCounts>& out, hipStream_t st) {
    const size_t n = cqs.size();
    out.assign(n, QueryCounts{});
    if (!n) return;
    std::vector<uint32_t> blob_off(n + 1, 0), span_base(n + 1, 0), qmap(n), counts_off(n + 1, 0);
    size_t lds_bytes = 0, desc_cap = 0;
    uint32_t stack_depth = 1;
    for (size_t i = 0; i < n; ++i) {
        const CompiledQuery& cq = *cqs[i];
        size_t desc = 0;
        const size_t bytes = pack_blob(cq, idx, nullptr, nullptr, 0, 0, {}, {}, &desc);
        blob_off[i + 1] = uint32_t(blob_off[i] + align_up(bytes, 16));
        span_base[i + 1] = span_base[i] + cq.n_spans;
        counts_off[i + 1] = counts_off[i] + cq.n_counts;
        qmap[i] = uint32_t(i);
        desc_cap = std::max(desc_cap, desc);
        stack_depth = std::max(stack_depth, cq.stack_depth);
    }
    desc_cap = align_up(desc_cap, 16);
    uint32_t list_table = 2;
    for (size_t i = 0; i < n; ++i) list_table = std::max<uint32_t>(list_table, uint32_t(cqs[i]->lists.size()));
    list_table = (list_table + 1u) & ~1u;
    const uint32_t cand_cap = 256;  // the candidate area doubles as the counter array (<= 256 counters)
    for (size_t i = 0; i < n; ++i)
        lds_bytes = std::max(lds_bytes, tile_scan_lds_bytes(uint32_t(cqs[i]->lists.size()) + cqs[i]->n_temps, uint32_t(cqs[i]->lists.size()), cqs[i]->tile_words,
                                                            stack_depth, cand_cap, uint32_t(desc_cap), false, list_table));
    if (lds_bytes > 160 * 1024) throw VelociError(ERR_UNSUPPORTED, "LDS tile larger than 160 KiB");
    const size_t o_off = align_up(blob_off[n], 256), o_span = o_off + align_up((n + 1) * 4, 256), o_qmap = o_span + align_up((n + 1) * 4, 256),
                 o_cnt = o_qmap + align_up(n * 4, 256), total = o_cnt + align_up(size_t(counts_off[n]) * 8, 256);
    std::vector<uint8_t> host(total, 0);
    ws.d_union_meta.ensure(total);
    uint8_t* dev = ws.d_union_meta.as<uint8_t>();
    for (size_t i = 0; i < n; ++i) pack_blob(*cqs[i], idx, host.data() + blob_off[i], dev + blob_off[i], 0, counts_off[i], {}, {});
    std::memcpy(host.data() + o_off, blob_off.data(), (n + 1) * 4);
    std::memcpy(host.data() + o_span, span_base.data(), (n + 1) * 4);
    std::memcpy(host.data() + o_qmap, qmap.data(), n * 4);
    VQ_HIP(hipMemcpyAsync(dev, host.data(), total, hipMemcpyHostToDevice, st));
    {
        uint64_t id_bytes = 0;  // presence only: the doc ids of every list
        for (size_t i = 0; i < n; ++i) id_bytes += 4ull * cqs[i]->total_len;
        LaunchTimer timer(idx.profile.enabled, ws, st, K_COUNT_PREPASS, id_bytes, id_bytes, n);
        launch_tile_scan(st, span_base[n], lds_bytes, dev, reinterpret_cast<const uint32_t*>(dev + o_off), reinterpret_cast<const uint32_t*>(dev + o_span),
                         reinterpret_cast<const uint32_t*>(dev + o_qmap), uint32_t(n), stack_depth, cand_cap, uint32_t(desc_cap), nullptr,
                         reinterpret_cast<unsigned long long*>(dev + o_cnt), nullptr, false, list_table);
    }
    VQ_HIP(hipGetLastError());
    std::vector<uint64_t> cnt(counts_off[n]);
    VQ_HIP(hipMemcpyAsync(cnt.data(), dev + o_cnt, cnt.size() * 8, hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < n; ++i) {
        const CompiledQuery& cq = *cqs[i];
        const uint64_t* c = cnt.data() + counts_off[i];
        out[i].has_filter = !cq.fops.empty();
        out[i].filter_count = c[cq.n_counts - 1];
        for (size_t k = 0; k < cq.count_nodes.size(); ++k) out[i].nodes[cq.count_nodes[k]] = {c[2 * k], c[2 * k + 1]};
    }
}


// blob and descriptor size of a compiled query (a dry run of pack_blob), kept with it: the serial part of a step does not walk every query three times
static void size_blob(CompiledQuery& cq, const Index& idx) {
    if (cq.status != 0) return;
    size_t d = 0;
    cq.blob_bytes = pack_blob(cq, idx, nullptr, nullptr, 0, 0, {}, {}, &d);
    cq.desc_bytes = d;
}

std::unique_ptr<PartialBatch> run_partial(const Index& idx, const vqreq::Request* const* reqs, size_t n, int slot, int64_t arena_offset) {
    const double t_start = now_ms();
    auto pb = std::make_unique<PartialBatch>();
    pb->index = &idx;
    pb->reqs.assign(reqs, reqs + n);
    pb->t0 = std::chrono::steady_clock::now();
    if (slot < 0) {
        // any workspace: the first free one from the round-robin position on, never one that a batch holds by name (the chunks of a sharded step
        // in flight: waiting for one of those on the thread that has to end the step would never return)
        const uint32_t start = idx.next_ws.fetch_add(1);
        int fallback = -1;
        for (uint32_t k = 0; k < uint32_t(kWorkspaces) && !pb->lock.owns_lock(); ++k) {
            Workspace& w = idx.ws[(start + k) % kWorkspaces];
            if (w.pinned.load(std::memory_order_acquire)) continue;
            if (fallback < 0) fallback = int((start + k) % kWorkspaces);
            std::unique_lock<std::mutex> l(w.mu, std::try_to_lock);
            if (l.owns_lock() && !w.pinned.load(std::memory_order_acquire)) {
                pb->ws = &w;
                pb->lock = std::move(l);
            }
        }
        if (!pb->lock.owns_lock()) {
            if (fallback < 0) throw vqreq::VelociError(vqreq::ERR_INVALID_ARGUMENT, "every workspace of the index is held by a sharded step in flight: end a step first");
            pb->ws = &idx.ws[fallback];  // held by another thread's batch: it will be handed on
            pb->lock = std::unique_lock<std::mutex>(pb->ws->mu);
        }
    } else {
        pb->ws = &idx.ws[slot % kWorkspaces];
        pb->lock = std::unique_lock<std::mutex>(pb->ws->mu, std::try_to_lock);
        if (!pb->lock.owns_lock()) {
            if (pb->ws->pinned.load(std::memory_order_acquire)) throw vqreq::VelociError(vqreq::ERR_INVALID_ARGUMENT, "the workspace named for this batch is held by a step in flight");
            pb->lock = std::unique_lock<std::mutex>(pb->ws->mu);
        }
        pb->ws->pinned.store(true, std::memory_order_release);
        pb->pinned_ws = true;
    }
    VQ_HIP(hipSetDevice(idx.device));
    Workspace& ws = *pb->ws;
    hipStream_t st = idx.stream;
    ws.timed.clear();
    ws.ev_used = 0;
    pb->profiled = idx.profile.enabled;

    // ---- dictionary scans (fuzzy / prefix leaves) of the whole batch, then compile
    FuzzyTable fuzzy;
    for (size_t i = 0; i < n; ++i)
        if (reqs[i]) collect_fuzzy_probes(idx, *reqs[i], fuzzy);
    static const bool pre_own = std::getenv("VQ_PRE_ON_SCAN_STREAM") == nullptr;
    hipStream_t pst = pre_own && idx.pre_stream ? idx.pre_stream : st;  // the pre-passes' stream (see Index::pre_stream)
    if (!fuzzy.empty()) run_fuzzy_probes(idx, ws, fuzzy, pst);
    const double t_probes = now_ms();
    pb->queries.reserve(n);
    pb->slot.assign(n, UINT32_MAX);
    pb->queries.resize(n);
    Boost1nCache boost_cache;  // resolved 1:n boost lists, shared by the batch's requests and compilation passes
    // Requests are compiled heaviest first: a request's cost follows the terms its prefix / fuzzy leaves matched (their posting lists, the 1:n boost
    // lists behind them), which the dictionary scans have just counted; one such request can take as long as a hundred others, and claimed last
    // it alone would be the end of the parallel pass.
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = uint32_t(i);
    const bool heavy_first = n >= 64 && !fuzzy.empty() && host_threads() > 1;
    if (heavy_first) {
        std::vector<uint64_t> weight(n, 0);
        std::function<void(const vqreq::SearchRequest&, uint64_t&)> walk = [&](const vqreq::SearchRequest& r, uint64_t& w) {
            if (r.kind == vqreq::SearchRequest::Search) {
                if (!needs_dictionary_scan(r.part)) return;
                auto it = fuzzy.find(fuzzy_key(r.part));
                if (it != fuzzy.end()) w += it->second.matches.size();
            } else
                for (auto& q : r.tree.queries) walk(q, w);
        };
        for (size_t i = 0; i < n; ++i)
            if (reqs[i] && reqs[i]->search_req) walk(*reqs[i]->search_req, weight[i]);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return weight[x] > weight[y]; });
    }
    auto compile_range = [&](size_t b, size_t e) {
        for (size_t k = b; k < e; ++k) {
            const size_t i = order[k];
            if (!reqs[i]) {
                pb->queries[i].status = ERR_INVALID_ARGUMENT;
                pb->queries[i].error = "null request";
            } else {
                pb->queries[i] = compile_query(idx, *reqs[i], fuzzy.empty() ? nullptr : &fuzzy, nullptr, nullptr, nullptr, &boost_cache);
                size_blob(pb->queries[i], idx);
            }
        }
    };
    if (n >= 64) {  // query compilation is independent per request: fan out over the index's host threads
        // Parts are claimed dynamically and shrink as the work runs out (guided scheduling: a part is 1/(2 x threads) of what is left, down to
        // one request): few claims while everybody is busy, single requests at the end — a worker that wakes late (an idle core takes ~0.1 ms,
        // a third of the whole job) still finds work, and requests of very different cost (a prefix leaf with a 1:n boost list takes 1000x a
        // plain AND) balance out.
        const std::vector<std::pair<size_t, size_t>> ranges = guided_ranges(n, host_threads(), heavy_first ? 4 * host_threads() : 0);
        host_pool(idx).run(ranges.size(), [&](size_t p) { compile_range(ranges[p].first, ranges[p].second); });
    } else compile_range(0, n);
    const double t_pass1 = now_ms();
    if (timing_enabled() && std::getenv("VQ_TIMING_SUB")) {  // (tools/host_profile.py stops at the first launch: the running totals, pass 1 only)
        std::fprintf(stderr, "[vq timing] pass 1 of %zu: %.3f ms wall;", n, t_pass1 - t_probes);
        for (int k = 0; k < 10; ++k) std::fprintf(stderr, " [%d] %.3f", k, g_compile_ns[k].load() * 1e-6);
        std::fprintf(stderr, "\n");
    }
    // ---- leaves that asked to be materialised first (K2): run the union jobs once per batch
    UnionTable unions;
    RangeTable ranges;
    LocalityTable localities;
    Boost1nTable boost1n;
    std::vector<size_t> again;
    for (size_t k = 0; k < n; ++k)
        if (const size_t i = order[k]; pb->queries[i].status == kStatusNeedsUnion || pb->queries[i].status == kStatusNeedsRanges) {
            again.push_back(i);
            for (auto& j : pb->queries[i].union_requests) unions.emplace(j.key, j);
            for (auto& j : pb->queries[i].locality_requests) localities.emplace(j.key, j);
            for (auto& j : pb->queries[i].boost1n_requests) boost1n.emplace(j.key, j);
        }
    double t_unions = t_pass1, t_ranges = t_pass1;
    if (!again.empty()) {
        if (!unions.empty()) {
            run_union_jobs(idx, ws, unions, pst);
            // merged lengths over all shards (the AND summation order follows them)
            std::vector<uint64_t> lens;
            for (auto& kv : unions) lens.push_back(kv.second.len);
            if (idx.can_sum_over_shards()) idx.sum_over_shards(lens);
            size_t k = 0;
            for (auto& kv : unions) kv.second.global_len = lens[k++];
        }
        if (!localities.empty()) run_locality_jobs(idx, ws, localities, pst);
        if (!boost1n.empty()) {
            run_boost1n_jobs(idx, ws, boost1n, pst);
            boost_cache.device = &boost1n;
        }
        t_unions = t_ranges = now_ms();
        // Compile again with the jobs' results.  A 1:n boost list only shows once it is resolved (K10) whether an anchor carries several values:
        // those leaves then ask for a range pre-pass — which of the values apply follows the leaf's hits around each anchor (k_range_hits, on the
        // merged list of a materialised leaf) — and are compiled a third time.
        for (int round = 0; round < 2 && !again.empty(); ++round) {
            bool any_ranges = false;
            for (size_t i : again)
                if (pb->queries[i].status == kStatusNeedsRanges) {
                    any_ranges = true;
                    for (auto& j : pb->queries[i].range_requests) ranges.emplace(j.key, j);
                }
            const bool ranges_ok = !any_ranges || !idx.sharded() || idx.can_sum_over_shards();
            if (any_ranges && ranges_ok) run_range_jobs(idx, ws, ranges, unions, pst);
            t_ranges = now_ms();
            auto recompile = [&](size_t b, size_t e) {
                for (size_t k = b; k < e; ++k) {
                    CompiledQuery& q = pb->queries[again[k]];
                    if (q.status == kStatusNeedsRanges && !ranges_ok) {
                        q.status = ERR_UNSUPPORTED;
                        q.error = "unsupported on the MI355X query path: 1:n field boost with several boosted values on one anchor, on a sharded index without vq_index_set_allreduce";
                        continue;
                    }
                    q = compile_query(idx, *reqs[again[k]], fuzzy.empty() ? nullptr : &fuzzy, unions.empty() ? nullptr : &unions, nullptr, ranges.empty() ? nullptr : &ranges, &boost_cache,
                                      localities.empty() ? nullptr : &localities);
                    if (q.status == kStatusNeedsUnion || (q.status == kStatusNeedsRanges && round == 1)) {
                        q.status = ERR_UNSUPPORTED;
                        q.error = "unsupported on the MI355X query path: leaf expansion changed between compilation passes (internal)";
                    }
                    size_blob(q, idx);
                }
            };
            if (again.size() >= 8 && host_threads() > 1) {
                const std::vector<std::pair<size_t, size_t>> ranges = guided_ranges(again.size(), host_threads(), heavy_first ? 4 * host_threads() : 0);
                host_pool(idx).run(ranges.size(), [&](size_t p) { recompile(ranges[p].first, ranges[p].second); });
            } else recompile(0, again.size());
            std::vector<size_t> still;
            for (size_t i : again)
                if (pb->queries[i].status == kStatusNeedsRanges) still.push_back(i);
            again.swap(still);
        }
    }
    const RangeTable* rangesp = ranges.empty() ? nullptr : &ranges;
    // ---- ANDs whose summation order / label follow run-time operand sizes: count pre-pass, then the final compilation
    {
        std::vector<size_t> need;
        std::vector<CompiledQuery*> cqs;
        for (size_t i = 0; i < n; ++i)
            if (pb->queries[i].status == kStatusNeedsCounts) {
                need.push_back(i);
                cqs.push_back(&pb->queries[i]);
            }
        if (!need.empty()) {
            std::vector<QueryCounts> counts;
            run_count_queries(idx, ws, cqs, counts, pst);
            if (idx.sharded()) {  // result sizes are sums over the shards
                std::vector<uint64_t> flat;
                for (auto& c : counts) {
                    flat.push_back(c.filter_count);
                    for (auto& kv : c.nodes) {
                        flat.push_back(kv.second.first);
                        flat.push_back(kv.second.second);
                    }
                }
                idx.sum_over_shards(flat);
                size_t k = 0;
                for (auto& c : counts) {
                    c.filter_count = flat[k++];
                    for (auto& kv : c.nodes) {
                        kv.second.first = flat[k++];
                        kv.second.second = flat[k++];
                    }
                }
            }
            for (size_t k = 0; k < need.size(); ++k) {
                CompiledQuery& q = pb->queries[need[k]];
                q = compile_query(idx, *reqs[need[k]], fuzzy.empty() ? nullptr : &fuzzy, unions.empty() ? nullptr : &unions, &counts[k], rangesp, &boost_cache,
                                  localities.empty() ? nullptr : &localities);
                if (q.status < 0) {
                    q.status = ERR_UNSUPPORTED;
                    q.error = "unsupported on the MI355X query path: query still needs a pre-pass after the count pre-pass (internal)";
                }
            }
        }
    }
    const double t_compiled = now_ms();
    // ---- layout
    uint32_t nq = 0;
    uint64_t total_keys = 0, total_hist = 0, total_span_keys = 0, total_spans = 0, blob_bytes = 0;
    uint32_t max_lists = 1, max_ww = 32;
    size_t lds_bytes = 0;
    uint32_t stack_depth = 1;
    std::vector<uint32_t> keys_base, part_keys_off, span_base;
    std::vector<std::vector<uint32_t>> hist_offs, fac_out_offs;
    std::vector<FacetJob> jobs;
    uint32_t fac_out_total = 0;
    {   // a small batch (one query = the latency case) would leave most of the chip idle with spans sized for streaming efficiency:
        // split its queries further until the launch holds about one wave per SIMD of every CU
        static const uint64_t target1 = [] {
            const char* e = std::getenv("VQ_SPAN_TARGET");
            return uint64_t(e ? std::atoll(e) : 2048);
        }();
        // (one request: 2048 spans — its merge is serial in the span count; more requests merge in parallel: up to one wave per slot)
        const uint64_t target = std::min<uint64_t>(target1 + 64 * uint64_t(n - 1), std::max<uint64_t>(target1, 5120));
        uint64_t have = 0;
        for (size_t i = 0; i < n; ++i)
            if (pb->queries[i].status == 0) have += pb->queries[i].n_spans;
        if (have && have * 3 <= target * 2) {
            const uint64_t f = (target + have - 1) / have;
            for (size_t i = 0; i < n; ++i) {
                CompiledQuery& cq = pb->queries[i];
                if (cq.status == 0) cq.n_spans = uint32_t(std::max<uint64_t>(cq.n_spans, std::min<uint64_t>(uint64_t(cq.n_spans) * f, cq.max_spans)));
            }
        }
        // Requests of very different weight in one launch (a prefix leaf over a third of the documents beside exact matches of a few): the launch
        // ends with the longest span, so a request gets spans in proportion to its postings — as many as keep every span of the launch near
        // total / target postings, but no span below 4096 (bench_jmdict shape, 256 requests: k_tile_scan 4.1 -> 0.8 ms)
        static const bool weighted = std::getenv("VQ_NO_WEIGHTED_SPANS") == nullptr;
        uint64_t total_postings = 0;
        for (size_t i = 0; i < n; ++i)
            if (pb->queries[i].status == 0) total_postings += pb->queries[i].total_len;
        if (weighted && n > 1 && total_postings) {
            const uint64_t per_span = std::max<uint64_t>(total_postings / target, 4096);
            for (size_t i = 0; i < n; ++i) {
                CompiledQuery& cq = pb->queries[i];
                if (cq.status != 0) continue;
                const uint64_t want = std::min<uint64_t>((cq.total_len + per_span - 1) / per_span, cq.max_spans);
                cq.n_spans = uint32_t(std::max<uint64_t>(cq.n_spans, want));
            }
        }
    }
    for (size_t i = 0; i < n; ++i) {
        CompiledQuery& cq = pb->queries[i];
        if (cq.status != 0) continue;
        pb->slot[i] = nq++;
        keys_base.push_back(uint32_t(total_span_keys));
        part_keys_off.push_back(uint32_t(total_keys));
        span_base.push_back(uint32_t(total_spans));
        total_span_keys += uint64_t(cq.n_spans) * cq.top_k;
        total_keys += cq.top_k;
        total_spans += cq.n_spans;
        std::vector<uint32_t> ho, fo;
        for (auto& f : cq.facets) {
            ho.push_back(uint32_t(total_hist));
            fo.push_back(fac_out_total);
            jobs.push_back(FacetJob{uint32_t(total_hist), f.num_values, f.top, fac_out_total});
            total_hist += f.num_values;
            fac_out_total += f.top;
        }
        hist_offs.push_back(std::move(ho));
        fac_out_offs.push_back(std::move(fo));
        if (!cq.blob_bytes) size_blob(cq, idx);  // (normally taken on the compiling thread)
        blob_bytes += cq.blob_bytes;
        max_lists = std::max<uint32_t>(max_lists, uint32_t(cq.lists.size()));
        max_ww = std::max(max_ww, cq.tile_words);
        stack_depth = std::max(stack_depth, cq.stack_depth);
    }
    if (total_span_keys > 0xFFFFFFFFull || total_hist > 0xFFFFFFFFull || total_spans > 0x7FFFFFFFull)
        throw VelociError(ERR_UNSUPPORTED, "batch too large for 32-bit workspace offsets: split the batch");
    span_base.push_back(uint32_t(total_spans));
    pb->nq_dev = nq;
    pb->total_spans = uint32_t(total_spans);
    pb->n_facet_jobs = uint32_t(jobs.size());
    pb->total_facet_out = fac_out_total;
    pb->facet_jobs = jobs;
    const double t_layout = now_ms();

    PartialLayout& lay = pb->layout;
    lay.nq = nq;
    lay.total_keys = total_keys;
    lay.total_hist = total_hist;
    lay.off_hits = 0;
    lay.off_stats = align_up(size_t(nq) * 8, 16);
    lay.off_keys = lay.off_stats + align_up(size_t(nq) * 8, 16);
    lay.off_hist = align_up(lay.off_keys + size_t(total_keys) * 8, 256);  // == bytes of the all-gathered part
    lay.bytes = align_up(lay.off_hist + size_t(total_hist) * 4, 256);

    // ---- upload area: [blobs][blob_off][span_base][facet jobs]
    const size_t up_blob_off = align_up(blob_bytes, 256);
    // two scan launches: pure simple queries (k_scan_simple) and everything else (k_tile_scan); each has its own
    // span table (prefix sums of n_spans over its queries) and a map from its local query index to the blob slot
    const size_t tbl = align_up(size_t(nq + 1) * 4, 256);
    const size_t up_span_base = up_blob_off + tbl;   // generic: span_base_g
    const size_t up_qmap_g = up_span_base + tbl;
    const size_t up_span_s = up_qmap_g + tbl;
    const size_t up_qmap_s = up_span_s + tbl;
    const size_t up_span_d = up_qmap_s + tbl;   // every posting is a hit (single leaves): k_scan_union
    const size_t up_qmap_d = up_span_d + tbl;
    const size_t up_span_w = up_qmap_d + tbl;   // simple ANDs: k_scan_simple with 16384-doc tiles
    const size_t up_qmap_w = up_span_w + tbl;
    const size_t up_span_r = up_qmap_w + tbl;   // rich simple queries (DSimple2): k_scan_simple<2, true>
    const size_t up_qmap_r = up_span_r + tbl;
    const size_t up_span_f = up_qmap_r + tbl;   // one materialised leaf: k_scan_leaf_f32
    const size_t up_qmap_f = up_span_f + tbl;
    const size_t up_span_x = up_qmap_f + tbl;   // wide queries (DWide): k_scan_wide
    const size_t up_qmap_x = up_span_x + tbl;
    const size_t up_span_p = up_qmap_x + tbl;   // ANDs of one id-list cover and bitmap operands: k_scan_probe
    const size_t up_qmap_p = up_span_p + tbl;
    const size_t up_qmap_n = up_qmap_p + tbl;   // ... with top + skip <= 32: k_scan_ring (a persistent grid: no span table, (query, span) items instead)
    const size_t up_work_n = up_qmap_n + tbl;   // its item counter and error word (zeroed with every upload)
    size_t ring_items_max = 0;                  // its item table: only when the launch's queries differ in their span counts
    for (size_t i = 0; i < n; ++i)
        if (pb->queries[i].status == 0 && ((pb->queries[i].simple_flags >> 26) & 1u)) ring_items_max += pb->queries[i].n_spans;
    const size_t up_items_n = up_work_n + 256;
    const size_t up_jobs = up_items_n + align_up(ring_items_max * 4, 256);
    const size_t up_bytes = up_jobs + align_up(jobs.size() * sizeof(FacetJob), 256) + 256;
    ws.h_up.ensure(up_bytes);
    ws.d_up.ensure(up_bytes);
    uint8_t* hup = ws.h_up.as<uint8_t>();
    uint8_t* dup = ws.d_up.as<uint8_t>();
    bool union_has_or = false;
    uint32_t scatter_wide = 0, scatter_simple = 0;  // id (scattered) lists per query: they alone need an LDS tile in k_scan_simple
    uint32_t n_simple = 0, n_generic = 0, n_dense = 0, n_wide = 0, n_rich = 0, spans_simple = 0, spans_generic = 0, spans_dense = 0, spans_wide = 0, spans_rich = 0;
    uint32_t scatter_rich = 0;
    bool facets_rich = false;
    uint32_t n_leaf = 0, spans_leaf = 0;
    uint32_t n_xwide = 0, spans_xwide = 0, leaves_xwide = 0, scatter_xwide = 0;
    uint32_t n_probe = 0, spans_probe = 0, nd_probe = 1;
    bool probe_any_and = false, probe_any_or = false;
    uint32_t n_ring = 0, items_ring = 0, nd_ring = 1, ring_spans_each = 0;
    uint64_t cls_layout[K_COUNT_] = {}, cls_algo[K_COUNT_] = {}, cls_q[K_COUNT_] = {};
    {
        size_t off = 0;
        uint32_t* hbo = reinterpret_cast<uint32_t*>(hup + up_blob_off);
        uint32_t qi = 0;
        for (size_t i = 0; i < n; ++i) {
            CompiledQuery& cq = pb->queries[i];
            if (cq.status != 0) continue;
            hbo[qi] = uint32_t(off);
            const size_t packed = pack_blob(cq, idx, hup + off, dup + off, keys_base[qi], part_keys_off[qi], hist_offs[qi], fac_out_offs[qi], nullptr,
                                            pb->profiled ? uint32_t((lay.off_stats - lay.off_hits) / 8 + qi) : 0u);  // 0: the kernels count nothing
            if (packed != cq.blob_bytes) throw VelociError(ERR_DEVICE, "query blob changed size between compilation and packing (internal)");
            off += packed;
            ++qi;
        }
        hbo[nq] = uint32_t(off);
        uint32_t* sg = reinterpret_cast<uint32_t*>(hup + up_span_base);
        uint32_t* mg = reinterpret_cast<uint32_t*>(hup + up_qmap_g);
        uint32_t* ss = reinterpret_cast<uint32_t*>(hup + up_span_s);
        uint32_t* ms = reinterpret_cast<uint32_t*>(hup + up_qmap_s);
        uint32_t* sd = reinterpret_cast<uint32_t*>(hup + up_span_d);
        uint32_t* md = reinterpret_cast<uint32_t*>(hup + up_qmap_d);
        uint32_t* sw = reinterpret_cast<uint32_t*>(hup + up_span_w);
        uint32_t* mw = reinterpret_cast<uint32_t*>(hup + up_qmap_w);
        uint32_t* sr = reinterpret_cast<uint32_t*>(hup + up_span_r);
        uint32_t* mr = reinterpret_cast<uint32_t*>(hup + up_qmap_r);
        uint32_t* sf = reinterpret_cast<uint32_t*>(hup + up_span_f);
        uint32_t* mf = reinterpret_cast<uint32_t*>(hup + up_qmap_f);
        uint32_t accf = 0, accx = 0;
        uint32_t* sx = reinterpret_cast<uint32_t*>(hup + up_span_x);
        uint32_t* mx = reinterpret_cast<uint32_t*>(hup + up_qmap_x);
        uint32_t accg = 0, accs = 0, accd = 0, accw = 0, accr = 0, accp = 0;
        uint32_t* sp = reinterpret_cast<uint32_t*>(hup + up_span_p);
        uint32_t* mp = reinterpret_cast<uint32_t*>(hup + up_qmap_p);
        uint32_t* mn = reinterpret_cast<uint32_t*>(hup + up_qmap_n);
        uint32_t ring_max_spans = 0;
        std::vector<uint32_t> ring_ns;  // spans of k_scan_ring's queries
        bool ring_uniform = true;
        std::memset(hup + up_work_n, 0, 256);
        qi = 0;
        for (size_t i = 0; i < n; ++i) {
            const CompiledQuery& cq = pb->queries[i];
            if (cq.status != 0) continue;
            // every posting is a hit: k_scan_union streams the scores with the doc ids.  Single leaves always (VQ_NO_UNION=1 turns it
            // off); its two-pass OR is correct but not yet faster than the survivor queue of k_scan_simple (50 vs 40 ms per 256
            // 3-term ORs on 100 M docs), so ORs take it only with VQ_UNION_OR=1
            static const bool union_enabled = std::getenv("VQ_NO_UNION") == nullptr;
            static const bool union_or = std::getenv("VQ_UNION_OR") != nullptr;
            const bool rich = (cq.simple_flags >> 18) & 1u;
            const bool dense = !rich && union_enabled && cq.simple_flags && (cq.simple_n == 1 || (union_or && cq.ops.back().kind == OP_OR));
            int kclass;
            if ((cq.simple_flags >> 19) & 1u) {
                kclass = K_SCAN_LEAF_F32;
                sf[n_leaf] = accf;
                mf[n_leaf++] = qi;
                accf += cq.n_spans;
            } else if ((cq.simple_flags >> 24) & 1u) {
                kclass = K_SCAN_WIDE;
                leaves_xwide = std::max<uint32_t>(leaves_xwide, cq.wide.n_leaves);
                scatter_xwide = std::max<uint32_t>(scatter_xwide, cq.wide.n_leaves - uint32_t(__builtin_popcount(cq.wide.bitmap_mask)));
                sx[n_xwide] = accx;
                mx[n_xwide++] = qi;
                accx += cq.n_spans;
            } else if (rich) {
                kclass = K_SCAN_RICH;
                facets_rich = facets_rich || !cq.facets.empty();
                scatter_rich = std::max<uint32_t>(scatter_rich, cq.simple_n - uint32_t(__builtin_popcount(cq.simple_flags & 0xFu)) + cq.simple2.n_side);
                sr[n_rich] = accr;
                mr[n_rich++] = qi;
                accr += cq.n_spans;
            } else if (dense) {
                kclass = K_SCAN_UNION;
                union_has_or = union_has_or || cq.simple_n > 1;
                sd[n_dense] = accd;
                md[n_dense++] = qi;
                accd += cq.n_spans;
            } else if ((cq.simple_flags >> 26) & 1u) {
                kclass = K_SCAN_RING;
                nd_ring = std::max<uint32_t>(nd_ring, cq.simple_n - 1);
                ring_uniform = ring_uniform && (n_ring == 0 || cq.n_spans == ring_max_spans);
                ring_max_spans = std::max(ring_max_spans, cq.n_spans);
                mn[n_ring++] = qi;
                ring_ns.push_back(cq.n_spans);
                items_ring += cq.n_spans;
            } else if ((cq.simple_flags >> 25) & 1u) {
                kclass = K_SCAN_PROBE;
                ((cq.simple_flags >> 27) & 1u ? probe_any_or : probe_any_and) = true;
                nd_probe = std::max<uint32_t>(nd_probe, cq.simple_n - 1);
                sp[n_probe] = accp;
                mp[n_probe++] = qi;
                accp += cq.n_spans;
            } else if (cq.simple_flags && cq.simple_n > 1 && cq.ops.back().kind == OP_AND) {
                kclass = K_SCAN_AND;
                scatter_wide = std::max<uint32_t>(scatter_wide, cq.simple_n - uint32_t(__builtin_popcount(cq.simple_flags & 0xFu)));
                sw[n_wide] = accw;
                mw[n_wide++] = qi;
                accw += cq.n_spans;
            } else if (cq.simple_flags) {
                kclass = K_SCAN_SIMPLE;
                scatter_simple = std::max<uint32_t>(scatter_simple, cq.simple_n - uint32_t(__builtin_popcount(cq.simple_flags & 0xFu)));
                ss[n_simple] = accs;
                ms[n_simple++] = qi;
                accs += cq.n_spans;
            } else {
                kclass = K_TILE_SCAN;
                sg[n_generic] = accg;
                mg[n_generic++] = qi;
                accg += cq.n_spans;
            }
            pb->qclass.push_back(uint8_t(kclass));
            cls_layout[kclass] += cq.layout_bytes;
            cls_algo[kclass] += cq.algorithmic_bytes;
            cls_q[kclass] += 1;
            ++qi;
        }
        sg[n_generic] = accg;
        ss[n_simple] = accs;
        sw[n_wide] = accw;
        spans_wide = accw;
        sr[n_rich] = accr;
        spans_rich = accr;
        sf[n_leaf] = accf;
        spans_leaf = accf;
        sx[n_xwide] = accx;
        spans_xwide = accx;
        sp[n_probe] = accp;
        spans_probe = accp;
        if (n_ring) {  // k_scan_ring's items in round-major order: span 0 of every query, then span 1, ... (a query's pool is warm after its first span)
            if (ring_uniform) ring_spans_each = ring_max_spans;
            else {
                uint32_t* it = reinterpret_cast<uint32_t*>(hup + up_items_n);
                uint32_t k = 0;
                for (uint32_t r = 0; r < ring_max_spans; ++r)
                    for (uint32_t j = 0; j < n_ring; ++j)
                        if (r < ring_ns[j]) it[k++] = (j << 12) | r;
            }
        }
        sd[n_dense] = accd;
        spans_generic = accg;
        spans_simple = accs;
        spans_dense = accd;
        if (!jobs.empty()) std::memcpy(hup + up_jobs, jobs.data(), jobs.size() * sizeof(FacetJob));
    }
    pb->d_blobs = dup;
    pb->d_blob_off = reinterpret_cast<const uint32_t*>(dup + up_blob_off);
    pb->d_span_base = reinterpret_cast<const uint32_t*>(dup + up_span_base);
    pb->d_facet_jobs = reinterpret_cast<const FacetJob*>(dup + up_jobs);
    if (nq == 0) return pb;

    const double t_packed = now_ms();
    VQ_HIP(hipMemcpyAsync(dup, hup, up_bytes, hipMemcpyHostToDevice, st));
    ws.d_span_keys.ensure(size_t(total_span_keys) * 8 + 16);
    if (arena_offset >= 0) {  // a chunk of a sharded step with one collective: its partial lives in the index's arena
        if (size_t(arena_offset) % 256 || size_t(arena_offset) + lay.bytes > Index::kArenaBytes)
            throw VelociError(ERR_UNSUPPORTED, "partial arena: the step's partials do not fit (" + std::to_string(size_t(arena_offset) + lay.bytes) + " bytes)");
        idx.arena.ensure(Index::kArenaBytes);
        pb->d_partial = idx.arena.as<uint8_t>() + arena_offset;
    } else {
        ws.d_partial.ensure(lay.bytes);
        pb->d_partial = ws.d_partial.as<uint8_t>();
    }
    VQ_HIP(hipMemsetAsync(pb->d_partial, 0, lay.bytes, st));

    // ---- the scan
    uint32_t max_top_k = 1;
    for (size_t i = 0; i < n; ++i)
        if (pb->queries[i].status == 0) max_top_k = std::max(max_top_k, pb->queries[i].top_k);
    uint32_t desc_cap = 0;  // bytes of the largest query descriptor (staged into LDS by every workgroup)
    for (size_t i = 0; i < n; ++i)
        if (pb->queries[i].status == 0) {
            desc_cap = std::max(desc_cap, uint32_t(pb->queries[i].desc_bytes));
        }
    desc_cap = uint32_t(align_up(desc_cap, 16));
    bool facets_generic = false;  // k_tile_scan queries with facets: room for the LDS counter cache behind the descriptor
    static const bool no_facet_cache = std::getenv("VQ_NO_FACET_CACHE") != nullptr;
    for (size_t i = 0; i < n; ++i)
        if (pb->queries[i].status == 0 && !pb->queries[i].simple_flags && !pb->queries[i].facets.empty()) facets_generic = !no_facet_cache;
    if (facets_generic) desc_cap += 2 * 1024 * 4;
    static const uint32_t cand_min = [] {
        const char* e = std::getenv("VQ_CAND_CAP");  // (a small buffer is pruned — and its threshold raised — sooner: 64 beats 256 by 2-7 %, 32 beats 64 by 1-2 %)
        return uint32_t(e ? std::max(32, std::atoi(e)) : 32);
    }();
    uint32_t cand_cap = cand_min;  // power of two >= 2 * top_k: candidate keys a workgroup keeps in LDS
    while (cand_cap < 2 * max_top_k) cand_cap <<= 1;
    const uint32_t list_table = (std::max<uint32_t>(max_lists, 2) + 1u) & ~1u;  // k_tile_scan sizes its per-list LDS arrays to the launch's longest list table
    static const bool tile_queue = std::getenv("VQ_NO_QUEUE") == nullptr;  // k_tile_scan: survivors of several tiles share a scoring round
    for (size_t i = 0; i < n; ++i)
        if (pb->queries[i].status == 0 && !pb->queries[i].simple_flags)
            lds_bytes = std::max(lds_bytes, tile_scan_lds_bytes(uint32_t(pb->queries[i].lists.size()) + pb->queries[i].n_temps, uint32_t(pb->queries[i].lists.size()), pb->queries[i].tile_words, stack_depth, cand_cap, desc_cap, tile_queue && !pb->queries[i].simple_n, list_table));
    if (lds_bytes > 160 * 1024) throw VelociError(ERR_UNSUPPORTED, "LDS tile larger than 160 KiB");
    const bool prof = pb->profiled;
    auto hits_ptr = reinterpret_cast<unsigned long long*>(pb->d_partial + lay.off_hits);
    auto hist_ptr = reinterpret_cast<uint32_t*>(pb->d_partial + lay.off_hist);
    auto keys_ptr = ws.d_span_keys.as<unsigned long long>();
    auto tab = [&](size_t o) { return reinterpret_cast<const uint32_t*>(dup + o); };
    if (spans_leaf) {
        LaunchTimer t(prof, ws, st, K_SCAN_LEAF_F32, cls_layout[K_SCAN_LEAF_F32], cls_algo[K_SCAN_LEAF_F32], cls_q[K_SCAN_LEAF_F32]);
        launch_scan_leaf_f32(st, spans_leaf, pb->d_blobs, pb->d_blob_off, tab(up_span_f), tab(up_qmap_f), n_leaf, cand_cap, keys_ptr, hits_ptr, hist_ptr);
    }
    VQ_HIP(hipGetLastError());
    if (spans_rich) {
        LaunchTimer t(prof, ws, st, K_SCAN_RICH, cls_layout[K_SCAN_RICH], cls_algo[K_SCAN_RICH], cls_q[K_SCAN_RICH]);
        launch_scan_simple(st, true, scatter_rich, spans_rich, pb->d_blobs, pb->d_blob_off, tab(up_span_r), tab(up_qmap_r), n_rich, cand_cap, keys_ptr, hits_ptr, hist_ptr, facets_rich);
    }
    VQ_HIP(hipGetLastError());
    if (items_ring) {
        static const uint32_t cus = [] {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
            return uint32_t(v);
        }();
        LaunchTimer t(prof, ws, st, K_SCAN_RING, cls_layout[K_SCAN_RING], cls_algo[K_SCAN_RING], cls_q[K_SCAN_RING]);
        launch_scan_ring(st, nd_ring, cus, pb->d_blobs, pb->d_blob_off, tab(up_qmap_n), n_ring, ring_spans_each, tab(up_items_n), items_ring,
                         reinterpret_cast<uint32_t*>(dup + up_work_n), keys_ptr, hits_ptr);
    }
    VQ_HIP(hipGetLastError());
    if (spans_probe) {
        LaunchTimer t(prof, ws, st, K_SCAN_PROBE, cls_layout[K_SCAN_PROBE], cls_algo[K_SCAN_PROBE], cls_q[K_SCAN_PROBE]);
        launch_scan_probe(st, nd_probe, spans_probe, pb->d_blobs, pb->d_blob_off, tab(up_span_p), tab(up_qmap_p), n_probe, cand_cap, keys_ptr, hits_ptr, probe_any_and, probe_any_or);
    }
    VQ_HIP(hipGetLastError());
    if (spans_wide) {
        LaunchTimer t(prof, ws, st, K_SCAN_AND, cls_layout[K_SCAN_AND], cls_algo[K_SCAN_AND], cls_q[K_SCAN_AND]);
        launch_scan_simple(st, false, scatter_wide, spans_wide, pb->d_blobs, pb->d_blob_off, tab(up_span_w), tab(up_qmap_w), n_wide, cand_cap, keys_ptr, hits_ptr, hist_ptr);
    }
    VQ_HIP(hipGetLastError());
    // (16384-doc tiles pay off for ORs too once LDS no longer bounds the occupancy)
    if (spans_simple) {
        LaunchTimer t(prof, ws, st, K_SCAN_SIMPLE, cls_layout[K_SCAN_SIMPLE], cls_algo[K_SCAN_SIMPLE], cls_q[K_SCAN_SIMPLE]);
        launch_scan_simple(st, false, scatter_simple, spans_simple, pb->d_blobs, pb->d_blob_off, tab(up_span_s), tab(up_qmap_s), n_simple, cand_cap, keys_ptr, hits_ptr, hist_ptr);
    }
    VQ_HIP(hipGetLastError());
    if (spans_dense) {
        LaunchTimer t(prof, ws, st, K_SCAN_UNION, cls_layout[K_SCAN_UNION], cls_algo[K_SCAN_UNION], cls_q[K_SCAN_UNION]);
        launch_scan_union(st, union_has_or, spans_dense, pb->d_blobs, pb->d_blob_off, tab(up_span_d), tab(up_qmap_d), n_dense, cand_cap, keys_ptr, hits_ptr);
    }
    VQ_HIP(hipGetLastError());
    if (spans_xwide) {
        LaunchTimer t(prof, ws, st, K_SCAN_WIDE, cls_layout[K_SCAN_WIDE], cls_algo[K_SCAN_WIDE], cls_q[K_SCAN_WIDE]);
        launch_scan_wide(st, leaves_xwide, scatter_xwide, spans_xwide, pb->d_blobs, pb->d_blob_off, tab(up_span_x), tab(up_qmap_x), n_xwide, cand_cap, keys_ptr, hits_ptr);
    }
    VQ_HIP(hipGetLastError());
    if (spans_generic) {
        LaunchTimer t(prof, ws, st, K_TILE_SCAN, cls_layout[K_TILE_SCAN], cls_algo[K_TILE_SCAN], cls_q[K_TILE_SCAN]);
        launch_tile_scan(st, spans_generic, lds_bytes, pb->d_blobs, pb->d_blob_off, pb->d_span_base, tab(up_qmap_g), n_generic, stack_depth, cand_cap, desc_cap, keys_ptr,
                         hits_ptr, hist_ptr, tile_queue, list_table, facets_generic);
    }
    VQ_HIP(hipGetLastError());
    {
        LaunchTimer t(prof, ws, st, K_MERGE_SPANS, total_span_keys * 8 + total_keys * 8, total_span_keys * 8 + total_keys * 8, nq);
        launch_merge_spans(st, nq, pb->d_blobs, pb->d_blob_off, keys_ptr, reinterpret_cast<unsigned long long*>(pb->d_partial + lay.off_keys));
    }
    VQ_HIP(hipGetLastError());
    VQ_HIP(hipEventRecord(ws.ev_done, st));
    pb->launched = true;
    if (timing_enabled())
        std::fprintf(stderr, "[vq timing] n=%zu compile %.3f ms (dictionary scans %.3f [%zu probes], pass 1 %.3f, unions %.3f [%zu jobs], pass 2 %.3f), range jobs %.3f [%zu], spans generic/simple/and/rich/union %u/%u/%u/%u/%u, pack+launch %.3f ms\n", n,
                     t_compiled - t_start, t_probes - t_start, fuzzy.size(), t_pass1 - t_probes, t_unions - t_pass1, unions.size(), t_compiled - t_ranges,
                     t_ranges - t_unions, ranges.size(), spans_generic, spans_simple, spans_wide, spans_rich, spans_dense + spans_leaf, now_ms() - t_compiled);
    if (timing_enabled())
        std::fprintf(stderr, "[vq timing] pack+launch: span sizing + layout %.3f, blobs + tables %.3f, upload + launches %.3f ms\n", t_layout - t_compiled, t_packed - t_layout, now_ms() - t_packed);
    if (timing_enabled()) {  // thread-time inside compile_query since the last batch (all passes, all threads)
        uint64_t v[16];
        for (int k = 0; k < 16; ++k) v[k] = g_compile_ns[k].exchange(0);
        if (std::getenv("VQ_TIMING_SUB")) {
            std::fprintf(stderr, "[vq timing] inside 1:n resolve (thread-ms; 6 value-id gather, 7 sort, 8 pairs, 9 order check + layers):");
            for (int k = 6; k < 10; ++k) std::fprintf(stderr, " [%d] %.3f", k, v[k] * 1e-6);
            std::fprintf(stderr, "\n");
        }
        std::fprintf(stderr, "[vq timing] compile thread-ms: total %.3f = dictionary lookups %.3f + 1:n resolve %.3f + 1:n layers %.3f + leaf lists %.3f + rest %.3f; longest request %.3f\n", v[5] * 1e-6,
                     v[0] * 1e-6, v[1] * 1e-6, (v[2] - v[1]) * 1e-6, v[3] * 1e-6, (double(v[5]) - double(v[0]) - double(v[2]) - double(v[3])) * 1e-6, v[4] * 1e-6);
    }
    return pb;
}

// suggest_multi (search_field.rs:194-219): dictionary side only — the parts' matched terms, equal texts merged keeping the best score, ranked
std::vector<SuggestEntry> run_suggest(const Index& idx, const vqreq::Request& req) {
    if (!req.suggest) throw VelociError(ERR_INVALID_REQUEST, "only suggest allowed in suggest function");
    VQ_HIP(hipSetDevice(idx.device));
    FuzzyTable fuzzy;
    collect_suggest_probes(idx, req, fuzzy);
    if (!fuzzy.empty()) {
        Workspace& ws = idx.ws[idx.next_ws.fetch_add(1) % kWorkspaces];
        std::unique_lock<std::mutex> lock(ws.mu);
        ws.timed.clear();
        ws.ev_used = 0;
        run_fuzzy_probes(idx, ws, fuzzy, idx.pre_stream ? idx.pre_stream : idx.stream);
    }
    std::vector<SuggestEntry> out;
    for (auto& part : *req.suggest) {
        auto one = suggest_part(idx, part, fuzzy.empty() ? nullptr : &fuzzy);
        out.insert(out.end(), one.begin(), one.end());
    }
    std::stable_sort(out.begin(), out.end(), [](const SuggestEntry& a, const SuggestEntry& b) { return a.text > b.text; });  // :176 (descending)
    std::vector<SuggestEntry> merged;
    for (auto& e : out) {
        if (!merged.empty() && merged.back().text == e.text) {
            if (e.score > merged.back().score) merged.back().score = e.score;
        } else merged.push_back(e);
    }
    std::stable_sort(merged.begin(), merged.end(), [](const SuggestEntry& a, const SuggestEntry& b) { return a.score > b.score; });  // :189
    const size_t skip = std::min(req.skip.value_or(0), merged.size());  // apply_top_skip, search.rs:230-239
    merged.erase(merged.begin(), merged.begin() + skip);
    if (req.top && merged.size() > *req.top) merged.resize(*req.top);
    return merged;
}

// search_field::highlight (search_field.rs:233-245): the part's terms normalised (util.rs:11-29), its dictionary scan on the device, the snippets on
// the host, ranked by score with the part's own top / skip
std::vector<SuggestEntry> run_highlight(const Index& idx, vqreq::RequestSearchPart part) {
    for (auto& t : part.terms) t = vqtext::normalize_text(t);
    VQ_HIP(hipSetDevice(idx.device));
    vqreq::Request probe_req;
    probe_req.suggest = std::vector<vqreq::RequestSearchPart>{part};
    FuzzyTable fuzzy;
    collect_suggest_probes(idx, probe_req, fuzzy);
    if (!fuzzy.empty()) {
        Workspace& ws = idx.ws[idx.next_ws.fetch_add(1) % kWorkspaces];
        std::unique_lock<std::mutex> lock(ws.mu);
        ws.timed.clear();
        ws.ev_used = 0;
        run_fuzzy_probes(idx, ws, fuzzy, idx.pre_stream ? idx.pre_stream : idx.stream);
    }
    std::vector<SuggestEntry> out = highlight_part(idx, part, fuzzy.empty() ? nullptr : &fuzzy);
    std::stable_sort(out.begin(), out.end(), [](const SuggestEntry& a, const SuggestEntry& b) { return a.score > b.score; });  // :189
    const size_t skip = std::min(part.skip.value_or(0), out.size());  // apply_top_skip, search.rs:230-239
    out.erase(out.begin(), out.begin() + skip);
    if (part.top && out.size() > *part.top) out.resize(*part.top);
    return out;
}

// The continuation of a request behind the ranked hit (score, id): the next kMaxTopK hits below that key, no facets (page 0 counted them)
vqreq::Request page_request_after(const vqreq::Request& R, float score, uint32_t id) {
    vqreq::Request page = R;
    page.top = size_t(kMaxTopK);
    page.skip = 0;
    page.facets.reset();
    uint32_t bits;
    std::memcpy(&bits, &score, 4);
    page.key_upper = (uint64_t(order_f32(bits)) << 32) | id;
    return page;
}

void complete_deep_requests(const Index& idx, const vqreq::Request* const* reqs, size_t n, std::vector<std::unique_ptr<Result>>& results,
                            std::vector<int>& status, std::vector<std::string>& errors) {
    constexpr uint64_t kMaxDeep = 65536;  // ranked hits one request may reach (64 scans)
    for (size_t i = 0; i < n; ++i) {
        if (status[i] != 0 || !results[i] || !results[i]->deep) continue;
        const vqreq::Request& R = *reqs[i];
        Result& out = *results[i];
        out.deep = false;
        const uint64_t top = R.top.value_or(10), skip = R.skip.value_or(0);
        const uint64_t want = top + skip < top ? ~0ull : top + skip;
        std::vector<uint32_t> ids = std::move(out.ids);  // page 0: the best kMaxTopK
        std::vector<float> scores = std::move(out.scores);
        out.ids.clear();
        out.scores.clear();
        if (skip >= out.num_hits) continue;  // apply_top_skip (search.rs:230-239): nothing left behind the skipped hits
        const uint64_t reach = std::min<uint64_t>(want, out.num_hits);
        if (reach > kMaxDeep) {
            status[i] = ERR_UNSUPPORTED;
            errors[i] = "unsupported on the MI355X query path: top + skip reaches more than " + std::to_string(kMaxDeep) + " ranked hits";
            results[i].reset();
            continue;
        }
        vqreq::Request page = R;
        page.top = size_t(kMaxTopK);
        page.skip = 0;
        page.facets.reset();  // (counted by page 0)
        bool failed = false;
        while (ids.size() < reach && ids.size() % size_t(kMaxTopK) == 0 && !ids.empty()) {
            uint32_t bits;
            std::memcpy(&bits, &scores.back(), 4);
            page.key_upper = (uint64_t(order_f32(bits)) << 32) | ids.back();
            const vqreq::Request* arr[1] = {&page};
            std::vector<std::unique_ptr<Result>> r;
            std::vector<int> st;
            std::vector<std::string> er;
            {
                auto pb = run_partial(idx, arr, 1);
                finish_batch(idx, *pb, nullptr, 1, r, st, er);
            }
            if (st[0] != 0) {
                status[i] = st[0];
                errors[i] = er[0];
                results[i].reset();
                failed = true;
                break;
            }
            if (r[0]->ids.empty()) break;
            ids.insert(ids.end(), r[0]->ids.begin(), r[0]->ids.end());
            scores.insert(scores.end(), r[0]->scores.begin(), r[0]->scores.end());
        }
        if (failed) continue;
        const size_t from = size_t(std::min<uint64_t>(skip, ids.size())), to = size_t(std::min<uint64_t>(want, ids.size()));
        out.ids.assign(ids.begin() + from, ids.begin() + to);
        out.scores.assign(scores.begin() + from, scores.begin() + to);
    }
}

// ---- explain: Explain records of the returned hits.  k_explain recomputes each hit's score through the request's tree and
// writes every value the records quote; what follows here is the reference's bookkeeping of WHICH records a hit collects, in which order.
std::string explain_records_json(const ExplainRecs& records) {  // serde's externally tagged enum (explain.rs:1-21), floats as %.9g (round-trips f32)
    auto f = [](float v) {
        char buf[48];
        std::snprintf(buf, sizeof buf, "%.9g", double(v));
        return std::string(buf);
    };
    std::string out = "[";
    for (size_t i = 0; i < records.size(); ++i) {
        const ExplainRec& e = records[i];
        if (i) out += ",";
        switch (e.kind) {
            case ExplainRec::Boost: out += "{\"Boost\":" + f(e.a) + "}"; break;
            case ExplainRec::MaxTokenToTextId: out += "{\"MaxTokenToTextId\":" + f(e.a) + "}"; break;
            case ExplainRec::OrSumOverDistinctTerms: out += "{\"OrSumOverDistinctTerms\":" + f(e.a) + "}"; break;
            case ExplainRec::TermToAnchor:
                out += "{\"TermToAnchor\":{\"term_score\":" + f(e.a) + ",\"anchor_score\":" + f(e.b) + ",\"final_score\":" + f(e.c) + ",\"term_id\":" + std::to_string(e.term_id) + "}}";
                break;
            case ExplainRec::LevenshteinScore:
                out += "{\"LevenshteinScore\":{\"score\":" + f(e.a) + ",\"text_or_token_id\":";
                vqjson::escape_to(out, e.text);
                out += ",\"term_id\":" + std::to_string(e.term_id) + "}}";
                break;
        }
    }
    return out + "]";
}

namespace {
struct ExplainEval {
    bool has = false;  // the node's explain map holds an entry for the doc
    ExplainRecs recs;
};
inline float trace_f32(uint32_t bits) {
    float v;
    std::memcpy(&v, &bits, 4);
    return v;
}
// The explain map entry of `doc` in the result of `node` — leaf: search_field.rs:419-441 on top of field_result.rs:44; and: set_op.rs:421-433;
// or: set_op.rs:132-137, 187-208.
ExplainEval explain_node(const ExplainPlan& P, int node, uint32_t doc, const uint32_t* T) {
    const ExplainNode& n = P.nodes[size_t(node)];
    ExplainEval out;
    if (n.kind == XP_LEAF) {
        auto stray = n.term_records.find(doc);  // new_from() copied the dictionary result's map: `doc` may be one of its TERM ids
        if (stray != n.term_records.end()) {
            out.has = true;
            out.recs = stray->second;
        }
        for (uint32_t j = 0; j < n.list_count; ++j) {
            const uint32_t* t = T + 3u * (n.list_begin + j);
            if (t[0] == 0xFFFFFFFFu) continue;
            out.has = true;
            ExplainRec e;
            e.kind = ExplainRec::TermToAnchor;
            e.term_id = n.list_term[j];
            e.a = P.lists[n.list_begin + j].term_score;
            e.b = trace_f32(t[1]);
            e.c = trace_f32(t[2]);
            out.recs.push_back(e);
            auto tr = n.term_records.find(n.list_term[j]);
            if (tr != n.term_records.end()) out.recs.insert(out.recs.end(), tr->second.begin(), tr->second.end());
        }
        return out;
    }
    const uint32_t* t_op = T + 3u * uint32_t(P.lists.size()) + 3u * uint32_t(n.op);
    const bool present = t_op[0] != 0;
    if (n.kind == XP_AND) {
        if (!present) return out;  // the result's map only has entries of its hits
        for (size_t k = 0; k + 1 < n.order.size(); ++k) {  // the operands left after the shortest one was taken out, in their new order (:393)
            ExplainEval c = explain_node(P, n.children[n.order[k]], doc, T);
            if (!c.has) continue;
            out.has = true;
            out.recs.insert(out.recs.end(), c.recs.begin(), c.recs.end());
        }
        return out;
    }
    std::vector<ExplainEval> ch;
    for (int c : n.children) ch.push_back(explain_node(P, c, doc, T));
    for (auto& c : ch)
        if (c.has) {  // HashMap::extend (:135): a later operand's entry replaces an earlier one's
            out.has = true;
            out.recs = c.recs;
        }
    if (present) {
        out.has = true;
        ExplainRec e;
        e.kind = ExplainRec::OrSumOverDistinctTerms;
        e.a = trace_f32(t_op[2]);
        out.recs.push_back(e);
        for (auto& c : ch)
            if (c.has) out.recs.insert(out.recs.end(), c.recs.begin(), c.recs.end());
    }
    return out;
}
}  // namespace

void complete_explain_requests(const Index& idx, std::vector<std::unique_ptr<Result>>& results, std::vector<int>& status, std::vector<std::string>& errors) {
    std::vector<ExQuery> queries;
    std::vector<size_t> owner;
    std::vector<uint32_t> doc_query, docs;
    std::vector<ExOp> ops;
    std::vector<uint16_t> aux;
    std::vector<ExList> lists;
    std::vector<DColBoost> cols;
    size_t trace_words = 0;
    for (size_t i = 0; i < results.size(); ++i) {
        if (status[i] != 0 || !results[i] || !results[i]->explain_plan) continue;
        Result& R = *results[i];
        const ExplainPlan& P = *R.explain_plan;
        R.has_explain = true;
        R.explain.assign(R.ids.size(), {});
        if (R.ids.empty()) continue;
        ExQuery q{};
        q.op_begin = uint32_t(ops.size());
        q.n_ops = uint32_t(P.ops.size());
        q.list_begin = uint32_t(lists.size());
        q.n_lists = uint32_t(P.lists.size());
        q.col_begin = uint32_t(cols.size());
        q.n_col = uint32_t(P.cols.size());
        q.doc_begin = uint32_t(docs.size());
        q.trace_begin = uint32_t(trace_words);
        for (ExOp op : P.ops) {
            if (op.kind != XP_LEAF) op.a += uint32_t(aux.size());
            ops.push_back(op);
        }
        aux.insert(aux.end(), P.aux.begin(), P.aux.end());
        lists.insert(lists.end(), P.lists.begin(), P.lists.end());
        cols.insert(cols.end(), P.cols.begin(), P.cols.end());
        docs.insert(docs.end(), R.ids.begin(), R.ids.end());
        doc_query.insert(doc_query.end(), R.ids.size(), uint32_t(queries.size()));
        trace_words += R.ids.size() * size_t(explain_trace_words(q.n_lists, q.n_ops, q.n_col));
        if (trace_words > (1ull << 30)) {
            status[i] = ERR_UNSUPPORTED;
            errors[i] = "unsupported on the MI355X query path: explain trace of more than 4 GiB";
            results[i].reset();
            return;
        }
        queries.push_back(q);
        owner.push_back(i);
    }
    if (docs.empty()) return;
    VQ_HIP(hipSetDevice(idx.device));
    // one upload area: [queries][doc_query][docs][ops][aux][lists][cols], then the trace
    size_t off = 0;
    auto place = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 64);
        return at;
    };
    const size_t o_q = place(queries.size() * sizeof(ExQuery)), o_dq = place(doc_query.size() * 4), o_d = place(docs.size() * 4), o_ops = place(ops.size() * sizeof(ExOp)),
                 o_aux = place(aux.size() * 2), o_l = place(lists.size() * sizeof(ExList)), o_c = place(cols.size() * sizeof(DColBoost));
    std::vector<uint8_t> up(off);
    std::memcpy(up.data() + o_q, queries.data(), queries.size() * sizeof(ExQuery));
    std::memcpy(up.data() + o_dq, doc_query.data(), doc_query.size() * 4);
    std::memcpy(up.data() + o_d, docs.data(), docs.size() * 4);
    std::memcpy(up.data() + o_ops, ops.data(), ops.size() * sizeof(ExOp));
    if (!aux.empty()) std::memcpy(up.data() + o_aux, aux.data(), aux.size() * 2);
    if (!lists.empty()) std::memcpy(up.data() + o_l, lists.data(), lists.size() * sizeof(ExList));
    if (!cols.empty()) std::memcpy(up.data() + o_c, cols.data(), cols.size() * sizeof(DColBoost));
    DevBuf d_up, d_trace;
    d_up.ensure(off);
    d_trace.ensure(trace_words * 4);
    hipStream_t st = idx.stream;
    VQ_HIP(hipMemcpyAsync(d_up.p, up.data(), off, hipMemcpyHostToDevice, st));
    const uint8_t* du = d_up.as<uint8_t>();
    launch_explain(st, uint32_t(docs.size()), reinterpret_cast<const ExQuery*>(du + o_q), reinterpret_cast<const uint32_t*>(du + o_dq), reinterpret_cast<const uint32_t*>(du + o_d),
                   reinterpret_cast<const ExOp*>(du + o_ops), reinterpret_cast<const uint16_t*>(du + o_aux), reinterpret_cast<const ExList*>(du + o_l),
                   reinterpret_cast<const DColBoost*>(du + o_c), d_trace.as<uint32_t>());
    VQ_HIP(hipGetLastError());
    std::vector<uint32_t> trace(trace_words);
    VQ_HIP(hipMemcpyAsync(trace.data(), d_trace.p, trace_words * 4, hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    for (size_t k = 0; k < queries.size(); ++k) {
        const ExQuery& q = queries[k];
        Result& R = *results[owner[k]];
        const ExplainPlan& P = *R.explain_plan;
        const uint32_t words = explain_trace_words(q.n_lists, q.n_ops, q.n_col);
        for (size_t h = 0; h < R.ids.size(); ++h) {
            const uint32_t* T = trace.data() + q.trace_begin + h * size_t(words);
            const uint32_t* t_cols = T + 3u * (q.n_lists + q.n_ops);
            if (t_cols[3u * q.n_col] == 0) {  // a returned hit is a hit of the tree
                status[owner[k]] = ERR_UNSUPPORTED;
                errors[owner[k]] = "internal: explain did not reproduce hit " + std::to_string(R.ids[h]);
                break;
            }
            ExplainEval e = explain_node(P, P.root, R.ids[h], T);
            for (uint32_t c = 0; c < q.n_col; ++c) {  // add_boost (boost.rs:470-504) -> apply_boost's records (:297-300, :371-374)
                if (t_cols[3u * c] == 0) continue;
                e.has = true;
                ExplainRec b;
                b.kind = ExplainRec::Boost;
                if (P.cols[c].fun == BF_LOG10) {
                    b.a = trace_f32(t_cols[3u * c + 1u]);
                    e.recs.push_back(b);
                }
                b.a = trace_f32(t_cols[3u * c + 2u]);
                e.recs.push_back(b);
            }
            R.explain[h] = {e.has, std::move(e.recs)};
        }
        if (status[owner[k]] != 0) results[owner[k]].reset();
    }
}

PartialBatch::~PartialBatch() {
    if (ws && launched && !finished && ws->ev_done) (void)hipEventSynchronize(ws->ev_done);
    release_workspace();
}
// the workspace goes back: a batch that named it takes its mark off FIRST (a workspace that stayed marked would be skipped by every later
// batch — and four of those left an index without workspaces)
void PartialBatch::release_workspace() {
    if (!lock.owns_lock()) return;
    if (ws && pinned_ws) {
        ws->pinned.store(false, std::memory_order_release);
        pinned_ws = false;
    }
    lock.unlock();
}

namespace {
struct DownLayout {  // download area: [hits u64 nq][n u32 nq][ids u32 K][scores f32 K][facet_n u32 J][facet_vals u32 F][facet_counts u32 F]
    size_t K, J, F, o_hits, o_n, o_ids, o_scores, o_fn, o_fv, o_fc, o_stat, bytes;
    explicit DownLayout(const PartialBatch& pb) {
        const uint32_t nq = pb.nq_dev;
        K = size_t(pb.layout.total_keys);
        J = pb.n_facet_jobs;
        F = pb.total_facet_out;
        o_hits = 0;
        o_n = align_up(o_hits + size_t(nq) * 8, 16);
        o_ids = align_up(o_n + size_t(nq) * 4, 16);
        o_scores = align_up(o_ids + K * 4, 16);
        o_fn = align_up(o_scores + K * 4, 16);
        o_fv = align_up(o_fn + J * 4, 16);
        o_fc = align_up(o_fv + F * 4, 16);
        o_stat = align_up(o_fc + F * 4, 16);  // profiling: gathered bytes per query (its own copy, straight into the pinned area)
        bytes = align_up(o_stat + size_t(nq) * 8, 256);
    }
};
}  // namespace

void finish_launch(const Index& idx, PartialBatch& pb, const void* gathered_device, uint32_t num_shards, size_t shard_stride) {
    if (pb.merge_launched) return;
    pb.merge_launched = true;
    Workspace& ws = *pb.ws;
    hipStream_t st = idx.fin_stream;
    const PartialLayout& lay = pb.layout;
    const uint32_t nq = pb.nq_dev;
    VQ_HIP(hipSetDevice(idx.device));
    const DownLayout D(pb);
    const size_t K = D.K, J = D.J;
    if (!nq) return;
    const uint8_t* gathered = gathered_device ? static_cast<const uint8_t*>(gathered_device) : pb.d_partial;
    if (!gathered_device) num_shards = 1;
    ws.d_down.ensure(D.bytes);
    ws.h_down.ensure(D.bytes);
    if (st != idx.stream) VQ_HIP(hipStreamWaitEvent(st, ws.ev_done, 0));
    uint8_t* dd = ws.d_down.as<uint8_t>();
    const bool prof = pb.profiled;
    {
        LaunchTimer t(prof, ws, st, K_FINALIZE, uint64_t(num_shards) * (K * 8 + nq * 8) + K * 8, uint64_t(num_shards) * (K * 8 + nq * 8) + K * 8, nq);
        launch_finalize(st, nq, pb.d_blobs, pb.d_blob_off, gathered, num_shards, shard_stride ? shard_stride : size_t(lay.off_hist), lay, reinterpret_cast<uint32_t*>(dd + D.o_ids),
                        reinterpret_cast<float*>(dd + D.o_scores), reinterpret_cast<uint32_t*>(dd + D.o_n), reinterpret_cast<unsigned long long*>(dd + D.o_hits));
    }
    VQ_HIP(hipGetLastError());
    if (J) {
        // the batch's own histogram area: the caller of the sharded path has summed it over the shards in place (all-reduce)
        const uint32_t* hist = reinterpret_cast<const uint32_t*>(pb.d_partial + lay.off_hist);
        LaunchTimer t(prof, ws, st, K_FACET_SELECT, lay.total_hist * 4, lay.total_hist * 4, J);
        launch_facet_select(st, uint32_t(J), pb.d_facet_jobs, hist, reinterpret_cast<uint32_t*>(dd + D.o_fv), reinterpret_cast<uint32_t*>(dd + D.o_fc),
                            reinterpret_cast<uint32_t*>(dd + D.o_fn));
        VQ_HIP(hipGetLastError());
    }
    VQ_HIP(hipMemcpyAsync(ws.h_down.p, dd, D.o_stat, hipMemcpyDeviceToHost, st));
    if (prof)  // bytes the scans read through per-hit gathers (counted by the kernels), per query — into PINNED memory: a copy into pageable memory
               // would hold this thread until the stream gets there
        VQ_HIP(hipMemcpyAsync(ws.h_down.as<uint8_t>() + D.o_stat, pb.d_partial + lay.off_stats, size_t(nq) * 8, hipMemcpyDeviceToHost, st));
}

void finish_batch(const Index& idx, PartialBatch& pb, const void* gathered_device, uint32_t num_shards, std::vector<std::unique_ptr<Result>>& out,
                  std::vector<int>& status, std::vector<std::string>& errors, size_t shard_stride) {
    const size_t n = pb.queries.size();
    out.clear();
    out.resize(n);
    status.assign(n, 0);
    errors.assign(n, std::string());
    finish_launch(idx, pb, gathered_device, num_shards, shard_stride);
    Workspace& ws = *pb.ws;
    hipStream_t st = idx.fin_stream;
    const PartialLayout& lay = pb.layout;
    const uint32_t nq = pb.nq_dev;
    const DownLayout D(pb);
    const size_t o_hits = D.o_hits, o_n = D.o_n, o_ids = D.o_ids, o_scores = D.o_scores, o_fn = D.o_fn, o_fv = D.o_fv, o_fc = D.o_fc;
    if (nq) {
        const bool prof = pb.profiled;
        VQ_HIP(hipStreamSynchronize(st));
        const uint64_t* gathered_bytes = reinterpret_cast<const uint64_t*>(ws.h_down.as<uint8_t>() + D.o_stat);
        pb.finished = true;
        if (prof) {
            std::lock_guard<std::mutex> g(idx.profile_mutex);
            Profile& P = idx.profile;
            P.batches += 1;
            for (const TimedLaunch& t : ws.timed) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ws.ev_pool[t.ev_begin], ws.ev_pool[t.ev_end]) != hipSuccess) continue;
                KernelProfile& k = P.k[t.kernel];
                k.ms += ms;
                k.launches += 1;
                k.layout_bytes += t.layout_bytes;
                k.algorithmic_bytes += t.algorithmic_bytes;
                k.queries += t.queries;
            }
            for (uint32_t q = 0; q < nq && q < pb.qclass.size(); ++q) {
                P.k[pb.qclass[q]].layout_bytes += gathered_bytes[q];
                P.k[pb.qclass[q]].gathered_bytes += gathered_bytes[q];
            }
            ws.timed.clear();
        }
    }
    const uint64_t ns = uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - pb.t0).count());
    const double t_synced = now_ms();

    const uint8_t* hd = ws.h_down.as<uint8_t>();
    size_t key_off = 0, job = 0, fac_off = 0;
    for (size_t i = 0; i < n; ++i) {
        const CompiledQuery& cq = pb.queries[i];
        if (cq.status != 0) {
            status[i] = cq.status;
            errors[i] = cq.error;
            continue;
        }
        const uint32_t q = pb.slot[i];
        auto r = std::make_unique<Result>();
        r->num_hits = reinterpret_cast<const uint64_t*>(hd + o_hits)[q];
        r->execution_time_ns = ns;
        const uint32_t have = reinterpret_cast<const uint32_t*>(hd + o_n)[q];
        const uint32_t* ids = reinterpret_cast<const uint32_t*>(hd + o_ids) + key_off;
        const float* scores = reinterpret_cast<const float*>(hd + o_scores) + key_off;
        // apply_top_skip (search.rs:230-239) on the top+skip window
        const uint32_t want = cq.top + cq.skip;
        const uint32_t avail = std::min(have, want);
        const uint32_t from = std::min(cq.skip, avail);
        const uint32_t to = std::min(avail, from + cq.top);
        r->ids.assign(ids + from, ids + to);
        r->scores.assign(scores + from, scores + to);
        r->deep = cq.deep;
        if (!std::isnan(cq.or_skip_bound)) {
            // k_scan_probe_or ranked only the docs that hold the OR's cover operand.  Nothing was missed if every hit holds it, or if the last key of
            // the ranked window lies above what a doc WITHOUT the cover can score at best
            const bool confirmed = r->num_hits == have || (have == cq.top_k && cq.top_k > 0 && scores[cq.top_k - 1] > cq.or_skip_bound);
            r->rerun_exact = !confirmed;
        }
        key_off += cq.top_k;
        if (!cq.facet_out.empty()) r->has_facets = true;
        r->why_found_terms = cq.why_found_terms;
        r->explain_plan = cq.explain_plan;
        r->why_found_plan = cq.why_found_plan;
        for (size_t f = 0; f < cq.facet_out.size(); ++f, ++job) {
            const FacetOut& fo = cq.facet_out[f];
            ResultFacet rf;
            rf.field = fo.field;
            const uint32_t fn = reinterpret_cast<const uint32_t*>(hd + o_fn)[job];
            // jobs were appended in query order: this job's output offset is the running sum of the tops
            const auto dit = idx.dict.find(fo.dict_path);
            const size_t out_off = fac_off;
            fac_off += fo.top;
            const uint32_t* fv = reinterpret_cast<const uint32_t*>(hd + o_fv) + out_off;
            const uint32_t* fc = reinterpret_cast<const uint32_t*>(hd + o_fc) + out_off;
            for (uint32_t k = 0; k < fn && k < fo.top; ++k) {
                std::string text = (dit != idx.dict.end() && fv[k] < dit->second.terms.size()) ? dit->second.terms[fv[k]] : std::string();
                rf.entries.push_back({std::move(text), uint64_t(fc[k])});
            }
            if (fo.host_top) {  // more entries than the device ranks: the job's counts (summed over the shards in place), ranked here — count descending,
                                // value id ascending like k_facet_select (the reference's sort is unstable, facet.rs:19-23)
                const FacetJob& fj = pb.facet_jobs[job];
                std::vector<uint32_t> counts(fj.num_values);
                VQ_HIP(hipMemcpyAsync(counts.data(), pb.d_partial + lay.off_hist + size_t(fj.hist_off) * 4, size_t(fj.num_values) * 4, hipMemcpyDeviceToHost, st));
                VQ_HIP(hipStreamSynchronize(st));
                std::vector<uint32_t> order;
                for (uint32_t v = 0; v < fj.num_values; ++v)
                    if (counts[v]) order.push_back(v);
                const size_t keep = std::min<size_t>(order.size(), fo.host_top);
                std::partial_sort(order.begin(), order.begin() + keep, order.end(), [&](uint32_t x, uint32_t y) { return counts[x] != counts[y] ? counts[x] > counts[y] : x < y; });
                for (size_t k = 0; k < keep; ++k) {
                    std::string text = (dit != idx.dict.end() && order[k] < dit->second.terms.size()) ? dit->second.terms[order[k]] : std::string();
                    rf.entries.push_back({std::move(text), uint64_t(counts[order[k]])});
                }
            }
            r->facets.push_back(std::move(rf));
        }
        out[i] = std::move(r);
    }
    pb.release_workspace();
    {  // requests whose speculative route could not be confirmed run again, as a batch of their own, on the exact routes (rare: an OR whose best
       // hits lack its rarest term, or fewer such hits than the request asks for)
        std::vector<size_t> redo;
        for (size_t i = 0; i < n; ++i)
            if (out[i] && out[i]->rerun_exact) redo.push_back(i);
        if (!redo.empty()) {
            std::vector<vqreq::Request> again;
            again.reserve(redo.size());
            for (size_t i : redo) {
                again.push_back(*pb.reqs[i]);
                again.back().exact_routes_only = true;
            }
            std::vector<const vqreq::Request*> arr;
            for (auto& a : again) arr.push_back(&a);
            std::vector<std::unique_ptr<Result>> r2;
            std::vector<int> st2;
            std::vector<std::string> er2;
            {
                auto pb2 = run_partial(idx, arr.data(), arr.size());
                finish_batch(idx, *pb2, nullptr, 1, r2, st2, er2, 0);
            }
            for (size_t k = 0; k < redo.size(); ++k) {
                out[redo[k]] = std::move(r2[k]);
                status[redo[k]] = st2[k];
                errors[redo[k]] = er2[k];
            }
            idx.or_reruns.fetch_add(redo.size(), std::memory_order_relaxed);
        }
    }
    if (timing_enabled()) std::fprintf(stderr, "[vq timing] batch wall %.3f ms, result assembly %.3f ms\n", double(ns) * 1e-6, now_ms() - t_synced);
}

}  // namespace vq
