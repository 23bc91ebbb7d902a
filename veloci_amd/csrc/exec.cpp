// Batch executor: compile n requests, pack their device programs, launch the kernels, assemble results.
// One batch == one k_tile_scan launch over all (query, span) pairs (SURVEY.md §7 "design for batches").
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "engine.hpp"

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

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// serialise one compiled query into `dst` (host), whose device address will be `dev`
static size_t pack_blob(const CompiledQuery& cq, const Index& idx, uint8_t* dst, const uint8_t* dev, uint32_t keys_base, uint32_t part_keys_off,
                        const std::vector<uint32_t>& hist_off, const std::vector<uint32_t>& fac_out_off, size_t* desc_bytes_out = nullptr) {
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
    h.n_col = uint32_t(cq.cols.size());
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
    h.n_temps = cq.n_temps;
    h.simple_n = cq.simple_n;
    h.bitmap_base = idx.bitmap_base;
    h.simple_flags = cq.simple_flags;
    h.desc_bytes = uint32_t(off);
    if (desc_bytes_out) *desc_bytes_out = off;
    std::vector<size_t> inline_off(cq.inline_lists.size());
    for (size_t i = 0; i < cq.inline_lists.size(); ++i) inline_off[i] = section(align_up(cq.inline_lists[i].size(), 4) * 4);
    h.top_k = cq.top_k;
    h.tile_words = cq.tile_words;
    h.n_spans = cq.n_spans;
    h.keys_base = keys_base;
    h.doc_lo = idx.doc_lo;
    h.doc_hi = idx.doc_hi;
    h.part_keys_off = part_keys_off;
    h.blob_bytes = uint32_t(off);
    if (!dst) return off;

    std::memcpy(dst, &h, sizeof h);
    DList* dl = reinterpret_cast<DList*>(dst + h.off_lists);
    for (size_t i = 0; i < cq.lists.size(); ++i) {
        const HList& l = cq.lists[i];
        DList d{};
        d.docs = l.inline_idx >= 0 ? reinterpret_cast<const uint32_t*>(dev + inline_off[l.inline_idx]) : l.d_docs;
        d.scores = l.d_scores;
        d.len = l.len;
        d.flags = l.flags;
        d.term_score = l.term_score;
        d.bitmap = l.d_bitmap;
        d.rank_dir = l.d_rank_dir;
        dl[i] = d;
    }
    if (!cq.ops.empty()) std::memcpy(dst + h.off_ops, cq.ops.data(), cq.ops.size() * sizeof(DOp));
    if (!cq.fops.empty()) std::memcpy(dst + h.off_fops, cq.fops.data(), cq.fops.size() * sizeof(DOp));
    if (!cq.groups.empty()) std::memcpy(dst + h.off_groups, cq.groups.data(), cq.groups.size() * sizeof(DGroup));
    if (!cq.tboosts.empty()) std::memcpy(dst + h.off_tboost, cq.tboosts.data(), cq.tboosts.size() * sizeof(DTermBoost));
    if (!cq.cols.empty()) std::memcpy(dst + h.off_col, cq.cols.data(), cq.cols.size() * sizeof(DColBoost));
    if (!cq.locf.empty()) std::memcpy(dst + h.off_locf, cq.locf.data(), cq.locf.size() * sizeof(DLocField));
    if (!cq.pres.empty()) std::memcpy(dst + h.off_pres, cq.pres.data(), cq.pres.size() * sizeof(DPresOp));
    if (!cq.pres_in.empty()) std::memcpy(dst + h.off_pres_in, cq.pres_in.data(), cq.pres_in.size() * sizeof(uint16_t));
    DFacet* df = reinterpret_cast<DFacet*>(dst + h.off_facets);
    for (size_t i = 0; i < cq.facets.size(); ++i) {
        DFacet f = cq.facets[i];
        f.hist_off = hist_off[i];
        f.out_off = fac_out_off[i];
        df[i] = f;
    }
    for (size_t i = 0; i < cq.inline_lists.size(); ++i) {
        uint32_t* p = reinterpret_cast<uint32_t*>(dst + inline_off[i]);
        const auto& v = cq.inline_lists[i];
        std::memcpy(p, v.data(), v.size() * 4);
        for (size_t k = v.size(); k < align_up(v.size(), 4); ++k) p[k] = 0xFFFFFFFFu;
    }
    return off;
}

static bool timing_enabled() {
    static const bool on = std::getenv("VQ_TIMING") != nullptr;
    return on;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::unique_ptr<PartialBatch> run_partial(const Index& idx, const vqreq::Request* const* reqs, size_t n, int slot) {
    const double t_start = now_ms();
    auto pb = std::make_unique<PartialBatch>();
    pb->index = &idx;
    pb->t0 = std::chrono::steady_clock::now();
    if (slot < 0) slot = int(idx.next_ws.fetch_add(1) % kWorkspaces);
    pb->ws = &idx.ws[slot % kWorkspaces];
    pb->lock = std::unique_lock<std::mutex>(pb->ws->mu);
    VQ_HIP(hipSetDevice(idx.device));
    Workspace& ws = *pb->ws;
    hipStream_t st = idx.stream;

    // ---- compile
    pb->queries.reserve(n);
    pb->slot.assign(n, UINT32_MAX);
    pb->queries.resize(n);
    auto compile_range = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            if (!reqs[i]) {
                pb->queries[i].status = ERR_INVALID_ARGUMENT;
                pb->queries[i].error = "null request";
            } else pb->queries[i] = compile_query(idx, *reqs[i]);
        }
    };
    if (n >= 256) {  // query compilation is independent per request: fan out over a few host threads
        const size_t nt = 4;
        std::vector<std::thread> th;
        for (size_t t = 1; t < nt; ++t) th.emplace_back(compile_range, n * t / nt, n * (t + 1) / nt);
        compile_range(0, n / nt);
        for (auto& t : th) t.join();
    } else compile_range(0, n);
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
    uint64_t algo_bytes = 0;
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
        blob_bytes += pack_blob(cq, idx, nullptr, nullptr, 0, 0, {}, {});
        max_lists = std::max<uint32_t>(max_lists, uint32_t(cq.lists.size()));
        max_ww = std::max(max_ww, cq.tile_words);
        stack_depth = std::max(stack_depth, cq.stack_depth);
        algo_bytes += cq.algorithmic_bytes;
    }
    if (total_span_keys > 0xFFFFFFFFull || total_hist > 0xFFFFFFFFull || total_spans > 0x7FFFFFFFull)
        throw VelociError(ERR_UNSUPPORTED, "batch too large for 32-bit workspace offsets: split the batch");
    span_base.push_back(uint32_t(total_spans));
    pb->nq_dev = nq;
    pb->total_spans = uint32_t(total_spans);
    pb->n_facet_jobs = uint32_t(jobs.size());
    pb->total_facet_out = fac_out_total;

    PartialLayout& lay = pb->layout;
    lay.nq = nq;
    lay.total_keys = total_keys;
    lay.total_hist = total_hist;
    lay.off_hits = 0;
    lay.off_keys = align_up(size_t(nq) * 8, 16);
    lay.off_hist = lay.off_keys + align_up(size_t(total_keys) * 8, 16);
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
    const size_t up_span_d = up_qmap_s + tbl;   // simple + dense survivors (OR / single-term): k_scan_simple<true>
    const size_t up_qmap_d = up_span_d + tbl;
    const size_t up_jobs = up_qmap_d + tbl;
    const size_t up_bytes = up_jobs + align_up(jobs.size() * sizeof(FacetJob), 256) + 256;
    ws.h_up.ensure(up_bytes);
    ws.d_up.ensure(up_bytes);
    uint8_t* hup = ws.h_up.as<uint8_t>();
    uint8_t* dup = ws.d_up.as<uint8_t>();
    uint32_t n_simple = 0, n_generic = 0, n_dense = 0, spans_simple = 0, spans_generic = 0, spans_dense = 0;
    {
        size_t off = 0;
        uint32_t* hbo = reinterpret_cast<uint32_t*>(hup + up_blob_off);
        uint32_t qi = 0;
        for (size_t i = 0; i < n; ++i) {
            CompiledQuery& cq = pb->queries[i];
            if (cq.status != 0) continue;
            hbo[qi] = uint32_t(off);
            off += pack_blob(cq, idx, hup + off, dup + off, keys_base[qi], part_keys_off[qi], hist_offs[qi], fac_out_offs[qi]);
            ++qi;
        }
        hbo[nq] = uint32_t(off);
        uint32_t* sg = reinterpret_cast<uint32_t*>(hup + up_span_base);
        uint32_t* mg = reinterpret_cast<uint32_t*>(hup + up_qmap_g);
        uint32_t* ss = reinterpret_cast<uint32_t*>(hup + up_span_s);
        uint32_t* ms = reinterpret_cast<uint32_t*>(hup + up_qmap_s);
        uint32_t* sd = reinterpret_cast<uint32_t*>(hup + up_span_d);
        uint32_t* md = reinterpret_cast<uint32_t*>(hup + up_qmap_d);
        uint32_t accg = 0, accs = 0, accd = 0;
        qi = 0;
        for (size_t i = 0; i < n; ++i) {
            const CompiledQuery& cq = pb->queries[i];
            if (cq.status != 0) continue;
            // the LDS-staged dense-tile variant is opt-in (VQ_DENSE=1): at its current occupancy it is slower than the queue path
            static const bool dense_enabled = std::getenv("VQ_DENSE") != nullptr;
            const bool dense = dense_enabled && cq.simple_flags && (cq.simple_n == 1 || cq.ops.back().kind == OP_OR);
            if (dense) {
                sd[n_dense] = accd;
                md[n_dense++] = qi;
                accd += cq.n_spans;
            } else if (cq.simple_flags) {
                ss[n_simple] = accs;
                ms[n_simple++] = qi;
                accs += cq.n_spans;
            } else {
                sg[n_generic] = accg;
                mg[n_generic++] = qi;
                accg += cq.n_spans;
            }
            ++qi;
        }
        sg[n_generic] = accg;
        ss[n_simple] = accs;
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

    VQ_HIP(hipMemcpyAsync(dup, hup, up_bytes, hipMemcpyHostToDevice, st));
    ws.d_span_keys.ensure(size_t(total_span_keys) * 8 + 16);
    ws.d_partial.ensure(lay.bytes);
    pb->d_partial = ws.d_partial.as<uint8_t>();
    VQ_HIP(hipMemsetAsync(pb->d_partial, 0, lay.bytes, st));

    // ---- the scan
    uint32_t max_top_k = 1;
    for (size_t i = 0; i < n; ++i)
        if (pb->queries[i].status == 0) max_top_k = std::max(max_top_k, pb->queries[i].top_k);
    uint32_t desc_cap = 0;  // bytes of the largest query descriptor (staged into LDS by every workgroup)
    for (size_t i = 0; i < n; ++i)
        if (pb->queries[i].status == 0) {
            size_t d = 0;
            pack_blob(pb->queries[i], idx, nullptr, nullptr, 0, 0, {}, {}, &d);
            desc_cap = std::max(desc_cap, uint32_t(d));
        }
    desc_cap = uint32_t(align_up(desc_cap, 16));
    uint32_t cand_cap = 256;  // power of two >= 2 * top_k: candidate keys a workgroup keeps in LDS
    while (cand_cap < 2 * max_top_k) cand_cap <<= 1;
    for (size_t i = 0; i < n; ++i)
        if (pb->queries[i].status == 0 && !pb->queries[i].simple_flags)
            lds_bytes = std::max(lds_bytes, tile_scan_lds_bytes(uint32_t(pb->queries[i].lists.size()) + pb->queries[i].n_temps, uint32_t(pb->queries[i].lists.size()), pb->queries[i].tile_words, stack_depth, cand_cap, desc_cap));
    if (lds_bytes > 160 * 1024) throw VelociError(ERR_UNSUPPORTED, "LDS tile larger than 160 KiB");
    pb->profiled = idx.profile.enabled;
    if (pb->profiled) VQ_HIP(hipEventRecord(ws.ev0, st));
    launch_scan_simple(st, false, spans_simple, pb->d_blobs, pb->d_blob_off, reinterpret_cast<const uint32_t*>(dup + up_span_s),
                       reinterpret_cast<const uint32_t*>(dup + up_qmap_s), n_simple, cand_cap, ws.d_span_keys.as<unsigned long long>(),
                       reinterpret_cast<unsigned long long*>(pb->d_partial + lay.off_hits));
    VQ_HIP(hipGetLastError());
    launch_scan_simple(st, true, spans_dense, pb->d_blobs, pb->d_blob_off, reinterpret_cast<const uint32_t*>(dup + up_span_d),
                       reinterpret_cast<const uint32_t*>(dup + up_qmap_d), n_dense, cand_cap, ws.d_span_keys.as<unsigned long long>(),
                       reinterpret_cast<unsigned long long*>(pb->d_partial + lay.off_hits));
    VQ_HIP(hipGetLastError());
    launch_tile_scan(st, spans_generic, lds_bytes, pb->d_blobs, pb->d_blob_off, pb->d_span_base, reinterpret_cast<const uint32_t*>(dup + up_qmap_g), n_generic,
                     stack_depth, cand_cap, desc_cap, ws.d_span_keys.as<unsigned long long>(),
                     reinterpret_cast<unsigned long long*>(pb->d_partial + lay.off_hits), reinterpret_cast<uint32_t*>(pb->d_partial + lay.off_hist));
    if (pb->profiled) {
        VQ_HIP(hipEventRecord(ws.ev1, st));
        std::lock_guard<std::mutex> g(idx.profile_mutex);
        idx.profile.scan_launches += 1;
        idx.profile.algorithmic_bytes += algo_bytes;
    }
    VQ_HIP(hipGetLastError());
    launch_merge_spans(st, nq, pb->d_blobs, pb->d_blob_off, ws.d_span_keys.as<unsigned long long>(),
                       reinterpret_cast<unsigned long long*>(pb->d_partial + lay.off_keys));
    VQ_HIP(hipGetLastError());
    VQ_HIP(hipEventRecord(ws.ev_done, st));
    if (timing_enabled())
        std::fprintf(stderr, "[vq timing] n=%zu compile %.3f ms, pack+launch %.3f ms\n", n, t_compiled - t_start, now_ms() - t_compiled);
    return pb;
}

void finish_batch(const Index& idx, PartialBatch& pb, const void* gathered_device, uint32_t num_shards, std::vector<std::unique_ptr<Result>>& out,
                  std::vector<int>& status, std::vector<std::string>& errors) {
    const size_t n = pb.queries.size();
    out.clear();
    out.resize(n);
    status.assign(n, 0);
    errors.assign(n, std::string());
    Workspace& ws = *pb.ws;
    hipStream_t st = idx.fin_stream;
    const PartialLayout& lay = pb.layout;
    const uint32_t nq = pb.nq_dev;
    VQ_HIP(hipSetDevice(idx.device));

    // download area: [hits u64 nq][n u32 nq][ids u32 K][scores f32 K][facet_n u32 J][facet_vals u32 F][facet_counts u32 F]
    const size_t K = size_t(lay.total_keys), J = pb.n_facet_jobs, F = pb.total_facet_out;
    const size_t o_hits = 0;
    const size_t o_n = align_up(o_hits + size_t(nq) * 8, 16);
    const size_t o_ids = align_up(o_n + size_t(nq) * 4, 16);
    const size_t o_scores = align_up(o_ids + K * 4, 16);
    const size_t o_fn = align_up(o_scores + K * 4, 16);
    const size_t o_fv = align_up(o_fn + J * 4, 16);
    const size_t o_fc = align_up(o_fv + F * 4, 16);
    const size_t down_bytes = align_up(o_fc + F * 4, 256);

    if (nq) {
        const uint8_t* gathered = gathered_device ? static_cast<const uint8_t*>(gathered_device) : pb.d_partial;
        if (!gathered_device) num_shards = 1;
        ws.d_down.ensure(down_bytes);
        ws.h_down.ensure(down_bytes);
        if (st != idx.stream) VQ_HIP(hipStreamWaitEvent(st, ws.ev_done, 0));
        uint8_t* dd = ws.d_down.as<uint8_t>();
        launch_finalize(st, nq, pb.d_blobs, pb.d_blob_off, gathered, num_shards, lay, reinterpret_cast<uint32_t*>(dd + o_ids),
                        reinterpret_cast<float*>(dd + o_scores), reinterpret_cast<uint32_t*>(dd + o_n), reinterpret_cast<unsigned long long*>(dd + o_hits));
        VQ_HIP(hipGetLastError());
        if (J) {
            const uint32_t* hist;
            if (num_shards > 1) {
                ws.d_hist_sum.ensure(size_t(lay.total_hist) * 4 + 16);
                launch_hist_reduce(st, gathered, num_shards, lay, ws.d_hist_sum.as<uint32_t>());
                hist = ws.d_hist_sum.as<uint32_t>();
            } else hist = reinterpret_cast<const uint32_t*>(gathered + lay.off_hist);
            launch_facet_select(st, uint32_t(J), pb.d_facet_jobs, hist, reinterpret_cast<uint32_t*>(dd + o_fv), reinterpret_cast<uint32_t*>(dd + o_fc),
                                reinterpret_cast<uint32_t*>(dd + o_fn));
            VQ_HIP(hipGetLastError());
        }
        VQ_HIP(hipMemcpyAsync(ws.h_down.p, dd, down_bytes, hipMemcpyDeviceToHost, st));
        VQ_HIP(hipStreamSynchronize(st));
        if (pb.profiled) {
            float ms = 0.f;
            std::lock_guard<std::mutex> g(idx.profile_mutex);
            if (hipEventElapsedTime(&ms, ws.ev0, ws.ev1) == hipSuccess) idx.profile.scan_ms += ms;
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
        key_off += cq.top_k;
        if (!cq.facet_out.empty()) r->has_facets = true;
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
            r->facets.push_back(std::move(rf));
        }
        out[i] = std::move(r);
    }
    pb.lock.unlock();
    if (timing_enabled()) std::fprintf(stderr, "[vq timing] batch wall %.3f ms, result assembly %.3f ms\n", double(ns) * 1e-6, now_ms() - t_synced);
}

}  // namespace vq
