// Index staging: decoded arrays handed over through the C ABI -> padded segmented arrays in HBM.
// Replaces Persistence::load / load_indices (reference src/persistence.rs:206-291, 393-410) at the
// decoded-array level; the on-disk formats are not read here.
#include <algorithm>
#include <chrono>
#include <cstring>

#include "engine.hpp"
#include "text.hpp"

namespace vq {

const char* const TOKENS_TO_TEXT_ID = ".tokens_to_text_id";
const char* const TO_ANCHOR_ID_SCORE = ".to_anchor_id_score";
const char* const PHRASE_PAIR_TO_ANCHOR = ".phrase_pair_to_anchor";
const char* const VALUE_ID_TO_PARENT = ".value_id_to_parent";
const char* const PARENT_TO_VALUE_ID = ".parent_to_value_id";
const char* const TEXT_ID_TO_ANCHOR = ".text_id_to_anchor";
const char* const ANCHOR_TO_TEXT_ID = ".anchor_to_text_id";
const char* const BOOST_VALID_TO_VALUE = ".boost_valid_to_value";
const char* const TOKEN_VALUES = ".token_values";
const char* const VALUE_ID_TO_ANCHOR = ".value_id_to_anchor";
const char* const TEXTINDEX = ".textindex";

static bool ends_with(const std::string& s, const char* suf) {
    size_t n = std::strlen(suf);
    return s.size() >= n && std::memcmp(s.data() + s.size() - n, suf, n) == 0;
}

void DevBuf::alloc(size_t n) {
    release();
    bytes = n;
    if (n == 0) return;
    VQ_HIP(hipMalloc(&p, n));
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}
void DevBuf::upload(const void* src, size_t n, hipStream_t st) {
    if (n == 0) return;
    if (st) VQ_HIP(hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, st));
    else VQ_HIP(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
}

Index::~Index() {
    for (auto& w : ws) {
        for (auto e : w.ev_pool)
            if (e) (void)hipEventDestroy(e);
        if (w.ev_done) (void)hipEventDestroy(w.ev_done);
    }
    if (own_stream) (void)hipStreamDestroy(own_stream);
    if (own_fin_stream) (void)hipStreamDestroy(own_fin_stream);
    if (pre_stream) (void)hipStreamDestroy(pre_stream);
}

bool Index::is_anchor_identity(const std::string& textindex_path) const {
    // reference util.rs:131-137 extract_field_name: drop the trailing ".textindex"
    std::string field = textindex_path;
    if (ends_with(field, TEXTINDEX)) field.resize(field.size() - std::strlen(TEXTINDEX));
    auto it = columns.find(field);
    return it != columns.end() && it->second.is_anchor_identity_column;
}

// IEEE binary16 round-to-nearest-even of an f32 (== half::f16::from_f32, reference
// src/indices/persistence_score/token_to_anchor_score_vint.rs:155)
static uint16_t f32_to_f16_rne(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t exp = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (exp == 0xFF) return uint16_t(sign | 0x7C00u | (man ? (0x200u | (man >> 13)) : 0));
    const int32_t e = int32_t(exp) - 112;
    if (e >= 0x1F) return uint16_t(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return uint16_t(sign);
        man |= 0x800000u;
        const uint32_t shift = uint32_t(14 - e);
        uint32_t hm = man >> shift;
        const uint32_t rb = 1u << (shift - 1);
        if ((man & rb) && ((man & (rb - 1)) || (hm & 1))) hm++;
        return uint16_t(sign | hm);
    }
    uint32_t h = sign | (uint32_t(e) << 10) | (man >> 13);
    if ((man & 0x1000u) && ((man & 0xFFFu) || (h & 1))) h++;
    return uint16_t(h);
}

// pad a row to a multiple of 4 entries with the 0xFFFFFFFF sentinel
static void append_padded(std::vector<uint32_t>& dst, const uint32_t* b, const uint32_t* e) {
    dst.insert(dst.end(), b, e);
    while (dst.size() & 3u) dst.push_back(0xFFFFFFFFu);
}

static void shard_subrange(const uint32_t* b, const uint32_t* e, uint32_t lo, uint32_t hi, const uint32_t** ob, const uint32_t** oe) {
    *ob = std::lower_bound(b, e, lo);
    *oe = std::lower_bound(*ob, e, hi);
}

std::unique_ptr<Index> build_index(const IndexBuilder& b, int device) {
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev <= 0)
        throw VelociError(vqreq::ERR_DEVICE, std::string("no HIP device available (hipGetDeviceCount: ") + hipGetErrorString(de) + ", count " +
                                                 std::to_string(ndev) + "): the veloci_amd query path needs an MI355X (no CPU fallback)");
    if (device < 0 || device >= ndev) throw VelociError(vqreq::ERR_INVALID_ARGUMENT, "device index out of range");
    VQ_HIP(hipSetDevice(device));

    auto idx = std::make_unique<Index>();
    idx->device = device;
    idx->num_anchors = b.num_anchors;
    idx->doc_lo = b.doc_lo;
    idx->doc_hi = b.doc_hi;
    idx->columns = b.columns;
    VQ_HIP(hipStreamCreateWithFlags(&idx->own_stream, hipStreamNonBlocking));
    {  // merges and downloads are short and somebody waits for them: ahead of the scans of the next batch wherever a wave slot frees up
        int lo = 0, hi = 0;
        VQ_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        VQ_HIP(hipStreamCreateWithPriority(&idx->own_fin_stream, hipStreamNonBlocking, hi));
    }
    VQ_HIP(hipStreamCreateWithFlags(&idx->pre_stream, hipStreamNonBlocking));
    idx->stream = idx->own_stream;
    idx->fin_stream = idx->own_fin_stream;
    for (auto& w : idx->ws) {
        VQ_HIP(hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming));
    }
    const uint32_t lo = b.doc_lo, hi = b.doc_hi;
    idx->bitmap_base = lo & ~65535u;
    // one bitmap covers [bitmap_base, hi) rounded up to 65536 docs, plus one tile of slack for the last tile's copy
    idx->bitmap_words = ((uint64_t(hi) - idx->bitmap_base + 65535u) / 65536u) * 2048u + 2048u;

    for (auto& [path, f] : b.fst) {
        Dictionary d;
        d.terms = f.terms;
        d.lower_map.reserve(d.terms.size());
        for (uint32_t i = 0; i < d.terms.size(); ++i) d.lower_map[vqtext::to_lower_utf8(d.terms[i])].push_back(i);
        {  // device image for k_dict_scan: code points as u16 (raw + lower-cased), CSR over the terms
            std::vector<uint32_t> off(d.terms.size() + 1, 0u);
            std::vector<uint16_t> raw, low;
            for (uint32_t i = 0; i < d.terms.size(); ++i) {
                for (uint32_t cp : vqtext::decode_utf8(d.terms[i])) {
                    if (cp > 0xFFFFu) d.bmp_only = false;
                    if (cp == 0x130u) d.low_exact = false;  // lower-cases to TWO code points ("i" + U+0307): the image is per code point
                    raw.push_back(uint16_t(cp));
                    low.push_back(uint16_t(vqtext::lower_cp(cp)));
                }
                off[i + 1] = uint32_t(raw.size());
            }
            if (d.bmp_only) {
                d.d_off.alloc(off.size() * 4 + 16);
                d.d_off.upload(off.data(), off.size() * 4);
                d.d_raw.alloc(raw.size() * 2 + 16);
                d.d_raw.upload(raw.data(), raw.size() * 2);
                d.d_low.alloc(low.size() * 2 + 16);
                d.d_low.upload(low.data(), low.size() * 2);
                idx->device_bytes += d.d_off.bytes + d.d_raw.bytes + d.d_low.bytes;
            }
        }
        idx->dict.emplace(path, std::move(d));
    }

    for (auto& [path, p] : b.postings) {
        PostingStore ps;
        ps.num_tokens = uint32_t(p.offsets.size() - 1);
        ps.start.resize(ps.num_tokens);
        ps.len.resize(ps.num_tokens);
        ps.global_len.resize(ps.num_tokens);
        ps.max_raw.assign(ps.num_tokens, 0);
        std::vector<uint32_t> docs;
        std::vector<uint16_t> scores;
        docs.reserve(p.anchors.size() + 4 * size_t(std::min<uint32_t>(ps.num_tokens, 1u << 20)));
        for (uint32_t t = 0; t < ps.num_tokens; ++t) {
            const uint32_t* rb = p.anchors.data() + p.offsets[t];
            const uint32_t* re = p.anchors.data() + p.offsets[t + 1];
            const uint32_t *sb, *se;
            shard_subrange(rb, re, lo, hi, &sb, &se);
            ps.start[t] = docs.size();
            ps.len[t] = uint32_t(se - sb);
            ps.global_len[t] = p.global_lens.empty() ? uint64_t(re - rb) : p.global_lens[t];
            append_padded(docs, sb, se);
            const uint32_t* sc = p.scores.data() + (sb - p.anchors.data());
            for (uint32_t i = 0; i < ps.len[t]; ++i) {
                const uint16_t h = f32_to_f16_rne(float(sc[i]));  // non-negative: bit order == value order
                scores.push_back(h);
                ps.max_raw[t] = std::max(ps.max_raw[t], h);
            }
            while (scores.size() < docs.size()) scores.push_back(0);
        }
        ps.total_padded = docs.size();
        // ---- bitmap images of the dense lists (reading a tile of a dense list becomes a straight 16 B/lane copy)
        ps.bm_start.assign(ps.num_tokens, -1);
        ps.rd_start.assign(ps.num_tokens, -1);
        {
            const uint64_t range = uint64_t(hi) - lo;
            const uint64_t words = idx->bitmap_words, blocks = words >> (kRankShift - 5);  // rank directory: one entry per 512 docs (kRankShift)
            std::vector<uint32_t> dense;
            for (uint32_t t = 0; t < ps.num_tokens; ++t)
                if (range >= 65536 && uint64_t(ps.len[t]) * 64 >= range) dense.push_back(t);
            if (!dense.empty()) {
                std::vector<uint32_t> bits(words * dense.size(), 0u);
                std::vector<uint32_t> rdir((blocks + 1) * dense.size(), 0u);
                for (size_t k = 0; k < dense.size(); ++k) {
                    const uint32_t t = dense[k];
                    uint32_t* bw = bits.data() + words * k;
                    uint32_t* rd = rdir.data() + (blocks + 1) * k;
                    const uint32_t* d = docs.data() + ps.start[t];
                    for (uint32_t i = 0; i < ps.len[t]; ++i) {
                        const uint32_t rel = d[i] - idx->bitmap_base;
                        bw[rel >> 5] |= 1u << (rel & 31u);
                        rd[(rel >> kRankShift) + 1] += 1;
                    }
                    for (uint64_t bl = 1; bl <= blocks; ++bl) rd[bl] += rd[bl - 1];
                    ps.bm_start[t] = int64_t(words * k);
                    ps.rd_start[t] = int64_t((blocks + 1) * k);
                }
                ps.bitmaps.alloc(bits.size() * 4 + 16);
                ps.bitmaps.upload(bits.data(), bits.size() * 4);
                ps.rank_dir.alloc(rdir.size() * 4 + 16);
                ps.rank_dir.upload(rdir.data(), rdir.size() * 4);
                idx->device_bytes += ps.bitmaps.bytes + ps.rank_dir.bytes;
            }
        }
        // ---- tile directories of the lists that hold at least 1/4096 of the shard's docs (dense ones included: one can be the cover of an AND)
        ps.td_start.assign(ps.num_tokens, -1);
        {
            const uint64_t range = uint64_t(hi) - lo;
            const uint64_t tiles = (idx->bitmap_words >> (kTileDirShift - 5)) + 3;  // (k_scan_probe reads every second entry, up to one behind its last 32768-doc tile)
            std::vector<uint32_t> tdir;
            for (uint32_t t = 0; t < ps.num_tokens; ++t) {
                if (range < 65536 || uint64_t(ps.len[t]) * 4096 < range) continue;
                ps.td_start[t] = int64_t(tdir.size());
                const uint32_t* d = docs.data() + ps.start[t];
                uint32_t i = 0;
                for (uint64_t k = 0; k <= tiles; ++k) {
                    const uint64_t bound = uint64_t(idx->bitmap_base) + (k << kTileDirShift);
                    while (i < ps.len[t] && d[i] < bound) ++i;
                    tdir.push_back(i);
                }
            }
            if (!tdir.empty()) {
                ps.tile_dir.alloc(tdir.size() * 4 + 16);
                ps.tile_dir.upload(tdir.data(), tdir.size() * 4);
                idx->device_bytes += ps.tile_dir.bytes;
            }
        }
        // ---- tile-packed images of the same lists (k_scan_probe: cover stream and 16-bit array operands); sized in a first pass, then
        // built and uploaded list by list (the host never holds a second copy of the whole store)
        ps.pk_start.assign(ps.num_tokens, -1);
        ps.ak_start.assign(ps.num_tokens, -1);
        ps.gd_start.assign(ps.num_tokens, -1);
        {
            const uint64_t ptiles = (idx->bitmap_words >> (kProbeTileShift - 5)) + 2;  // (entry `one behind the last tile` included)
            uint64_t cov_gran = 0, arr_gran = 0;
            std::vector<uint32_t> gd;
            for (uint32_t t = 0; t < ps.num_tokens; ++t) {
                if (ps.td_start[t] < 0) continue;
                const uint32_t* d = docs.data() + ps.start[t];
                uint64_t gran = 0;
                uint32_t i = 0, most = 0;
                ps.gd_start[t] = int64_t(gd.size());
                for (uint64_t k = 0; k <= ptiles; ++k) {
                    gd.push_back(uint32_t(gran));
                    const uint64_t thi = uint64_t(idx->bitmap_base) + ((k + 1) << kProbeTileShift);
                    const uint32_t i0 = i;
                    while (i < ps.len[t] && d[i] < thi) ++i;
                    const uint32_t g = (i - i0 + 7u) / 8u;
                    most = std::max(most, g);
                    gran += g;
                }
                if (gran >= (1ull << 28)) throw VelociError(vqreq::ERR_INVALID_ARGUMENT, "posting list too long for its tile-packed image");
                ps.pk_start[t] = int64_t(cov_gran);
                cov_gran += gran;
                if (most <= 256) {  // no tile holds more than 2048 entries: the list can be an array operand
                    ps.ak_start[t] = int64_t(arr_gran);
                    arr_gran += gran;
                }
            }
            if (cov_gran) {
                ps.cov32.alloc(cov_gran * 32 + 4096);  // (slack: the kernel guards its loads per 16-byte vector)
                ps.arr16.alloc(arr_gran * 16 + 4096);
                ps.gdir.alloc(gd.size() * 4 + 16);
                ps.gdir.upload(gd.data(), gd.size() * 4);
                idx->device_bytes += ps.cov32.bytes + ps.arr16.bytes + ps.gdir.bytes;
                std::vector<uint32_t> cov;
                std::vector<uint16_t> arr;
                for (uint32_t t = 0; t < ps.num_tokens; ++t) {
                    if (ps.pk_start[t] < 0) continue;
                    const uint32_t* d = docs.data() + ps.start[t];
                    const uint16_t* sc = scores.data() + ps.start[t];
                    cov.clear();
                    arr.clear();
                    uint32_t i = 0;
                    for (uint64_t k = 0; k <= ptiles; ++k) {
                        const uint64_t tlo = uint64_t(idx->bitmap_base) + (k << kProbeTileShift), thi = tlo + (1u << kProbeTileShift);
                        for (; i < ps.len[t] && d[i] < thi; ++i) {
                            const uint32_t rel = uint32_t(d[i] - tlo);
                            cov.push_back(rel << 16 | sc[i]);
                            arr.push_back(uint16_t(rel));
                        }
                        while (cov.size() % 8) {
                            cov.push_back(0xFFFFFFFFu);
                            arr.push_back(0xFFFFu);
                        }
                    }
                    if (cov.empty()) continue;
                    VQ_HIP(hipMemcpy(ps.cov32.as<uint8_t>() + uint64_t(ps.pk_start[t]) * 32, cov.data(), cov.size() * 4, hipMemcpyHostToDevice));
                    if (ps.ak_start[t] >= 0) VQ_HIP(hipMemcpy(ps.arr16.as<uint8_t>() + uint64_t(ps.ak_start[t]) * 16, arr.data(), arr.size() * 2, hipMemcpyHostToDevice));
                }
            }
        }
        ps.docs.alloc(docs.size() * 4 + 16);
        ps.docs.upload(docs.data(), docs.size() * 4);
        ps.scores.alloc(scores.size() * 2 + 16);
        ps.scores.upload(scores.data(), scores.size() * 2);
        idx->device_bytes += ps.docs.bytes + ps.scores.bytes;
        idx->postings.emplace(path, std::move(ps));
    }

    for (auto& [path, k] : b.kv) {
        KVStore s;
        s.key_base = k.key_base;
        s.num_keys = uint32_t(k.offsets.size() - 1);
        s.host_off = k.offsets;
        s.host_values = k.values;
        // which stores hold anchors as VALUES (rows usable as doc-id lists on the device)?
        bool identity_t2t = false;
        if (ends_with(path, TOKENS_TO_TEXT_ID)) {
            std::string ti = path.substr(0, path.size() - std::strlen(TOKENS_TO_TEXT_ID));
            identity_t2t = idx->is_anchor_identity(ti);
        }
        s.list_rows = ends_with(path, TEXT_ID_TO_ANCHOR) || identity_t2t;
        if (identity_t2t) {  // text id == anchor: row t must then be the doc list of posting list t; verify instead of assuming
            std::string pp = path.substr(0, path.size() - std::strlen(TOKENS_TO_TEXT_ID)) + TO_ANCHOR_ID_SCORE;
            auto bit = b.postings.find(pp);
            if (bit != b.postings.end() && bit->second.offsets.size() == k.offsets.size() && k.key_base == 0) {
                const auto& po = bit->second;
                bool same = po.offsets == k.offsets && po.anchors.size() == k.values.size() &&
                            std::memcmp(po.anchors.data(), k.values.data(), k.values.size() * 4) == 0;
                s.rows_equal_postings = same;
            }
        }
        if (s.list_rows) {
            s.start.resize(s.num_keys);
            s.len.resize(s.num_keys);
            std::vector<uint32_t> vals;
            vals.reserve(k.values.size() + 4);
            std::vector<uint32_t> tmp;
            for (uint32_t r = 0; r < s.num_keys; ++r) {
                const uint32_t* rb = k.values.data() + k.offsets[r];
                const uint32_t* re = k.values.data() + k.offsets[r + 1];
                if (!std::is_sorted(rb, re) || std::adjacent_find(rb, re) != re) {
                    // the device row is a sorted set; multiplicities, where they matter, come from the host copy
                    tmp.assign(rb, re);
                    std::sort(tmp.begin(), tmp.end());
                    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
                    rb = tmp.data();
                    re = tmp.data() + tmp.size();
                    s.rows_sorted_unique = false;
                }
                const uint32_t *sb, *se;
                shard_subrange(rb, re, lo, hi, &sb, &se);
                s.start[r] = vals.size();
                s.len[r] = uint32_t(se - sb);
                append_padded(vals, sb, se);
            }
            s.values.alloc(vals.size() * 4 + 16);
            s.values.upload(vals.data(), vals.size() * 4);
            idx->device_bytes += s.values.bytes;
            if (ends_with(path, TEXT_ID_TO_ANCHOR)) {  // row table on the device: the text-locality pre-pass (K7) expands text ids to anchors there
                s.d_row_start.alloc(s.start.size() * 8 + 16);
                s.d_row_start.upload(s.start.data(), s.start.size() * 8);
                s.d_row_len.alloc(s.len.size() * 4 + 16);
                s.d_row_len.upload(s.len.data(), s.len.size() * 4);
                idx->device_bytes += s.d_row_start.bytes + s.d_row_len.bytes;
            }
        }
        if (ends_with(path, TOKENS_TO_TEXT_ID) && !identity_t2t) {  // text ids are not anchors: the table itself goes to HBM for K7
            s.text_csr = true;
            s.d_text_vals.alloc(k.values.size() * 4 + 16);
            s.d_text_vals.upload(k.values.data(), k.values.size() * 4);
            idx->device_bytes += s.d_text_vals.bytes;
        }
        if (ends_with(path, VALUE_ID_TO_PARENT) || ends_with(path, VALUE_ID_TO_ANCHOR)) {  // the 1:n boost pre-pass (K10) walks these on the device
            s.value_csr = true;
            s.d_text_vals.alloc(k.values.size() * 4 + 16);
            s.d_text_vals.upload(k.values.data(), k.values.size() * 4);
            s.d_csr_off.alloc(k.offsets.size() * 8 + 16);
            s.d_csr_off.upload(k.offsets.data(), k.offsets.size() * 8);
            idx->device_bytes += s.d_text_vals.bytes + s.d_csr_off.bytes;
        }
        // facet sources: anchor-keyed value lists (facet.rs:38-44)
        s.facet_csr = ends_with(path, ANCHOR_TO_TEXT_ID) || (ends_with(path, PARENT_TO_VALUE_ID) && path.find("[]") == std::string::npos);
        if (s.facet_csr) {
            const uint64_t kb = k.key_base, ke = uint64_t(k.key_base) + s.num_keys;
            const uint64_t nb = std::min<uint64_t>(std::max<uint64_t>(kb, lo), ke);
            const uint64_t ne = std::max<uint64_t>(nb, std::min<uint64_t>(ke, hi));
            std::vector<uint64_t> off(ne - nb + 1);
            const uint64_t first = k.offsets[nb - kb];
            for (uint64_t r = nb; r <= ne; ++r) off[r - nb] = k.offsets[r - kb] - first;
            s.csr_off.alloc(off.size() * 8);
            s.csr_off.upload(off.data(), off.size() * 8);
            const uint64_t nvals = k.offsets[ne - kb] - first;
            s.csr_values.alloc(nvals * 4 + 16);
            s.csr_values.upload(k.values.data() + first, nvals * 4);
            idx->device_bytes += s.csr_off.bytes + s.csr_values.bytes;
            s.csr_key_base = uint32_t(nb);
            s.csr_num_keys = uint32_t(ne - nb);
            for (uint64_t e = first; e < first + nvals; ++e) s.csr_max_value = std::max(s.csr_max_value, k.values[e]);
            bool single = true;  // scalar fields: at most one value per anchor -> a direct column
            for (size_t r = 0; r + 1 < off.size() && single; ++r) single = off[r + 1] - off[r] <= 1;
            if (single && s.csr_num_keys) {
                std::vector<uint32_t> direct(s.csr_num_keys, 0xFFFFFFFFu);
                for (size_t r = 0; r + 1 < off.size(); ++r)
                    if (off[r + 1] > off[r]) direct[r] = k.values[first + off[r]];
                s.csr_direct.alloc(direct.size() * 4 + 16);
                s.csr_direct.upload(direct.data(), direct.size() * 4);
                idx->device_bytes += s.csr_direct.bytes;
            }
        }
        idx->kv.emplace(path, std::move(s));
    }

    for (auto& [path, p] : b.phrase) {
        PhraseStore s;
        const size_t n = p.t1.size();
        s.keys.reserve(n);
        s.start.resize(n);
        s.len.resize(n);
        std::vector<uint32_t> vals;
        for (size_t i = 0; i < n; ++i) {
            s.keys.push_back({p.t1[i], p.t2[i]});
            const uint32_t* rb = p.anchors.data() + p.offsets[i];
            const uint32_t* re = p.anchors.data() + p.offsets[i + 1];
            const uint32_t *sb, *se;
            shard_subrange(rb, re, lo, hi, &sb, &se);
            s.start[i] = vals.size();
            s.len[i] = uint32_t(se - sb);
            append_padded(vals, sb, se);
        }
        if (!std::is_sorted(s.keys.begin(), s.keys.end())) throw VelociError(vqreq::ERR_INVALID_ARGUMENT, "phrase pair keys must be sorted by (t1, t2): " + path);
        s.anchors.alloc(vals.size() * 4 + 16);
        s.anchors.upload(vals.data(), vals.size() * 4);
        idx->device_bytes += s.anchors.bytes;
        idx->phrase.emplace(path, std::move(s));
    }

    for (auto& [path, bc] : b.boost) {
        BoostColumn c;
        c.key_base = bc.key_base;
        c.num_keys = uint32_t(bc.bits.size());
        for (size_t i = 0; i < bc.bits.size(); ++i) {  // the range of the present values: what bounds a boost factor (exact top-k pruning of boosted requests)
            if (!bc.present.empty() && !bc.present[i]) continue;
            float v;
            std::memcpy(&v, &bc.bits[i], 4);
            if (v != v) {
                c.has_nan = true;
                continue;
            }
            c.vmin = c.any_value ? std::min(c.vmin, v) : v;
            c.vmax = c.any_value ? std::max(c.vmax, v) : v;
            c.any_value = true;
        }
        c.values.alloc(bc.bits.size() * 4 + 16);
        c.values.upload(bc.bits.data(), bc.bits.size() * 4);
        idx->device_bytes += c.values.bytes;
        // host copies: 1:n boost columns (keys are value ids) and token_values columns (keys are term ids) are resolved on the host
        if (path.find("[]") != std::string::npos || path.find(".token_values") != std::string::npos) {
            c.host_bits = bc.bits;
            c.host_present.assign(bc.present.begin(), bc.present.end());
        }
        if (!bc.present.empty()) {
            c.has_present = true;
            std::vector<uint32_t> bits((bc.bits.size() + 31) / 32, 0u);
            for (size_t i = 0; i < bc.present.size(); ++i)
                if (bc.present[i]) bits[i >> 5] |= 1u << (i & 31);
            c.present.alloc(bits.size() * 4 + 16);
            c.present.upload(bits.data(), bits.size() * 4);
            idx->device_bytes += c.present.bytes;
        }
        idx->boost.emplace(path, std::move(c));
    }
    VQ_HIP(hipDeviceSynchronize());
    return idx;
}

// join_anchor_to_leaf (facet.rs:75-93) for every anchor of the shard: anchor -> value ids of step 0 -> ... -> text ids of the last step
const KVStore& Index::composed_facet(const std::vector<std::string>& steps) const {
    std::string key;
    for (auto& s : steps) key += s + "|";
    std::lock_guard<std::mutex> g(composed_mu);
    auto it = composed_facets.find(key);
    if (it != composed_facets.end()) return *it->second;
    std::vector<const KVStore*> chain;
    for (auto& s : steps) {
        auto kit = kv.find(s + PARENT_TO_VALUE_ID);
        if (kit == kv.end()) throw VelociError(vqreq::ERR_INDEX_NOT_FOUND, "Did not found path in indices " + s + PARENT_TO_VALUE_ID);
        chain.push_back(&kit->second);
    }
    VQ_HIP(hipSetDevice(device));
    auto out = std::make_unique<KVStore>();
    std::vector<uint64_t> off(size_t(doc_hi - doc_lo) + 1, 0);
    std::vector<uint32_t> vals, level, next;
    for (uint32_t a = doc_lo; a < doc_hi; ++a) {
        level.assign(1, a);
        for (const KVStore* st : chain) {
            next.clear();
            for (uint32_t id : level) {
                const uint32_t *rb, *re;
                if (st->host_row(id, &rb, &re)) next.insert(next.end(), rb, re);
            }
            level.swap(next);
        }
        vals.insert(vals.end(), level.begin(), level.end());
        off[a - doc_lo + 1] = vals.size();
    }
    out->facet_csr = true;
    out->csr_key_base = doc_lo;
    out->csr_num_keys = doc_hi - doc_lo;
    for (uint32_t v : vals) out->csr_max_value = std::max(out->csr_max_value, v);
    out->csr_off.alloc(off.size() * 8);
    out->csr_off.upload(off.data(), off.size() * 8);
    out->csr_values.alloc(vals.size() * 4 + 16);
    out->csr_values.upload(vals.data(), vals.size() * 4);
    VQ_HIP(hipDeviceSynchronize());
    const KVStore& ref = *out;
    composed_facets.emplace(key, std::move(out));
    return ref;
}

}  // namespace vq
