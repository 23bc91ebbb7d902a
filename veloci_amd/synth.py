"""Deterministic synthetic corpora of the shapes BASELINE.json names.

The generator emits the *decoded index* directly (no text), following the reference's list-shape rules:
sorted unique anchors per posting list (src/create.rs:389-411), integer scores from
`calculate_token_score_for_entry` (src/create/calculate_score.rs:34-49), phrase-pair anchor lists
sorted and de-duplicated (src/create.rs:505-517), one text per doc with text_id == anchor
(`is_anchor_identity_column`).

Membership of doc d in a probe list is a pure function of (seed, term key, d) — a 64-bit mix compared
against the list's density — so any doc range [doc_lo, doc_hi) can be generated on its own and the
shards of a sharded index are exactly the slices of the unsharded one.  The heavy per-doc hashing
runs in torch (on the GPU when there is one); it is bench/test input generation, not the query path.
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from .index import IndexData

SEED = 0x5EEDC0DE2024
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_GOLD = 0x9E3779B97F4A7C15


def _s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix_t(x):
    """splitmix64 finalizer on a torch int64 tensor (two's-complement wraparound)."""
    x = (x ^ _lsr(x, 30)) * _s64(_M1)
    x = (x ^ _lsr(x, 27)) * _s64(_M2)
    return x ^ _lsr(x, 31)


def _h32(key, docs):
    """32 uniform bits per doc for stream `key` (torch int64 in, int64 in [0, 2^32) out)."""
    return _lsr(_mix_t(docs * _s64(_GOLD) + _s64(_mix_int(key))), 32)


def _mix_int(x):
    m = (1 << 64) - 1
    x &= m
    x = ((x ^ (x >> 30)) * _M1) & m
    x = ((x ^ (x >> 27)) * _M2) & m
    return x ^ (x >> 31)


def token_score_table(df):
    """calculate_token_score_for_entry(pos 0..15, num_occurences=df, num_tokens 4..16, false) + 1
    (src/create/calculate_score.rs:34-49; +min(dups,5) with one occurrence, src/create.rs:400-403)."""
    pos = np.arange(16, dtype=np.float32)[:, None]
    ntok = np.arange(4, 17, dtype=np.float32)[None, :]
    score = np.float32(2000.0) / (np.log2(pos + np.float32(10.0)) + np.float32(10.0))
    m = np.log10(np.float32(df) + np.float32(1000.0)) - np.float32(2.0)
    m = np.float32(m - (m - np.float32(1.0)) * np.float32(0.7))
    score = (score / m).astype(np.float32)
    t = np.log10(ntok + np.float32(10.0)).astype(np.float32)
    t = (t - (t - np.float32(1.0)) * np.float32(0.7)).astype(np.float32)
    return ((score / t).astype(np.float32)).astype(np.uint32) + np.uint32(1)


@dataclass
class SynthSpec:
    num_docs: int
    num_terms: int = 100_000           # dictionary size T of field `body`
    triples: int = 1                   # probe triples (a, b, c) with df fractions `fractions`
    fractions: tuple = (0.1, 0.03, 0.01)
    overlap: float = 0.1               # |A ∩ B ∩ C| ~= overlap * min df (planted)
    extra_probe_dfs: tuple = ()        # extra single probe terms with these absolute dfs (config #2)
    background_terms: int = 0          # Zipf(1.0) background lists, top df = 0.1 N
    phrase_fraction: float = 0.3       # share of A∩B (and B∩C) stored as phrase-pair anchors
    with_phrase: bool = True
    with_t2t: bool = True              # tokens_to_text_id (== posting docs, identity column)
    with_facets: bool = True           # `cat` (C=1024, one value/doc), `tags[]` (C=65536, 1-3 values/doc)
    with_boost: bool = True            # `pop` f32 in [1, 1e5)
    cat_values: int = 1024
    tag_values: int = 65536
    seed: int = SEED
    chunk: int = 1 << 24


@dataclass
class SynthMeta:
    triples: list = field(default_factory=list)        # [(term_a, term_b, term_c)]
    extra_probes: list = field(default_factory=list)   # [term]
    background: list = field(default_factory=list)     # [term]
    local_lens: dict = field(default_factory=dict)     # path -> u64[T] local posting lengths


def make_vocabulary(num_terms, seed):
    """ASCII lowercase terms, length 4..12, unique, bytewise sorted (ordinal == term id)."""
    rng = np.random.default_rng(seed & 0xFFFFFFFF)
    out = set()
    while len(out) < num_terms:
        need = int((num_terms - len(out)) * 1.05) + 16
        lens = rng.integers(4, 13, need)
        raw = rng.integers(97, 123, (need, 12), dtype=np.uint8)
        for i in range(need):
            out.add(bytes(raw[i, :lens[i]]))
            if len(out) >= num_terms:
                break
    return sorted(out)


def _bernoulli_list(dev, lo, hi, chunk, key, thr32, shared=None):
    """docs d in [lo, hi) with h32(key, d) < thr32, or (shared) h32(shared_key, d) < shared_thr32; plus score streams."""
    docs_out = []
    for c0 in range(lo, hi, chunk):
        c1 = min(hi, c0 + chunk)
        d = torch.arange(c0, c1, dtype=torch.int64, device=dev)
        m = _h32(key, d) < thr32
        if shared is not None:
            m |= _h32(shared[0], d) < shared[1]
        docs_out.append(d[m])
    docs = torch.cat(docs_out) if docs_out else torch.zeros(0, dtype=torch.int64, device=dev)
    pos = _h32(key ^ 0xA5A5A5A5, docs) % 16
    ntok = _h32(key ^ 0x5A5A5A5A, docs) % 13
    return docs.to(torch.int32).cpu().numpy().view(np.uint32), (pos * 13 + ntok).to(torch.int16).cpu().numpy().astype(np.int64)


def _zipf_cdf(n):
    w = 1.0 / np.arange(1, n + 1, dtype=np.float64)
    c = np.cumsum(w)
    return c / c[-1]


def generate(spec: SynthSpec, doc_lo=0, doc_hi=None, device=None):
    """Returns (IndexData restricted to [doc_lo, doc_hi), SynthMeta)."""
    N = spec.num_docs
    lo, hi = int(doc_lo), int(N if doc_hi is None else doc_hi)
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    data = IndexData(N)
    meta = SynthMeta()
    terms = make_vocabulary(spec.num_terms, spec.seed)
    T = len(terms)
    data.add_fst("body.textindex", terms)
    data.set_column_meta("body", True, True)

    lists = {}  # term id -> (docs u32 sorted, scores u32)
    n_probe = 3 * spec.triples + len(spec.extra_probe_dfs)
    if n_probe + spec.background_terms > T:
        raise ValueError("dictionary too small for the requested probe/background terms")
    # probe term ids: evenly spread over the dictionary
    slots = [int((i + 0.5) * T / (n_probe + 1)) for i in range(n_probe)]
    used = set(slots)

    def add_list(tid, docs, stream, df_nominal):
        table = token_score_table(max(int(df_nominal), 1)).reshape(-1)
        lists[tid] = (docs, table[stream].astype(np.uint32))

    pairs = {}  # (t1, t2) -> anchors
    si = 0
    for j in range(spec.triples):
        tids = slots[si:si + 3]
        si += 3
        fr = spec.fractions
        q = spec.overlap * min(fr)                 # density of the planted common docs
        shared = (spec.seed ^ (0x7111 + j), int(q * (1 << 32)))
        tri_docs = []
        for x in range(3):
            p = (fr[x] - q) / (1.0 - q)
            docs, stream = _bernoulli_list(dev, lo, hi, spec.chunk, spec.seed ^ (tids[x] * 0x9E37 + 0x1234567), int(p * (1 << 32)), shared)
            add_list(tids[x], docs, stream, fr[x] * N)
            tri_docs.append(docs)
        meta.triples.append(tuple(terms[t].decode() for t in tids))
        if spec.with_phrase:
            for (x, y) in ((0, 1), (1, 2)):
                both = np.intersect1d(tri_docs[x], tri_docs[y], assume_unique=True)
                keep = _h32(spec.seed ^ (0xFACE + 2 * j + x), torch.from_numpy(both.astype(np.int64)).to(dev)) < int(spec.phrase_fraction * (1 << 32))
                pairs[(tids[x], tids[y])] = both[keep.cpu().numpy()]
    for k, df in enumerate(spec.extra_probe_dfs):
        tid = slots[si]
        si += 1
        p = min(float(df) / N, 1.0)
        thr = (1 << 32) if p >= 1.0 else int(p * (1 << 32))
        docs, stream = _bernoulli_list(dev, lo, hi, spec.chunk, spec.seed ^ (tid * 0x9E37 + 0x7654321), thr)
        add_list(tid, docs, stream, df)
        meta.extra_probes.append(terms[tid].decode())
    # Zipf background: rank r has df = 0.1 N / r (sampled without replacement over [0, N), then sliced)
    rng = np.random.default_rng((spec.seed >> 8) & 0xFFFFFFFF)
    bg_ids = []
    cand = 0
    while len(bg_ids) < spec.background_terms:
        tid = int((cand * 2654435761) % T)
        cand += 1
        if tid not in used:
            used.add(tid)
            bg_ids.append(tid)
    for r, tid in enumerate(bg_ids, start=1):
        df = max(int(0.1 * N / r), 1)
        docs = np.unique(rng.integers(0, N, int(df * 1.02) + 8, dtype=np.int64))[:df].astype(np.uint32)
        docs = docs[(docs >= lo) & (docs < hi)]
        stream = (_h32(spec.seed ^ (tid * 0x9E37 + 0x2468ACE), torch.from_numpy(docs.astype(np.int64))) % (16 * 13)).numpy()
        add_list(tid, docs, stream, df)
        meta.background.append(terms[tid].decode())

    # ---- CSR over all T terms
    lens = np.zeros(T, np.uint64)
    for tid, (docs, _) in lists.items():
        lens[tid] = len(docs)
    offsets = np.zeros(T + 1, np.uint64)
    offsets[1:] = np.cumsum(lens)
    total = int(offsets[-1])
    anchors = np.zeros(total, np.uint32)
    scores = np.zeros(total, np.uint32)
    for tid in sorted(lists):  # (every list is let go of as soon as it is copied: the generator never holds the postings twice)
        docs, sc = lists.pop(tid)
        o = int(offsets[tid])
        anchors[o:o + len(docs)] = docs
        scores[o:o + len(docs)] = sc
        del docs, sc
    data.add_token_to_anchor_score("body.textindex.to_anchor_id_score", offsets, anchors, scores, None)
    meta.local_lens["body.textindex.to_anchor_id_score"] = lens
    if spec.with_t2t:
        data.add_key_value_store("body.textindex.tokens_to_text_id", offsets, anchors)
    if spec.with_phrase and pairs:
        keys = sorted(pairs)
        po = np.zeros(len(keys) + 1, np.uint64)
        po[1:] = np.cumsum([len(pairs[k]) for k in keys])
        pa = np.concatenate([pairs[k] for k in keys]).astype(np.uint32) if int(po[-1]) else np.zeros(0, np.uint32)
        data.add_phrase_pair_to_anchor("body.textindex.phrase_pair_to_anchor", [k[0] for k in keys], [k[1] for k in keys], po, pa)

    # ---- per-doc stores over [lo, hi)
    if spec.with_facets or spec.with_boost:
        d = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    if spec.with_facets:
        C1, C2 = spec.cat_values, spec.tag_values
        data.add_fst("cat.textindex", ["cat%05d" % i for i in range(C1)])
        data.add_fst("tags[].textindex", ["tag%06d" % i for i in range(C2)])
        u = (_h32(spec.seed ^ 0xCA7, d).to(torch.float64) / float(1 << 32)).cpu().numpy()
        cat = np.searchsorted(_zipf_cdf(C1), u).astype(np.uint32).clip(0, C1 - 1)
        data.add_key_value_store("cat.textindex.parent_to_value_id", np.arange(hi - lo + 1, dtype=np.uint64), cat, key_base=lo)
        ntags = (1 + (_h32(spec.seed ^ 0x7A65, d) % 3)).cpu().numpy().astype(np.int64)
        cdf2 = _zipf_cdf(C2)
        cols = []
        for k in range(3):
            uk = (_h32(spec.seed ^ (0x7A60 + 16 * (k + 1)), d).to(torch.float64) / float(1 << 32)).cpu().numpy()
            cols.append(np.searchsorted(cdf2, uk).astype(np.uint32).clip(0, C2 - 1))
        tag_mat = np.stack(cols, axis=1)
        mask = np.arange(3)[None, :] < ntags[:, None]
        to = np.zeros(hi - lo + 1, np.uint64)
        to[1:] = np.cumsum(ntags)
        data.add_key_value_store("tags[].textindex.anchor_to_text_id", to, tag_mat[mask], key_base=lo)
    if spec.with_boost:
        u = (_h32(spec.seed ^ 0xB005, d).to(torch.float64) / float(1 << 32)).cpu().numpy()
        pop = (1.0 + u * (1e5 - 1.0)).astype(np.float32)
        data.add_boost("pop.boost_valid_to_value", pop, None, key_base=lo)
    return data, meta


# ------------------------------------------------------------------------------------------ requests
def req_single(term, top=10, **extra):
    r = {"search_req": {"search": {"path": "body", "terms": [term]}}, "top": top}
    r.update(extra)
    return r


def req_and(terms, top=10, **extra):
    r = {"search_req": {"and": {"queries": [{"search": {"path": "body", "terms": [t]}} for t in terms]}}, "top": top}
    r.update(extra)
    return r


def req_or(terms, top=10, **extra):
    r = {"search_req": {"or": {"queries": [{"search": {"path": "body", "terms": [t]}} for t in terms]}}, "top": top}
    r.update(extra)
    return r


def req_and_phrase_locality(terms, top=10):
    """config #3: 3-term AND + 2 phrase pairs (a,b),(b,c) + text locality."""
    r = req_and(terms, top)
    r["phrase_boosts"] = [{"search1": {"path": "body", "terms": [terms[i]]}, "search2": {"path": "body", "terms": [terms[i + 1]]}} for i in range(len(terms) - 1)]
    r["text_locality"] = True
    return r


def req_and_of_ors(t1, t2, top=10):
    """config #5 third shape: AND(OR, OR) + Log10 `pop` boost + one phrase pair + locality."""
    sub = lambda ts: {"or": {"queries": [{"search": {"path": "body", "terms": [t]}} for t in ts]}}
    return {"search_req": {"and": {"queries": [sub(t1), sub(t2)]}}, "top": top,
            "boost": [{"path": "pop", "boost_fun": "Log10", "param": 1.0}],
            "phrase_boosts": [{"search1": {"path": "body", "terms": [t1[0]]}, "search2": {"path": "body", "terms": [t1[1]]}}],
            "text_locality": True}
