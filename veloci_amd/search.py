"""Host-side mirror of the reference's search surface (src/search.rs:143, src/search/request/mod.rs,
src/search/result/search_result.rs) over the C ABI: `search(request, index) -> SearchResult`."""
import ctypes as C
import json

import numpy as np

from . import _lib
from ._lib import VelociError  # noqa: F401  (re-exported)


class Hit:
    """search::Hit (src/search.rs:53-57)."""
    __slots__ = ("id", "score")

    def __init__(self, id, score):
        self.id = int(id)
        self.score = float(score)

    def __repr__(self):
        return f"Hit(id={self.id}, score={self.score!r})"

    def __eq__(self, o):
        return isinstance(o, Hit) and self.id == o.id and self.score == o.score


class SearchResult:
    """search::SearchResult (src/search/result/search_result.rs:9-26)."""

    def __init__(self, num_hits, ids, scores, facets, execution_time_ns):
        self.num_hits = int(num_hits)
        self.ids = ids              # np.uint32
        self.scores = scores        # np.float32
        self.facets = facets        # None or {field: [(value, count)]}
        self.execution_time_ns = int(execution_time_ns)
        self.is_page = False

    @property
    def data(self):
        return [Hit(i, s) for i, s in zip(self.ids.tolist(), self.scores.tolist())]

    def __repr__(self):
        return f"SearchResult(num_hits={self.num_hits}, data={self.data[:3]}..., facets={self.facets})"


class Request:
    """A parsed search::Request (`vq_request`).  Accepts the reference's JSON (str/bytes) or a dict."""

    def __init__(self, request):
        if isinstance(request, dict):
            request = json.dumps(request)
        if isinstance(request, str):
            request = request.encode()
        self.L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self.L.vq_request_parse(request, len(request), C.byref(h)))
        self.h = h
        self.has_facets = bool(self.L.vq_request_has_facets(self.h))  # from the PARSED request: decides the exchange path of a sharded step on every rank alike

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vq_request_free(self.h)
            self.h = None

    def page_after(self, score, doc_id):
        """The continuation of this request behind the ranked hit (score, doc_id) — `vq_request_page_after`."""
        h = C.c_void_p()
        _lib.check(self.L.vq_request_page_after(self.h, C.c_float(float(score)), int(doc_id), C.byref(h)))
        page = Request.__new__(Request)
        page.L, page.h, page.has_facets = self.L, h, False  # (a continuation page carries no facets: page 0 counted them)
        return page

    def top_skip(self):
        """(top, skip) as parsed (top: 10 when absent, search.rs:146)."""
        d = json.loads(self.L.vq_request_to_json(self.h).decode())
        return (10 if d.get("top") is None else int(d["top"])), int(d.get("skip") or 0)


def complete_deep_pages(requests, results, run_page):
    """Requests whose top + skip reaches beyond one scan's ranking come back from the sharded merge as their page 0 (`res.is_page`): page on
    — `run_page(list of Request) -> list of SearchResult` is one more partial -> exchange -> merge round over all shards — and apply skip / top
    (search.rs:230-239).  The host loop of `complete_deep_requests` (exec.cpp) for callers that own the exchange."""
    max_deep, page_len = 65536, 1024
    state = {}
    for i, res in enumerate(results):
        if isinstance(res, SearchResult) and res.is_page:
            req = _as_request(requests[i])
            top, skip = req.top_skip()
            want = top + skip
            res.is_page = False
            ids, scores = res.ids, res.scores
            if skip >= res.num_hits:
                res.ids, res.scores = np.zeros(0, np.uint32), np.zeros(0, np.float32)
                continue
            reach = min(want, res.num_hits)
            if reach > max_deep:
                results[i] = VelociError(4, f"unsupported on the MI355X query path: top + skip reaches more than {max_deep} ranked hits")
                continue
            state[i] = [req, [ids], [scores], len(ids), reach, want, skip]
    pending = sorted(i for i, s in state.items() if s[3] < s[4] and s[3] % page_len == 0 and s[3] > 0)
    while pending:
        pages = [state[i][0].page_after(state[i][2][-1][-1], state[i][1][-1][-1]) for i in pending]
        got = run_page(pages)
        nxt = []
        for i, g in zip(pending, got):
            s = state[i]
            if isinstance(g, VelociError):
                results[i] = g
                state.pop(i)
                continue
            if len(g.ids) == 0:
                continue
            s[1].append(g.ids)
            s[2].append(g.scores)
            s[3] += len(g.ids)
            if s[3] < s[4] and s[3] % page_len == 0:
                nxt.append(i)
        pending = nxt
    for i, s in state.items():
        ids, scores = np.concatenate(s[1]), np.concatenate(s[2])
        results[i].ids, results[i].scores = ids[s[6]:s[5]], scores[s[6]:s[5]]
    return results


def _take_result(L, h):
    try:
        n = L.vq_result_len(h)
        ids = np.ctypeslib.as_array(L.vq_result_ids(h), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        scores = np.ctypeslib.as_array(L.vq_result_scores(h), shape=(n,)).copy() if n else np.zeros(0, np.float32)
        nf = L.vq_result_num_facets(h)
        facets = None
        if nf:
            facets = {}
            for f in range(nf):
                fl = L.vq_result_facet_len(h, f)
                facets[L.vq_result_facet_field(h, f).decode()] = [(L.vq_result_facet_value(h, f, i).decode(), int(L.vq_result_facet_count(h, f, i)))
                                                                   for i in range(fl)]
        res = SearchResult(L.vq_result_num_hits(h), ids, scores, facets, L.vq_result_execution_time_ns(h))
        res.is_page = bool(L.vq_result_is_page(h))  # page 0 of a request that reaches beyond one scan's ranking (sharded merge only)
        wf = L.vq_result_why_found_terms_json(h)  # (requests without why_found / explain: two short strings, no JSON parsing per result)
        res.why_found_terms = {} if wf == b"{}" else json.loads(wf.decode())
        wi = L.vq_result_why_found_info_json(h)  # why_found with select (search.rs:220-224): {anchor id: {field: [highlighted texts]}}; None when not asked
        res.why_found_info = None if wi == b"null" else {int(k): v for k, v in json.loads(wi.decode()).items()}
        ex = L.vq_result_explain_json(h)  # per hit: null or its Explain records (src/search.rs:86,96); "null" without `explain`
        res.explain_json = ex.decode()
        res.explain = None if ex == b"null" else json.loads(res.explain_json)
        return res
    finally:
        L.vq_result_free(h)


def _as_request(r):
    return r if isinstance(r, Request) else Request(r)


def search(request, index):
    """== veloci::search::search(request, &persistence) (src/search.rs:143-228)."""
    L = _lib.lib()
    req = _as_request(request)
    out = C.c_void_p()
    _lib.check(L.vq_search(index.h, req.h, C.byref(out)))
    return _take_result(L, out)


def suggest(request, index):
    """== search_field::suggest_multi / suggest (src/search/search_field.rs:194-231): `request` is a Request with "suggest" parts or a bare
    RequestSearchPart (dict or JSON).  -> [(text, score, term_id)]"""
    L = _lib.lib()
    if isinstance(request, dict):
        request = json.dumps(request)
    if isinstance(request, str):
        request = request.encode()
    out = C.c_void_p()
    _lib.check(L.vq_suggest_json(index.h, request, len(request), C.byref(out)))
    try:
        return [(L.vq_suggest_text(out, i).decode(), float(L.vq_suggest_score(out, i)), int(L.vq_suggest_term_id(out, i))) for i in range(L.vq_suggest_len(out))]
    finally:
        L.vq_suggest_free(out)


def highlight(part, index):
    """== search_field::highlight (src/search/search_field.rs:233-245): `part` is a RequestSearchPart (dict or JSON) with "snippet": true.
    -> [(snippet, score, text_id)]"""
    L = _lib.lib()
    if isinstance(part, dict):
        part = json.dumps(part)
    if isinstance(part, str):
        part = part.encode()
    out = C.c_void_p()
    _lib.check(L.vq_highlight_json(index.h, part, len(part), C.byref(out)))
    try:
        return [(L.vq_suggest_text(out, i).decode(), float(L.vq_suggest_score(out, i)), int(L.vq_suggest_term_id(out, i))) for i in range(L.vq_suggest_len(out))]
    finally:
        L.vq_suggest_free(out)


def highlight_text(text, terms, snippet_info=None, tokenized=True):
    """== highlight_field::highlight_text (src/highlight_field.rs:92-146) — `vq_highlight_text`.  -> the snippet, or None."""
    L = _lib.lib()
    raw = text.encode()
    enc = [t.encode() for t in terms]
    arr = (C.c_char_p * len(enc))(*enc)
    lens = (C.c_size_t * len(enc))(*[len(e) for e in enc])
    si = json.dumps(snippet_info).encode() if snippet_info is not None else None
    cap = 2 * len(raw) + 256
    while True:
        buf = C.create_string_buffer(cap)
        n = L.vq_highlight_text(raw, len(raw), arr, lens, len(enc), si, len(si) if si else 0, int(tokenized), buf, cap)
        if n == C.c_size_t(-1).value:
            return None
        if n == C.c_size_t(-2).value:
            msg = L.vq_last_error().decode("utf-8", "replace")
            raise _lib.VelociError(1 if msg.startswith("InvalidRequest") else 7 if msg.startswith("JsonError") else 6, msg)
        if n <= cap:
            return buf.raw[:n].decode()
        cap = n


def search_batch(requests, index, raise_on_error=True):
    """n independent searches executed as one device batch (`vq_search_batch`)."""
    L = _lib.lib()
    reqs = [_as_request(r) for r in requests]
    n = len(reqs)
    arr = (C.c_void_p * n)(*[r.h for r in reqs])
    outs = (C.c_void_p * n)()
    status = (C.c_int * n)()
    _lib.check(L.vq_search_batch(index.h, arr, n, outs, status))
    results = []
    for i in range(n):
        if status[i] != 0:
            if raise_on_error:
                for j in range(i + 1, n):
                    if outs[j]:
                        L.vq_result_free(C.c_void_p(outs[j]))
                raise VelociError(status[i], L.vq_last_error().decode("utf-8", "replace"))
            results.append(VelociError(status[i], _lib.ERR_NAMES.get(status[i], "error")))
        else:
            results.append(_take_result(L, C.c_void_p(outs[i])))
    return results


class RequestBatch:
    """n parsed requests as one C array: build once, search many times (the throughput path)."""

    def __init__(self, requests):
        self.reqs = [_as_request(r) for r in requests]
        self.n = len(self.reqs)
        self.arr = (C.c_void_p * self.n)(*[r.h for r in self.reqs])
        self.has_facets = any(getattr(r, "has_facets", True) for r in self.reqs)
        self._splits = {}

    def split(self, k):
        """k contiguous sub-batches (cached): used to pipeline a large batch."""
        if k <= 1:
            return [self]
        if k not in self._splits:
            self._splits[k] = [RequestBatch(self.reqs[self.n * c // k:self.n * (c + 1) // k]) for c in range(k)]
        return self._splits[k]


def search_batch_flat(batch, index, stride=10):
    """`vq_search_batch_flat`: returns (num_hits u64[n], counts u32[n], ids u32[n, stride], scores f32[n, stride], status i32[n])."""
    L = _lib.lib()
    if not isinstance(batch, RequestBatch):
        batch = RequestBatch(batch)
    n = batch.n
    num_hits = np.zeros(n, np.uint64)
    counts = np.zeros(n, np.uint32)
    ids = np.zeros((n, stride), np.uint32)
    scores = np.zeros((n, stride), np.float32)
    status = np.zeros(n, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(L.vq_search_batch_flat(index.h, batch.arr, n, stride, p(num_hits), p(counts), p(ids), p(scores), p(status)))
    return num_hits, counts, ids, scores, status


class PartialBatch:
    """Shard-local partial results of a batch, resident in HBM (`vq_partial_batch`)."""

    def __init__(self, index, requests, slot=None, arena_offset=None):
        """slot / arena_offset: a chunk of a sharded step with one collective (vq_search_batch_partial_at) — its partial is placed at
        `arena_offset` of the index's partial arena instead of its workspace's own buffer."""
        self.L = _lib.lib()
        self.index = index
        self.arena_offset = arena_offset
        self.named = arena_offset is not None  # holds a workspace by name until it is closed (nobody else may take that one meanwhile)
        if isinstance(requests, RequestBatch):
            self.reqs, n, arr = requests.reqs, requests.n, requests.arr
        else:
            self.reqs = [_as_request(r) for r in requests]
            n = len(self.reqs)
            arr = (C.c_void_p * n)(*[r.h for r in self.reqs])
        h = C.c_void_p()
        if arena_offset is None:
            _lib.check(self.L.vq_search_batch_partial(index.h, arr, n, C.byref(h)))
        else:
            _lib.check(self.L.vq_search_batch_partial_at(index.h, arr, n, slot, arena_offset, C.byref(h)))
        self.h = h
        self.n = n

    @property
    def nbytes(self):
        return int(self.L.vq_partial_bytes(self.h))

    @property
    def total_nbytes(self):
        """all-gathered part + histograms: what the partial occupies"""
        return int(self.L.vq_partial_total_bytes(self.h))

    @property
    def device_ptr(self):
        return int(self.L.vq_partial_device_ptr(self.h) or 0)

    @property
    def hist_nbytes(self):
        """Bytes of the batch's facet histograms (u32 counts): summed over the shards by an all-reduce, not gathered."""
        return int(self.L.vq_partial_hist_bytes(self.h))

    @property
    def hist_device_ptr(self):
        return int(self.L.vq_partial_hist_device_ptr(self.h) or 0)

    def merge(self, gathered_device_ptr=None, num_shards=1, raise_on_error=True):
        outs = (C.c_void_p * self.n)()
        status = (C.c_int * self.n)()
        _lib.check(self.L.vq_merge_partials(self.index.h, self.h, C.c_void_p(gathered_device_ptr) if gathered_device_ptr else None, num_shards, outs, status))
        results = []
        for i in range(self.n):
            if status[i] != 0:
                if raise_on_error:
                    raise VelociError(status[i], self.L.vq_last_error().decode("utf-8", "replace"))
                results.append(VelociError(status[i], _lib.ERR_NAMES.get(status[i], "error")))
            else:
                results.append(_take_result(self.L, C.c_void_p(outs[i])))
        return results

    def merge_flat(self, gathered_device_ptr=None, num_shards=1, stride=10, out=None, offset=0, shard_stride=0):
        """out: (num_hits, counts, ids, scores, status) arrays of a larger batch; this partial's rows start at `offset` (a pipeline of
        chunks fills one set of arrays instead of concatenating per-chunk ones)."""
        n = self.n
        if out is None:
            out = (np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, stride), np.uint32), np.zeros((n, stride), np.float32), np.zeros(n, np.int32))
            offset = 0
        num_hits, counts, ids, scores, status = out
        at = lambda a, row: C.c_void_p(a.ctypes.data + row * a.strides[0])
        if shard_stride:  # the shards' copies of this partial lie `shard_stride` bytes apart (one gathered arena holds several partials)
            _lib.check(self.L.vq_merge_partials_flat_strided(self.index.h, self.h, C.c_void_p(gathered_device_ptr), num_shards, shard_stride, stride,
                                                             at(num_hits, offset), at(counts, offset), at(ids, offset), at(scores, offset), at(status, offset)))
            return out
        _lib.check(self.L.vq_merge_partials_flat(self.index.h, self.h, C.c_void_p(gathered_device_ptr) if gathered_device_ptr else None, num_shards, stride,
                                                 at(num_hits, offset), at(counts, offset), at(ids, offset), at(scores, offset), at(status, offset)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.vq_partial_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
