"""TEST INFRASTRUCTURE — ctypes binding of the CPU oracle (oracle/libveloci_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


def build(force=False):
    so = os.path.join(_HERE, "libveloci_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.vo_last_error.restype = C.c_char_p
        L.vo_index_new.restype = C.c_void_p
        L.vo_index_new.argtypes = [C.c_uint32]
        L.vo_index_free.argtypes = [C.c_void_p]
        L.vo_index_add_fst.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.vo_index_add_token_to_anchor_score.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vo_index_add_key_value_store.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.vo_index_add_phrase_pair_to_anchor.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vo_index_add_boost.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.vo_index_set_column_meta.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.vo_search_json.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        for name, rt in [("num_hits", C.c_uint64), ("execution_time_ns", C.c_uint64), ("len", C.c_size_t), ("ids", u32p), ("scores", f32p),
                         ("num_facets", C.c_size_t)]:
            f = getattr(L, "vo_result_" + name)
            f.restype = rt
            f.argtypes = [C.c_void_p]
        L.vo_result_facet_field.restype = C.c_char_p
        L.vo_result_facet_field.argtypes = [C.c_void_p, C.c_size_t]
        L.vo_result_facet_len.restype = C.c_size_t
        L.vo_result_facet_len.argtypes = [C.c_void_p, C.c_size_t]
        L.vo_result_facet_value.restype = C.c_char_p
        L.vo_result_facet_value.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.vo_result_facet_count.restype = C.c_uint64
        L.vo_result_facet_count.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.vo_result_free.argtypes = [C.c_void_p]
        L.vo_result_why_found_terms_json.restype = C.c_char_p
        L.vo_result_why_found_terms_json.argtypes = [C.c_void_p]
        L.vo_result_why_found_info_json.restype = C.c_char_p
        L.vo_result_why_found_info_json.argtypes = [C.c_void_p]
        L.vo_result_explain_json.restype = C.c_char_p
        L.vo_result_explain_json.argtypes = [C.c_void_p]
        L.vo_suggest_json.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.vo_highlight_json.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.vo_highlight_text.restype = C.c_void_p
        L.vo_highlight_text.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int)]
        L.vo_string_free.argtypes = [C.c_void_p]
        L.vo_suggest_len.restype = C.c_size_t
        L.vo_suggest_len.argtypes = [C.c_void_p]
        L.vo_suggest_text.restype = C.c_char_p
        L.vo_suggest_text.argtypes = [C.c_void_p, C.c_size_t]
        L.vo_suggest_score.restype = C.c_float
        L.vo_suggest_score.argtypes = [C.c_void_p, C.c_size_t]
        L.vo_suggest_term_id.restype = C.c_uint32
        L.vo_suggest_term_id.argtypes = [C.c_void_p, C.c_size_t]
        L.vo_suggest_free.argtypes = [C.c_void_p]
        L.vo_bench_search.restype = C.c_double
        L.vo_bench_search.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, u64p]
        L.vo_op_default_score_for_distance.restype = C.c_float
        L.vo_op_f16_to_f32.restype = C.c_float
        L.vo_op_f16_to_f32.argtypes = [C.c_uint16]
        L.vo_op_f32_to_f16.restype = C.c_uint16
        L.vo_op_f32_to_f16.argtypes = [C.c_float]
        L.vo_op_score_expression.argtypes = [C.c_char_p, C.c_float, f32p]
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class OracleResult:
    def __init__(self, num_hits, ids, scores, facets, execution_time_ns=0):
        self.num_hits = num_hits
        self.ids = ids
        self.scores = scores
        self.facets = facets  # list of (field, [(value, count)])
        self.execution_time_ns = execution_time_ns


def highlight_text(text, terms, snippet_info=None, tokenized=True):
    """highlight_field::highlight_text (highlight_field.rs:92-146) -> the snippet or None"""
    import json
    L = lib()
    raw, tj = text.encode(), json.dumps(list(terms)).encode()
    si = json.dumps(snippet_info).encode() if snippet_info is not None else b""
    none = C.c_int(0)
    p = L.vo_highlight_text(raw, len(raw), tj, len(tj), si, len(si), int(tokenized), C.byref(none))
    if not p:
        if none.value:
            return None
        raise OracleError(1, L.vo_last_error().decode())
    try:
        return C.string_at(p).decode()
    finally:
        L.vo_string_free(p)


class OracleIndex:
    """Receives the same add_* calls as the product's index builder (veloci_amd.index.IndexData.load_into)."""

    def __init__(self, num_anchors):
        self.L = lib()
        self.h = C.c_void_p(self.L.vo_index_new(num_anchors))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vo_index_free(self.h)
            self.h = None

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def add_fst(self, path, term_bytes, term_offsets):
        self.L.vo_index_add_fst(self.h, path.encode(), len(term_offsets) - 1, self._p(term_bytes), self._p(term_offsets))

    def add_token_to_anchor_score(self, path, offsets, anchors, scores, global_lens=None):
        self.L.vo_index_add_token_to_anchor_score(self.h, path.encode(), len(offsets) - 1, self._p(offsets), self._p(anchors), self._p(scores),
                                                  self._p(global_lens))

    def add_key_value_store(self, path, key_base, offsets, values):
        self.L.vo_index_add_key_value_store(self.h, path.encode(), key_base, len(offsets) - 1, self._p(offsets), self._p(values))

    def add_phrase_pair_to_anchor(self, path, t1, t2, offsets, anchors):
        self.L.vo_index_add_phrase_pair_to_anchor(self.h, path.encode(), len(t1), self._p(t1), self._p(t2), self._p(offsets), self._p(anchors))

    def add_boost(self, path, key_base, present, value_bits):
        self.L.vo_index_add_boost(self.h, path.encode(), key_base, len(value_bits), self._p(present), self._p(value_bits))

    def set_column_meta(self, field, is_anchor_identity_column, tokenize=True):
        self.L.vo_index_set_column_meta(self.h, field.encode(), int(is_anchor_identity_column), int(tokenize))

    def search_json(self, js):
        if not isinstance(js, (bytes, bytearray)):
            js = js.encode()
        out = C.c_void_p()
        rc = self.L.vo_search_json(self.h, js, len(js), C.byref(out))
        if rc != 0:
            raise OracleError(rc, self.L.vo_last_error().decode())
        try:
            n = self.L.vo_result_len(out)
            ids = np.ctypeslib.as_array(self.L.vo_result_ids(out), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
            scores = np.ctypeslib.as_array(self.L.vo_result_scores(out), shape=(n,)).copy() if n else np.zeros(0, np.float32)
            facets = []
            for f in range(self.L.vo_result_num_facets(out)):
                fl = self.L.vo_result_facet_len(out, f)
                facets.append((self.L.vo_result_facet_field(out, f).decode(),
                               [(self.L.vo_result_facet_value(out, f, i).decode(), int(self.L.vo_result_facet_count(out, f, i))) for i in range(fl)]))
            res = OracleResult(int(self.L.vo_result_num_hits(out)), ids, scores, facets, int(self.L.vo_result_execution_time_ns(out)))
            import json as _json
            res.why_found_terms = _json.loads(self.L.vo_result_why_found_terms_json(out).decode())
            # search.rs:220-224 (why_found with select): {anchor id: {field: [highlighted texts]}}
            res.why_found_info = {int(k): v for k, v in _json.loads(self.L.vo_result_why_found_info_json(out).decode()).items()}
            res.explain_json = self.L.vo_result_explain_json(out).decode()  # per hit: null or the Explain records (search.rs:86,96)
            res.explain = _json.loads(res.explain_json)
            return res
        finally:
            self.L.vo_result_free(out)

    def suggest_json(self, js):
        """suggest_multi / suggest (search_field.rs:194-231): -> [(text, score, term_id)]"""
        if not isinstance(js, (bytes, bytearray)):
            js = js.encode()
        out = C.c_void_p()
        rc = self.L.vo_suggest_json(self.h, js, len(js), C.byref(out))
        if rc != 0:
            raise OracleError(rc, self.L.vo_last_error().decode())
        try:
            return [(self.L.vo_suggest_text(out, i).decode(), float(self.L.vo_suggest_score(out, i)), int(self.L.vo_suggest_term_id(out, i)))
                    for i in range(self.L.vo_suggest_len(out))]
        finally:
            self.L.vo_suggest_free(out)

    def highlight_json(self, js):
        """search_field::highlight (search_field.rs:233-245) of a bare RequestSearchPart: -> [(snippet, score, text_id)]"""
        if not isinstance(js, (bytes, bytearray)):
            js = js.encode()
        out = C.c_void_p()
        rc = self.L.vo_highlight_json(self.h, js, len(js), C.byref(out))
        if rc != 0:
            raise OracleError(rc, self.L.vo_last_error().decode())
        try:
            return [(self.L.vo_suggest_text(out, i).decode(), float(self.L.vo_suggest_score(out, i)), int(self.L.vo_suggest_term_id(out, i)))
                    for i in range(self.L.vo_suggest_len(out))]
        finally:
            self.L.vo_suggest_free(out)

    def bench(self, jsons, repeat=1, threads=1):
        """Run the requests on `threads` host threads; returns (wall_seconds, latencies_ns, checksum)."""
        enc = [j.encode() if isinstance(j, str) else j for j in jsons]
        arr = (C.c_char_p * len(enc))(*enc)
        lens = (C.c_size_t * len(enc))(*[len(e) for e in enc])
        lat = np.zeros(len(enc) * repeat, np.uint64)
        chk = C.c_uint64()
        secs = self.L.vo_bench_search(self.h, arr, lens, len(enc), repeat, threads, lat.ctypes.data_as(C.c_void_p), C.byref(chk))
        return secs, lat, chk.value


# ---- single-function entry points (golden vectors) ----------------------------------------------

def _flat(lists, with_scores):
    lens = np.array([len(l) for l in lists], np.uint32)
    ids = np.array([e[0] if with_scores else e for l in lists for e in l], np.uint32)
    scores = np.array([e[1] for l in lists for e in l], np.float32) if with_scores else None
    return lens, ids, scores


def _hits(n, oi, osc=None):
    if osc is None:
        return [int(x) for x in oi[:n]]
    return [(int(oi[i]), float(osc[i])) for i in range(n)]


def _p(a, t):
    return a.ctypes.data_as(t)


def intersect_hits_score(lists):
    lens, ids, sc = _flat(lists, True)
    oi = np.zeros(len(ids) + 1, np.uint32)
    osc = np.zeros(len(ids) + 1, np.float32)
    n = lib().vo_op_intersect_hits_score(len(lists), _p(lens, u32p), _p(ids, u32p), _p(sc, f32p), _p(oi, u32p), _p(osc, f32p))
    return _hits(n, oi, osc)


def union_hits_score(lists, terms):
    lens, ids, sc = _flat(lists, True)
    oi = np.zeros(len(ids) + 1, np.uint32)
    osc = np.zeros(len(ids) + 1, np.float32)
    t = (C.c_char_p * len(terms))(*[x.encode() for x in terms])
    n = lib().vo_op_union_hits_score(len(lists), _p(lens, u32p), _p(ids, u32p), _p(sc, f32p), t, _p(oi, u32p), _p(osc, f32p))
    return _hits(n, oi, osc)


def intersect_hits_ids(lists):
    lens, ids, _ = _flat(lists, False)
    oi = np.zeros(len(ids) + 1, np.uint32)
    n = lib().vo_op_intersect_hits_ids(len(lists), _p(lens, u32p), _p(ids, u32p), _p(oi, u32p))
    return _hits(n, oi)


def union_hits_ids(lists):
    lens, ids, _ = _flat(lists, False)
    oi = np.zeros(len(ids) + 1, np.uint32)
    n = lib().vo_op_union_hits_ids(len(lists), _p(lens, u32p), _p(ids, u32p), _p(oi, u32p))
    return _hits(n, oi)


def intersect_score_hits_with_ids(hits, ids):
    _, hi, hs = _flat([hits], True)
    f = np.array(ids, np.uint32)
    oi = np.zeros(len(hi) + 1, np.uint32)
    osc = np.zeros(len(hi) + 1, np.float32)
    n = lib().vo_op_intersect_score_hits_with_ids(len(hi), _p(hi, u32p), _p(hs, f32p), len(f), _p(f, u32p), _p(oi, u32p), _p(osc, f32p))
    return _hits(n, oi, osc)


def boost_hits_ids_vec_multi(hits, boost_lists, boost_vals=None):
    _, hi, hs = _flat([hits], True)
    lens, bids, _ = _flat(boost_lists, False)
    bv = np.array([float("nan") if v is None else v for v in (boost_vals or [None] * len(boost_lists))], np.float32)
    oi = np.zeros(len(hi) + 1, np.uint32)
    osc = np.zeros(len(hi) + 1, np.float32)
    n = lib().vo_op_boost_hits_ids_vec_multi(len(hi), _p(hi, u32p), _p(hs, f32p), len(boost_lists), _p(lens, u32p), _p(bids, u32p), _p(bv, f32p),
                                             _p(oi, u32p), _p(osc, f32p))
    return _hits(n, oi, osc)


_BF = {None: -1, "Log2": 0, "Log10": 1, "Multiply": 2, "Add": 3, "Replace": 4}


def apply_boost_values_anchor(hits, boosts, boost_fun=None, param=None, expression=None):
    _, hi, hs = _flat([hits], True)
    _, bi, bs = _flat([boosts], True)
    oi = np.zeros(len(hi) + 1, np.uint32)
    osc = np.zeros(len(hi) + 1, np.float32)
    n = lib().vo_op_apply_boost_values_anchor(len(hi), _p(hi, u32p), _p(hs, f32p), len(bi), _p(bi, u32p), _p(bs, f32p), _BF[boost_fun],
                                              C.c_float(param or 0.0), int(param is not None), expression.encode() if expression else None,
                                              _p(oi, u32p), _p(osc, f32p))
    return _hits(n, oi, osc)


def top_n_sort(hits, top_n):
    _, hi, hs = _flat([hits], True)
    oi = np.zeros(len(hi) + 1, np.uint32)
    osc = np.zeros(len(hi) + 1, np.float32)
    n = lib().vo_op_top_n_sort(len(hi), _p(hi, u32p), _p(hs, f32p), top_n, _p(oi, u32p), _p(osc, f32p))
    return _hits(n, oi, osc)


def distance(a, b):
    return lib().vo_op_distance(a.encode(), b.encode())


def levenshtein(a, b, transposition=False, ci=False):
    return lib().vo_op_levenshtein(a.encode(), b.encode(), int(transposition), int(ci))


def score_expression(expr, rank):
    out = C.c_float()
    rc = lib().vo_op_score_expression(expr.encode(), C.c_float(rank), C.byref(out))
    if rc:
        raise OracleError(rc, lib().vo_last_error().decode())
    return out.value


def default_score_for_distance(d, prefix):
    return lib().vo_op_default_score_for_distance(d, int(prefix))


def calculate_token_score(pos, occ, ntok, exact=False):
    return lib().vo_op_calculate_token_score(pos, occ, ntok, int(exact))


def f32_to_f16(f):
    return lib().vo_op_f32_to_f16(C.c_float(f))


def f16_to_f32(h):
    return lib().vo_op_f16_to_f32(h)


def steps_to_anchor(path):
    buf = C.create_string_buffer(4096)
    lib().vo_op_steps_to_anchor(path.encode(), buf, 4096)
    return [s for s in buf.value.decode().split("\n") if s]
