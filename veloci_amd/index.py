"""Decoded index arrays (`IndexData`) and the HBM-staged index (`Index`).

`IndexData` mirrors the reference's `PersistenceIndices` (src/persistence.rs:52-60) at the decoded
level: stores are kept under the reference's own index names.  `IndexData.load_into(target)` feeds
any object with the builder's add_* methods — the HIP library's builder here, the CPU oracle in tests.
"""
import ctypes as C

import numpy as np

from . import _lib

TEXTINDEX = ".textindex"


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def csr_from_lists(lists):
    """list of sequences -> (offsets u64[n+1], values u32[])."""
    offsets = np.zeros(len(lists) + 1, np.uint64)
    if lists:
        offsets[1:] = np.cumsum([len(l) for l in lists], dtype=np.uint64)
    values = np.concatenate([_u32(l) for l in lists]) if lists and int(offsets[-1]) else np.zeros(0, np.uint32)
    return offsets, _u32(values)


class IndexData:
    def __init__(self, num_anchors):
        self.num_anchors = int(num_anchors)
        self.fst = {}                    # path -> (term_bytes u8, term_offsets u64, terms list[bytes])
        self.token_to_anchor_score = {}  # path -> (offsets u64, anchors u32, scores u32, global_lens u64|None)
        self.key_value_stores = {}       # path -> (key_base, offsets u64, values u32)
        self.phrase_pair_to_anchor = {}  # path -> (t1 u32, t2 u32, offsets u64, anchors u32)
        self.boost = {}                  # path -> (key_base, present u8|None, value_bits u32)
        self.columns = {}                # field -> (is_anchor_identity_column, tokenize)

    # -- builders -------------------------------------------------------------------------------
    def add_fst(self, path, terms):
        """terms: bytewise-sorted unique list of str/bytes; ordinal == term id (create_fulltext.rs:71-80)."""
        tb = [t.encode() if isinstance(t, str) else bytes(t) for t in terms]
        if any(tb[i] >= tb[i + 1] for i in range(len(tb) - 1)):
            raise ValueError("terms must be bytewise sorted and unique")
        offsets = np.zeros(len(tb) + 1, np.uint64)
        if tb:
            offsets[1:] = np.cumsum([len(t) for t in tb], dtype=np.uint64)
        data = np.frombuffer(b"".join(tb), dtype=np.uint8).copy() if tb else np.zeros(0, np.uint8)
        self.fst[path] = (data, offsets, tb)

    def term_id(self, path, term):
        import bisect
        tb = self.fst[path][2]
        t = term.encode() if isinstance(term, str) else term
        i = bisect.bisect_left(tb, t)
        return i if i < len(tb) and tb[i] == t else None

    def add_token_to_anchor_score(self, path, offsets, anchors, scores, global_lens=None):
        self.token_to_anchor_score[path] = (_u64(offsets), _u32(anchors), _u32(scores), None if global_lens is None else _u64(global_lens))

    def add_key_value_store(self, path, offsets, values, key_base=0):
        self.key_value_stores[path] = (int(key_base), _u64(offsets), _u32(values))

    def add_phrase_pair_to_anchor(self, path, t1, t2, offsets, anchors):
        self.phrase_pair_to_anchor[path] = (_u32(t1), _u32(t2), _u64(offsets), _u32(anchors))

    def add_boost(self, path, values_f32, present=None, key_base=0):
        bits = np.ascontiguousarray(values_f32, dtype=np.float32).view(np.uint32)
        self.boost[path] = (int(key_base), None if present is None else np.ascontiguousarray(present, dtype=np.uint8), bits)

    def set_column_meta(self, field, is_anchor_identity_column, tokenize=True):
        self.columns[field] = (bool(is_anchor_identity_column), bool(tokenize))

    # -- feeding a builder ------------------------------------------------------------------------
    def load_into(self, target):
        for field, (ident, tok) in self.columns.items():
            target.set_column_meta(field, ident, tok)
        for path, (data, offsets, _) in self.fst.items():
            target.add_fst(path, data, offsets)
        for path, (offsets, anchors, scores, gl) in self.token_to_anchor_score.items():
            target.add_token_to_anchor_score(path, offsets, anchors, scores, gl)
        for path, (kb, offsets, values) in self.key_value_stores.items():
            target.add_key_value_store(path, kb, offsets, values)
        for path, (t1, t2, offsets, anchors) in self.phrase_pair_to_anchor.items():
            target.add_phrase_pair_to_anchor(path, t1, t2, offsets, anchors)
        for path, (kb, present, bits) in self.boost.items():
            target.add_boost(path, kb, present, bits)
        return target


class _Builder:
    """Thin wrapper of vq_index_builder with the add_* surface `IndexData.load_into` expects."""

    def __init__(self, num_anchors, doc_lo, doc_hi):
        self.L = _lib.lib()
        self.h = C.c_void_p(self.L.vq_index_builder_new(num_anchors, doc_lo, doc_hi))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vq_index_builder_free(self.h)
            self.h = None

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def add_fst(self, path, term_bytes, term_offsets):
        _lib.check(self.L.vq_index_add_fst(self.h, path.encode(), len(term_offsets) - 1, self._p(term_bytes), self._p(term_offsets)))

    def add_token_to_anchor_score(self, path, offsets, anchors, scores, global_lens=None):
        _lib.check(self.L.vq_index_add_token_to_anchor_score(self.h, path.encode(), len(offsets) - 1, self._p(offsets), self._p(anchors), self._p(scores),
                                                             self._p(global_lens)))

    def add_key_value_store(self, path, key_base, offsets, values):
        _lib.check(self.L.vq_index_add_key_value_store(self.h, path.encode(), key_base, len(offsets) - 1, self._p(offsets), self._p(values)))

    def add_phrase_pair_to_anchor(self, path, t1, t2, offsets, anchors):
        _lib.check(self.L.vq_index_add_phrase_pair_to_anchor(self.h, path.encode(), len(t1), self._p(t1), self._p(t2), self._p(offsets), self._p(anchors)))

    def add_boost(self, path, key_base, present, value_bits):
        _lib.check(self.L.vq_index_add_boost(self.h, path.encode(), key_base, len(value_bits), self._p(present), self._p(value_bits)))

    def set_column_meta(self, field, is_anchor_identity_column, tokenize=True):
        _lib.check(self.L.vq_index_set_column_meta(self.h, field.encode(), int(is_anchor_identity_column), int(tokenize)))


class Index:
    """An index shard staged in HBM (`vq_index`): the `&Persistence` argument of search::search."""

    def __init__(self, data, device=0, doc_lo=0, doc_hi=None):
        self.L = _lib.lib()
        self.num_anchors = data.num_anchors
        self.doc_lo = int(doc_lo)
        self.doc_hi = int(data.num_anchors if doc_hi is None else doc_hi)
        b = _Builder(data.num_anchors, self.doc_lo, self.doc_hi)
        data.load_into(b)
        h = C.c_void_p()
        _lib.check(self.L.vq_index_build(b.h, int(device), C.byref(h)))
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.L.vq_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _lib.check(self.L.vq_index_set_stream(self.h, C.c_void_p(hip_stream) if hip_stream else None))

    def set_streams(self, scan_stream, finish_stream):
        """vq_index_set_streams: scans on one caller stream, merges / downloads on another (hipStream_t handles as ints)."""
        _lib.check(self.L.vq_index_set_streams(self.h, C.c_void_p(scan_stream), C.c_void_p(finish_stream)))

    def set_allreduce(self, fn):
        """fn(numpy uint64 array) must sum the array over all shards in place (vq_index_set_allreduce); None removes it."""
        if fn is None:
            self._allreduce_cb = None
            _lib.check(self.L.vq_index_set_allreduce(self.h, None, None))
            return
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t)

        def trampoline(_ctx, values, n):
            try:
                fn(np.ctypeslib.as_array(values, shape=(n,)))
                return 0
            except Exception:  # noqa: BLE001 - reported as VQ_ERR_DEVICE by the library
                return 1

        self._allreduce_cb = proto(trampoline)  # keep the callback alive
        _lib.check(self.L.vq_index_set_allreduce(self.h, C.cast(self._allreduce_cb, C.c_void_p), None))

    @property
    def device_bytes(self):
        return int(self.L.vq_index_device_bytes(self.h))

    @property
    def speculative_reruns(self):
        """requests that ran a second time because a speculative kernel's short cut could not be confirmed (vq_index_speculative_reruns)"""
        return int(self.L.vq_index_speculative_reruns(self.h))

    @property
    def partial_arena_ptr(self):
        """device address of the partial arena (vq_index_partial_arena_ptr): where the chunks of a one-collective sharded step put their partials"""
        p = self.L.vq_index_partial_arena_ptr(self.h)
        if not p:
            _lib.check(5)
        return int(p)

    def profile_enable(self, on=True):
        _lib.check(self.L.vq_profile_enable(self.h, int(on)))

    def profile_json(self, reset=True):
        """vq_profile_json: per-kernel device time, launches, layout / algorithmic bytes since the last reset."""
        import json
        return json.loads(self.L.vq_profile_json(self.h, int(reset)).decode())

    def profile_read(self, reset=True):
        ms, n, b = C.c_double(), C.c_uint64(), C.c_uint64()
        _lib.check(self.L.vq_profile_read(self.h, int(reset), C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value
