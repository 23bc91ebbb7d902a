"""Mini-indexer: JSON documents + field configuration -> decoded index arrays (`IndexData`).

A host-side restatement of the reference's index creation at the level the query path consumes
(SURVEY.md §8f-1), so that the reference's integration-test fixtures (inline JSON documents, `tests/all/*.rs`)
can be replayed against the MI355X query path and the CPU oracle:

* document walk, anchor / value id assignment ... json_converter/src/lib.rs:69-160, 187-207
* field configuration, feature -> index gating . src/create/fields_config.rs:17-90, src/create/features.rs:20-94
* tokenizer (grouped separator runs) ........... src/tokenizer/simple_tokenizer_group.rs:51-82, src/tokenizer/mod.rs:21-23
* term counting, term ids = bytewise rank ...... src/create/create_fulltext.rs:25-152
* identity-column detection .................... src/create/create_fulltext.rs:37-38
* per-text / per-token postings and scores ..... src/create.rs:187-283, src/create/calculate_score.rs:6-49
* list building (sort, dedup, score merge) ..... src/create.rs:350-411, 505-517, 554-722

Not covered (not consumed by the query path built here): the document store, text_id_to_token_ids
(why_found / select), token values, on-disk formats.
"""
import json

import numpy as np

from .index import IndexData, csr_from_lists

TEXTINDEX = ".textindex"
ALL_FIELD_CONFIG = "*GLOBAL*"
DEFAULT_SEPARATORS = [' ', '\t', '\n', '\r', ':', '(', ')', ',', '.', '…', ';', '・', '’', '—', '-', '\\', '[', ']', '{', '}', '<', '>', '\'',
                      '"', '“', '™']  # tokenizer/mod.rs:21-23
DEFAULT_TEXT_LENGTH_STORE = 64  # metadata.rs:67-69

# create/features.rs:69-93: an index type is disabled unless one of these features is requested
_INDEX_FEATURES = {
    "TokensToTextID": ["All", "TokensToTextID", "BoostTextLocality", "Highlight", "BoostingFieldData"],
    "TokenToAnchorIDScore": ["All", "Search"],
    "ParentToValueID": ["All", "Select", "Facets"],
    "ValueIDToParent": ["All", "BoostingFieldData"],
    "PhrasePairToAnchor": ["All", "PhraseBoost"],
    "TextIDToTokenIds": ["All", "Select", "WhyFound"],
    "TextIDToParent": ["All", "BoostingFieldData"],
    "ParentToTextID": ["All", "Facets", "Select"],
    "TextIDToAnchor": ["All", "BoostTextLocality", "Select", "Filters"],
}
_ALL_FEATURES = ["TokensToTextID", "BoostTextLocality", "BoostingFieldData", "Search", "Filters", "Facets", "Select", "WhyFound", "Highlight", "PhraseBoost"]


class FieldConfig:
    """create/fields_config.rs:56-84."""

    def __init__(self, raw=None, is_default=False):
        raw = raw or {}
        self.facet = bool(raw.get("facet", False))
        self.fulltext = raw.get("fulltext")
        self.disabled_indices = set(raw["disabled_indices"]) if raw.get("disabled_indices") is not None else None
        self.features = set(raw["features"]) if raw.get("features") is not None else None
        self.disabled_features = set(raw["disabled_features"]) if raw.get("disabled_features") is not None else None
        self.boost = raw.get("boost")
        if is_default:  # impl Default: default features, tokenizing full text
            self.features = {"Search", "TokensToTextID"}
            self.fulltext = {"tokenize": True}

    def is_index_enabled(self, index):
        return self.disabled_indices is None or index not in self.disabled_indices

    def fulltext_options(self):
        ft = self.fulltext if self.fulltext is not None else {"tokenize": True}
        return {
            "tokenize": bool(ft.get("tokenize", True)) if self.fulltext is not None else True,
            "separators": ft.get("tokenize_on_chars") or DEFAULT_SEPARATORS,
            "limit": int(ft.get("do_not_store_text_longer_than", DEFAULT_TEXT_LENGTH_STORE)),
        }


def parse_config(indices):
    """config_from_string (fields_config.rs:98-112): JSON when the text starts with '{', TOML otherwise; dicts pass through."""
    if isinstance(indices, dict):
        raw = indices
    elif indices.strip().startswith("{"):
        raw = json.loads(indices)
    elif indices.strip() == "":
        raw = {}
    else:
        import tomli
        raw = tomli.loads(indices)
    cfg = {k: FieldConfig(v) for k, v in raw.items()}
    # features_to_indices (fields_config.rs:30-52)
    if ALL_FIELD_CONFIG not in cfg:
        cfg[ALL_FIELD_CONFIG] = FieldConfig(is_default=True)
    for key, val in cfg.items():
        if val.features is not None and val.disabled_features is not None:
            raise ValueError("features and disabled_features are not allowed at the same time in field %r" % key)
        feats = val.features
        if feats is None and val.disabled_features is not None:
            # Features::invert (features.rs:22-37) keeps the listed features (it does not invert); restated as is
            feats = {f for f in _ALL_FEATURES if f in val.disabled_features}
        if feats is not None:
            disabled = {idx for idx, needs in _INDEX_FEATURES.items() if not any(f in feats for f in needs)}
            val.disabled_indices = (val.disabled_indices or set()) | disabled
    return cfg


def field_config(cfg, path):
    if path.endswith(TEXTINDEX):
        path = path[:-len(TEXTINDEX)]
    return cfg.get(path, cfg[ALL_FIELD_CONFIG])


def tokenize(text, separators):
    """Yield (token, is_separator); consecutive separator characters form one token (simple_tokenizer_group.rs:51-82)."""
    out = []
    last = 0
    in_sep = False
    pos = 0  # positions in code points; slices of a Python str are per code point, like the byte slices on char boundaries
    for pos, ch in enumerate(text):
        if ch in separators:
            if pos == 0:
                in_sep = True
            elif not in_sep:
                out.append((text[last:pos], False))
                in_sep = True
                last = pos
        elif in_sep:
            out.append((text[last:pos], True))
            in_sep = False
            last = pos
    if last != len(text):
        out.append((text[last:], in_sep))
    return out


def has_tokens(tokens):
    return len(tokens) >= 2  # simple_tokenizer_group.rs:10-14


def convert_to_string(value):
    """json_converter/src/lib.rs:5-14."""
    if isinstance(value, bool):
        return "true" if value else "false"
    if isinstance(value, str):
        return value
    if isinstance(value, int):
        return str(value) if value >= 0 else ""  # only is_u64 / is_f64 numbers are rendered
    if isinstance(value, float):
        r = repr(value)
        if "e" in r or "E" in r or "inf" in r or "nan" in r:
            raise NotImplementedError("float rendering outside the plain decimal range: %r" % value)
        return r[:-2] if r.endswith(".0") else r
    return ""


class _IdHolder:
    def __init__(self):
        self.ids = {}

    def get_id(self, path):
        if path in self.ids:
            self.ids[path] += 1
        else:
            self.ids[path] = 0
        return self.ids[path]


def _walk(data, anchor_id, ids, parent_id, path, name, cb_text, cb_ids):
    """for_each_elemento (json_converter/src/lib.rs:111-160); serde_json maps iterate in key order (no preserve_order)."""
    if isinstance(data, list):
        path = path + name + "[]"
        for el in data:
            vid = ids.get_id(path)
            cb_ids(anchor_id, path, vid, parent_id)
            _walk(el, anchor_id, ids, vid, path, "", cb_text, cb_ids)
    elif isinstance(data, dict):
        path = path + name
        if path != "":
            path += "."
        for key in sorted(data.keys(), key=lambda k: k.encode("utf-8")):
            _walk(data[key], anchor_id, ids, parent_id, path, key, cb_text, cb_ids)
    elif data is not None:
        cb_text(anchor_id, convert_to_string(data), path + name, parent_id)


def _for_each_element(docs, cb_text, cb_ids):
    ids = _IdHolder()
    for doc in docs:
        root = ids.get_id("")
        _walk(doc, root, ids, root, "", "", cb_text, cb_ids)


def token_score(token_best_pos, num_occurences, num_tokens_in_text, is_exact):
    """calculate_token_score_for_entry (create/calculate_score.rs:34-49), f32 arithmetic."""
    f = np.float32
    score = f(400.0) if is_exact else f(2000.0) / (np.log2(f(token_best_pos) + f(10.0)) + f(10.0))
    m = np.log10(f(num_occurences) + f(1000.0)) - f(2.0)
    m = m - (m - f(1.0)) * f(0.7)
    score = f(score / m)
    t = np.log10(f(num_tokens_in_text + 10))
    t = t - (t - f(1.0)) * f(0.7)
    score = f(score / t)
    return int(score)


def _multi_store(pairs, sort_and_dedup):
    """(key, value) pairs -> list of per-key value lists (index = key); stream_iter_to_indirect_index create.rs:350-366."""
    if not pairs:
        return []
    n = max(k for k, _ in pairs) + 1
    rows = [[] for _ in range(n)]
    for k, v in pairs:
        rows[k].append(v)
    if sort_and_dedup:
        rows = [sorted(set(r)) for r in rows]
    return rows


def build_index(docs, indices="", token_values=None):
    """docs: list of JSON objects (one anchor each).  Returns (IndexData, info) with info = {path: {"terms": [...], "identity": bool}}.
    token_values: optional (entries, path) with entries = [{"text": .., "value": ..}] — per-term boost values of field `path`
    (add_token_values_to_tokens, create/token_values_to_tokens.rs:26-75)."""
    cfg = parse_config(indices)
    num_docs = len(docs)

    # ---- pass 1: count texts and tokens per path (get_allterms_per_path, create_fulltext.rs:114-152)
    terms = {}       # path -> {text: occurrences}
    long_count = {}  # path -> id_counter_for_large_texts
    opts = {}

    def count_text(anchor, value, path, parent):
        o = opts.setdefault(path, field_config(cfg, path).fulltext_options())
        t = terms.setdefault(path, {})
        long_count.setdefault(path, 0)
        if o["limit"] < len(value.encode("utf-8")):
            long_count[path] += 1
        else:
            t[value] = t.get(value, 0) + 1
        if o["tokenize"]:
            toks = tokenize(value, o["separators"])
            if has_tokens(toks):
                for tok, _ in toks:
                    t[tok] = t.get(tok, 0) + 1

    _for_each_element(docs, count_text, lambda *a: None)

    # ---- term ids: rank in bytewise order (set_ids, create_fulltext.rs:71-80); identity columns (:37-38)
    term_id = {}
    identity = {}
    for path, t in terms.items():
        ordered = sorted(t.keys(), key=lambda s: s.encode("utf-8"))
        term_id[path] = {s: i for i, s in enumerate(ordered)}
        identity[path] = ("[]" not in path) and num_docs == len(t) and all(c == 1 for c in t.values())
        for s in ordered:
            if len(s.encode("utf-8")) > opts[path]["limit"]:
                raise NotImplementedError("token longer than do_not_store_text_longer_than (term id != FST ordinal): %r" % s)

    # ---- pass 2: raw relations (parse_json_and_prepare_indices, create.rs:163-325)
    raw = {}

    def rel(path):
        if path not in raw:
            fc = field_config(cfg, path)
            on = fc.is_index_enabled
            raw[path] = {
                "fc": fc,
                "tokens_to_text_id": [] if on("TokensToTextID") else None,
                "text_id_to_token_ids": {} if on("TextIDToTokenIds") else None,  # text id -> its tokens, stored once (create.rs:228, 267-270)
                "text_id_to_parent": [] if on("TextIDToParent") else None,
                "text_id_to_anchor": [] if on("TextIDToAnchor") else None,
                "phrase": [] if on("PhrasePairToAnchor") else None,
                "parent_to_text_id": [] if on("ParentToTextID") else None,
                "postings": [] if on("TokenToAnchorIDScore") else None,
                "anchor_to_text_id": [] if (fc.facet and "[]" in path) else None,  # path_data.rs:72-78
                "boost": [] if fc.boost is not None else None,
                "value_id_to_anchor": [] if fc.boost is not None else None,
                "long_seen": 0,
            }
        return raw[path]

    def add(lst, item):
        if lst is not None:
            lst.append(item)

    def index_text(anchor, value, path, parent):
        d = rel(path)
        o = opts[path]
        t = terms[path]
        if o["limit"] < len(value.encode("utf-8")):  # get_text_info create.rs:139-160
            d["long_seen"] += 1
            text_id = len(t) + 1 + long_count[path] + d["long_seen"]
            occurrences = 1
        else:
            text_id = term_id[path][value]
            occurrences = t[value]
        add(d["text_id_to_parent"], (text_id, parent))
        add(d["parent_to_text_id"], (parent, text_id))
        if d["text_id_to_anchor"] is not None and not identity[path]:
            d["text_id_to_anchor"].append((text_id, anchor))
        add(d["anchor_to_text_id"], (anchor, text_id))
        if d["boost"] is not None and value.strip() != "":
            num = np.float32(float(value))
            if not np.isnan(num):
                d["boost"].append((parent, int(num.view(np.uint32))))
        add(d["value_id_to_anchor"], (parent, anchor))
        add(d["postings"], (text_id, anchor, token_score(0, occurrences, 1, True)))
        if o["tokenize"]:
            toks = tokenize(value, o["separators"])
            if has_tokens(toks):
                seen = []
                prev = None
                pos = 0
                t2t = d["text_id_to_token_ids"]
                store_tokens = t2t is not None and text_id not in t2t
                if store_tokens:
                    t2t[text_id] = []
                for tok, is_sep in toks:
                    tid = term_id[path][tok]
                    if store_tokens:
                        t2t[text_id].append(tid)
                    add(d["tokens_to_text_id"], (tid, text_id))
                    if d["postings"] is not None:
                        seen.append((tid, pos, t[tok]))
                        pos += 1
                    if not is_sep and d["phrase"] is not None:
                        if prev is not None:
                            d["phrase"].append(((prev, tid), anchor))
                        prev = tid
                if d["postings"] is not None:  # calculate_and_add_token_score_in_doc calculate_score.rs:6-31
                    best = {}
                    for tid, p, occ in sorted(seen):
                        if tid not in best:
                            best[tid] = (p, occ)
                    for tid, (p, occ) in best.items():
                        d["postings"].append((tid, anchor, token_score(p, occ, pos, False)))

    id_rel = {}

    def index_ids(anchor, path, value_id, parent):
        if path not in id_rel:
            fc = field_config(cfg, path)
            id_rel[path] = {"value_to_parent": [] if fc.is_index_enabled("ValueIDToParent") else None,
                            "parent_to_value": [] if fc.is_index_enabled("ParentToValueID") else None}
        add(id_rel[path]["value_to_parent"], (value_id, parent))
        add(id_rel[path]["parent_to_value"], (parent, value_id))

    _for_each_element(docs, index_text, index_ids)

    # ---- stores (convert_raw_path_data_to_indices, create.rs:554-722)
    data = IndexData(num_docs)
    info = {}
    for path, t in terms.items():
        ordered = sorted(t.keys(), key=lambda s: s.encode("utf-8"))
        data.add_fst(path + TEXTINDEX, [s.encode("utf-8") for s in ordered])
        data.set_column_meta(path, identity[path], opts[path]["tokenize"])
        info[path] = {"terms": ordered, "identity": identity[path]}
    for path, d in raw.items():
        tp = path + TEXTINDEX
        if d["tokens_to_text_id"] is not None:
            data.add_key_value_store(tp + ".tokens_to_text_id", *csr_from_lists(_multi_store(d["tokens_to_text_id"], True)))
        if d["text_id_to_token_ids"] is not None:  # create.rs:633-635: insertion order kept, separators included, nothing deduplicated
            # (an untokenised field's store is written too, empty — persistence.rs:258-262 loads it as a store without keys: highlight_document finds
            #  the index and no row, highlight_field.rs:196-207)
            rows = [[] for _ in range(max(d["text_id_to_token_ids"], default=-1) + 1)]
            for k, v in d["text_id_to_token_ids"].items():
                rows[k] = v
            data.add_key_value_store(tp + ".text_id_to_token_ids", *csr_from_lists(rows))
        if d["postings"] is not None:  # stream_iter_to_anchor_score create.rs:389-411
            by_token = {}
            for tid, anchor, score in d["postings"]:
                by_token.setdefault(tid, {}).setdefault(anchor, []).append(score)
            n = (max(by_token) + 1) if by_token else 0
            anchors, scores = [], []
            for tid in range(n):
                a_row, s_row = [], []
                for anchor in sorted(by_token.get(tid, {})):
                    group = by_token[tid][anchor]
                    s = max(group)
                    if len(group) > 1:
                        s += min(len(group), 5)  # small boost for multi hits
                    a_row.append(anchor)
                    s_row.append(s)
                anchors.append(a_row)
                scores.append(s_row)
            offsets, flat_a = csr_from_lists(anchors)
            _, flat_s = csr_from_lists(scores)
            data.add_token_to_anchor_score(tp + ".to_anchor_id_score", offsets, flat_a, flat_s, None)
        if d["phrase"] is not None:  # create.rs:505-517: keys sorted, anchors sorted + deduped
            groups = {}
            for key, anchor in d["phrase"]:
                groups.setdefault(key, set()).add(anchor)
            keys = sorted(groups)
            offsets, flat = csr_from_lists([sorted(groups[k]) for k in keys])
            data.add_phrase_pair_to_anchor(tp + ".phrase_pair_to_anchor", [k[0] for k in keys], [k[1] for k in keys], offsets, flat)
        if d["text_id_to_parent"] is not None:
            data.add_key_value_store(tp + ".value_id_to_parent", *csr_from_lists(_multi_store(d["text_id_to_parent"], False)))
        if d["value_id_to_anchor"] is not None:
            data.add_key_value_store(path + ".value_id_to_anchor", *csr_from_lists(_multi_store(d["value_id_to_anchor"], False)))
        if d["parent_to_text_id"] is not None:  # 1:1 direct index: the last write of a key wins
            rows = _multi_store(d["parent_to_text_id"], False)
            data.add_key_value_store(tp + ".parent_to_value_id", *csr_from_lists([r[-1:] for r in rows]))
        if d["text_id_to_anchor"] is not None:
            data.add_key_value_store(tp + ".text_id_to_anchor", *csr_from_lists(_multi_store(d["text_id_to_anchor"], True)))
        if d["anchor_to_text_id"] is not None:
            data.add_key_value_store(tp + ".anchor_to_text_id", *csr_from_lists(_multi_store(d["anchor_to_text_id"], False)))
        if d["boost"] is not None:  # extract_field_name(path) + BOOST_VALID_TO_VALUE; get_value reads the first entry
            rows = _multi_store(d["boost"], False)
            present = np.array([1 if r else 0 for r in rows], np.uint8)
            bits = np.array([r[0] if r else 0 for r in rows], np.uint32)
            data.add_boost(path + ".boost_valid_to_value", bits.view(np.float32), present)
    for path, d in id_rel.items():
        if d["value_to_parent"] is not None:  # 1:1
            rows = _multi_store(d["value_to_parent"], False)
            data.add_key_value_store(path + ".value_id_to_parent", *csr_from_lists([r[-1:] for r in rows]))
        if d["parent_to_value"] is not None:
            data.add_key_value_store(path + ".parent_to_value_id", *csr_from_lists(_multi_store(d["parent_to_value"], False)))
    if token_values is not None:
        # every entry's text is looked up exactly (levenshtein 0, ignore_case false) in the field's dictionary; the value is stored
        # under the term id in "<path>.textindex.token_values.boost_valid_to_value" (a direct single-value boost store)
        entries, tv_path = token_values
        ids = term_id.get(tv_path, {})
        hits = {}
        for e in entries:
            if e.get("value") is not None and e["text"] in ids:
                hits[ids[e["text"]]] = np.float32(e["value"])
        n = (max(hits) + 1) if hits else 0
        vals, present = np.zeros(n, np.float32), np.zeros(n, np.uint8)
        for k, v in hits.items():
            vals[k], present[k] = v, 1
        data.add_boost(tv_path + ".textindex.token_values.boost_valid_to_value", vals, present)
    return data, info
