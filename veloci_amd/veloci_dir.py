"""Reader for the parts of a veloci index directory whose byte formats live in the reference's own tree.

A veloci index directory (`Persistence`, src/persistence.rs:206-291) holds `metaData.json` (`PeristenceMetaData`, src/metadata.rs:11-44) and
one or two files per index listed there.  This module maps every listed index to the `vq_index_add_*` call that takes its decoded arrays and
decodes the file kinds that are specified by the reference's own sources:

  metaData.json                      serde-JSON of PeristenceMetaData: num_docs, columns{name -> FieldInfo{indices[IndexMetadata], ...}}
  <path>            SingleValue      `SingleArrayPacked` (src/indices/direct/single_array.rs:17-63,135-137): one little-endian integer of
                                     1-4 bytes per key — the width follows `metadata.max_value_id` (get_bytes_required) —, holding value + 1;
                                     0 = the key has no value (EMPTY_BUCKET, src/indices/mod.rs:17-19)
  <path>.indirect   MultiValue       `Indirect` (src/indices/indirect/indirect.rs:24-89, indirect/mod.rs:12-20, create_indirect.rs:69-72): one
                                     little-endian u32 per key — 0 = no values; high bit set = the key's ONE value, inlined (low 31 bits);
                                     otherwise a byte offset into <path>.data

NOT decodable here, because their formats belong to crates that are not under /root/reference: the value lists of keys with two
or more values in `<path>.data` (crate vint32 0.3.0, "common encoding" vint arrays), posting lists `*.to_anchor_id_score.{indirect,data}`
(`TokenToAnchorScoreVint`: the .indirect file is again plain u32 offsets, the .data side is vint32), phrase-pair tables
(`IndirectIMBinarySearch`, vint32 value lists) and the term dictionaries `*.fst` (crate fst 0.4.7).  `plan()` reports every index of a directory
with what it maps to and whether its bytes can be read; `load()` fills an `IndexData` with everything that can.  A Rust-side exporter that
hands over the decoded arrays covers the rest.
"""
import json
import os

import numpy as np

from .index import IndexData

HIGH_BIT = 1 << 31

# IndexCategory / IndexCardinality -> the builder call that takes the index (persistence.rs:206-291 builds the same maps)
ADDERS = {
    "AnchorScore": "vq_index_add_token_to_anchor_score",
    "Phrase": "vq_index_add_phrase_pair_to_anchor",
    "Boost": "vq_index_add_boost",
    "KeyValue": "vq_index_add_key_value_store",
}


def bytes_required(max_value_id):
    """get_bytes_required (single_array.rs:17-28): `val += val` — the width is taken from TWICE the largest value"""
    v = (int(max_value_id) * 2) & 0xFFFFFFFF
    return 1 if v < (1 << 8) else 2 if v < (1 << 16) else 3 if v < (1 << 24) else 4


def decode_single_array(raw, max_value_id):
    """SingleArrayPacked -> (present bool[n], values u32[n]); decode_bit_packed_val (single_array.rs:47-63): stored = value + 1, 0 = empty"""
    w = bytes_required(max_value_id)
    raw = np.frombuffer(raw, np.uint8)
    n = len(raw) // w
    padded = np.zeros((n, 4), np.uint8)
    padded[:, :w] = raw[:n * w].reshape(n, w)
    stored = padded.view("<u4").reshape(n)
    present = stored != 0
    return present, np.where(present, stored - 1, 0).astype(np.uint32)


def encode_single_array(values, present, max_value_id):
    """encode_vals (single_array.rs:31-44) of create_direct's cache (value + 1, create_direct.rs:46-49): fixtures for the tests"""
    w = bytes_required(max_value_id)
    stored = np.where(present, np.asarray(values, np.uint64) + 1, 0).astype("<u4")
    return stored.view(np.uint8).reshape(-1, 4)[:, :w].tobytes()


def decode_indirect_heads(raw):
    """Indirect's .indirect file -> (kind u8[n]: 0 empty / 1 one inlined value / 2 offset into .data, value u32[n]: the value or the offset)"""
    heads = np.frombuffer(raw, "<u4")
    kind = np.where(heads == 0, 0, np.where(heads & HIGH_BIT, 1, 2)).astype(np.uint8)
    return kind, (heads & (HIGH_BIT - 1)).astype(np.uint32)


def encode_indirect_inline(lists):
    """a store whose keys all hold at most one value, as create_indirect.rs:60-72 writes it (no .data bytes are needed): fixtures for the tests"""
    heads = np.zeros(len(lists), "<u4")
    for k, vs in enumerate(lists):
        if len(vs) > 1:
            raise ValueError("keys with several values go to the .data file (vint32): not written here")
        if len(vs) == 1:
            heads[k] = int(vs[0]) | HIGH_BIT
    return heads.tobytes()


def read_metadata(directory):
    with open(os.path.join(directory, "metaData.json"), "rb") as f:
        meta = json.loads(f.read())
    if "columns" not in meta or "num_docs" not in meta:
        raise ValueError("metaData.json: not a PeristenceMetaData (num_docs / columns missing)")
    return meta


def plan(directory):
    """-> (metadata, [entry per index]); entry: path, column, category, cardinality, adder (the vq_index_add_* that takes it), files,
    readable: "yes" | "keys with one value only" | "no: <which crate's format>" """
    meta = read_metadata(directory)
    entries = []
    for col, info in sorted(meta["columns"].items()):
        for ix in info.get("indices", []):
            cat, card = ix.get("index_category", "KeyValue"), ix.get("index_cardinality", "MultiValue")
            path = ix["path"]
            e = {"path": path, "column": col, "category": cat, "cardinality": card, "adder": ADDERS[cat], "is_empty": bool(ix.get("is_empty", False)),
                 "max_value_id": int(ix.get("metadata", {}).get("max_value_id", 0))}
            if cat == "AnchorScore":
                e["files"], e["readable"] = [path + ".indirect", path + ".data"], "no: posting bytes are vint32 arrays (TokenToAnchorScoreVint, crate vint32 0.3.0)"
            elif cat == "Phrase":
                e["files"], e["readable"] = [path + ".indirect", path + ".data"], "no: value lists are vint32 arrays (IndirectIMBinarySearch, crate vint32 0.3.0)"
            elif card == "SingleValue":
                e["files"], e["readable"] = [path], "yes"
            else:
                e["files"], e["readable"] = [path + ".indirect", path + ".data"], "keys with one value only"
            if e["is_empty"]:
                e["files"], e["readable"] = [], "yes"
            entries.append(e)
        if info.get("has_fst"):
            entries.append({"path": col + ".textindex", "column": col, "category": "Fst", "cardinality": "-", "adder": "vq_index_add_fst", "is_empty": False, "max_value_id": 0,
                            "files": [col + ".textindex.fst"], "readable": "no: crate fst 0.4.7's automaton file"})
    return meta, entries


def load(directory, data=None):
    """Fill an IndexData with every index of the directory whose bytes can be read -> (IndexData, report).  report: list of the plan's entries
    with `loaded` (bool) and, where not, `why`.  Column facts (is_anchor_identity_column, the tokenize option) are taken over for every column."""
    meta, entries = plan(directory)
    if data is None:
        data = IndexData(int(meta["num_docs"]))
    for col, info in meta["columns"].items():
        tok = bool(info.get("textindex_metadata", {}).get("options", {}).get("tokenize", True))
        data.set_column_meta(col, bool(info.get("is_anchor_identity_column", False)), tok)
    for e in entries:
        e["loaded"] = False
        if e["category"] in ("AnchorScore", "Phrase", "Fst"):
            e["why"] = e["readable"]
            continue
        if e["is_empty"]:
            if e["category"] == "KeyValue":
                data.add_key_value_store(e["path"], np.zeros(1, np.uint64), np.zeros(0, np.uint32))
                e["loaded"] = True
            continue
        missing = [f for f in e["files"][:1] if not os.path.exists(os.path.join(directory, f))]
        if missing:
            e["why"] = "file missing: " + missing[0]
            continue
        with open(os.path.join(directory, e["files"][0]), "rb") as f:
            raw = f.read()
        if e["cardinality"] == "SingleValue":
            present, values = decode_single_array(raw, e["max_value_id"])
            if e["category"] == "Boost":  # boost_valueid_to_value: the stored u32 is the bit pattern of the f32 boost value (boost.rs:490-494)
                data.boost[e["path"]] = (0, present.astype(np.uint8), values)
            else:
                offsets = np.zeros(len(present) + 1, np.uint64)
                offsets[1:] = np.cumsum(present)
                data.add_key_value_store(e["path"], offsets, values[present])
            e["loaded"] = True
        else:
            kind, val = decode_indirect_heads(raw)
            if (kind == 2).any():
                e["why"] = f"{int((kind == 2).sum())} of {len(kind)} keys hold several values: their lists are vint32 arrays in {e['files'][1]}"
                continue
            if e["category"] == "Boost":
                e["why"] = "1:n boost values without a multi-valued key: nothing the reference writes this way"
                continue
            has = kind == 1
            offsets = np.zeros(len(kind) + 1, np.uint64)
            offsets[1:] = np.cumsum(has)
            data.add_key_value_store(e["path"], offsets, val[has])
            e["loaded"] = True
    return data, entries


def write_fixture_directory(directory, data, num_docs):
    """TEST FIXTURES: write the parts of `data` (an IndexData, e.g. the mini-indexer's) that this module can read back in the reference's
    layout — metaData.json listing EVERY index of `data` under its category and cardinality (create.rs:827-871), the SingleValue files and
    the .indirect files of MultiValue stores whose keys hold at most one value.  Stores with multi-valued keys, posting lists, phrase tables
    and dictionaries are listed but their files are not written (their byte formats are not the tree's)."""
    os.makedirs(directory, exist_ok=True)
    columns = {}

    def col_of(path):
        for suffix in (".textindex", ".value_id_to_parent", ".parent_to_value_id", ".boost_valid_to_value", ".value_id_to_anchor", ".anchor_to_text_id"):
            k = path.find(suffix)
            if k >= 0:
                return path[:k]
        return path

    def entry(col):
        ident, tok = data.columns.get(col, (False, True))
        return columns.setdefault(col, {"name": col, "textindex_metadata": {"num_text_ids": 0, "num_long_text_ids": 0, "options": {"tokenize": bool(tok), "tokenize_on_chars": None,
                                                                           "do_not_store_text_longer_than": 32}},
                                        "indices": [], "is_anchor_identity_column": bool(ident), "has_fst": False})

    written = []
    for path, (key_base, offsets, values) in sorted(data.key_value_stores.items()):
        lens = np.diff(offsets.astype(np.int64))
        maxv = int(values.max()) if len(values) else 0
        single = bool((lens <= 1).all()) and path.endswith((".parent_to_value_id", ".value_id_to_parent"))  # (the reference's 1:1 direct stores)
        md = {"path": path, "index_category": "KeyValue", "index_cardinality": "SingleValue" if single else "MultiValue", "is_empty": len(lens) == 0,
              "metadata": {"max_value_id": maxv, "avg_join_size": float(lens.mean()) if len(lens) else 0.0, "num_values": int(len(lens)), "num_ids": int(len(values))}, "data_type": "U32"}
        entry(col_of(path))["indices"].append(md)
        if len(lens) == 0 or key_base:
            continue
        lists = [values[int(offsets[k]):int(offsets[k + 1])] for k in range(len(lens))]
        if single:
            present = lens == 1
            dense = np.zeros(len(lens), np.uint32)
            dense[present] = values
            with open(os.path.join(directory, path), "wb") as f:
                f.write(encode_single_array(dense, present, maxv))
            written.append(path)
        elif (lens <= 1).all():
            with open(os.path.join(directory, path + ".indirect"), "wb") as f:
                f.write(encode_indirect_inline(lists))
            open(os.path.join(directory, path + ".data"), "wb").close()
            written.append(path)
    for path, (key_base, present, bits) in sorted(data.boost.items()):
        pres = np.ones(len(bits), bool) if present is None else np.asarray(present, bool)
        maxv = int(bits[pres].max()) if pres.any() else 0
        md = {"path": path, "index_category": "Boost", "index_cardinality": "SingleValue", "is_empty": False,
              "metadata": {"max_value_id": maxv, "avg_join_size": 1.0, "num_values": int(pres.sum()), "num_ids": int(pres.sum())}, "data_type": "U32"}
        entry(col_of(path))["indices"].append(md)
        if not key_base:
            with open(os.path.join(directory, path), "wb") as f:
                f.write(encode_single_array(bits, pres, maxv))
            written.append(path)
    for path, (offsets, anchors, scores, _) in sorted(data.token_to_anchor_score.items()):
        entry(col_of(path))["indices"].append({"path": path, "index_category": "AnchorScore", "index_cardinality": "MultiValue", "is_empty": False,
                                               "metadata": {"max_value_id": int(anchors.max()) if len(anchors) else 0, "avg_join_size": 0.0, "num_values": int(len(offsets) - 1),
                                                            "num_ids": int(len(anchors))}, "data_type": "U32"})
    for path, (t1, t2, offsets, anchors) in sorted(data.phrase_pair_to_anchor.items()):
        entry(col_of(path))["indices"].append({"path": path, "index_category": "Phrase", "index_cardinality": "MultiValue", "is_empty": len(t1) == 0,
                                               "metadata": {"max_value_id": int(anchors.max()) if len(anchors) else 0, "avg_join_size": 0.0, "num_values": int(len(t1)),
                                                            "num_ids": int(len(anchors))}, "data_type": "U32"})
    for path in data.fst:
        entry(path[:-len(".textindex")] if path.endswith(".textindex") else path)["has_fst"] = True
    for col in data.columns:
        entry(col)
    with open(os.path.join(directory, "metaData.json"), "w") as f:
        json.dump({"num_docs": int(num_docs), "bytes_indexed": 0, "columns": columns}, f)
    return written
