"""TEST INFRASTRUCTURE — an independent restatement of how serde deserialises the reference's `search::Request` from JSON
(src/search/request/mod.rs:15-87, search_request.rs:6-23,102-179, boost_request.rs:4-33, facet_request.rs:2-11), written against Python's own
JSON parser: neither the product's parser (veloci_amd/csrc/json.hpp, request.hpp) nor anything under oracle/ is used.  `canonical(text)`
renders what was understood in the layout of `vq_request_to_json` (include/veloci_amd.h): declaration order, absent Options as null, `select`
and `snippet_info` as "present" booleans, f32 values as their bit patterns.  `ParseError` stands for any serde_json error.

serde rules restated: unknown keys are ignored (no deny_unknown_fields); a missing Option field is None, a null one too; `#[serde(default)]`
bools default to false when MISSING but reject null; `top` (Request, FacetRequest) defaults to Some(10) when missing and is None when null
(default = "default_top"); usize / u32 accept only non-negative integers without fraction or exponent; f32 goes through f64 (`as f32`);
an externally tagged enum is a map with exactly one key; a duplicate key of a struct is an error."""
import json
import struct


class ParseError(Exception):
    pass


def _no_duplicates(pairs):
    keys = [k for k, _ in pairs]
    if len(set(keys)) != len(keys):
        raise ParseError("duplicate field")
    return dict(pairs)


class _Int(int):
    """an integer literal (no fraction, no exponent)"""


def _loads(text):
    try:
        return json.loads(text, object_pairs_hook=_no_duplicates, parse_int=_Int, parse_constant=lambda c: (_ for _ in ()).throw(ParseError(c)))
    except ParseError:
        raise
    except (ValueError, RecursionError) as e:
        raise ParseError(str(e))


def _f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def _str(v):
    if not isinstance(v, str):
        raise ParseError("expected a string")
    if any(0xD800 <= ord(c) <= 0xDFFF for c in v):  # Python keeps a lone \uD83D; serde_json: "lone leading surrogate in hex escape"
        raise ParseError("lone surrogate")
    return v


def _bool(v):
    if not isinstance(v, bool):
        raise ParseError("expected a boolean")
    return v


def _uint(v, bits=64):
    if isinstance(v, bool) or not isinstance(v, _Int) or v < 0 or v >= 1 << bits:
        raise ParseError("expected an unsigned integer")
    return int(v)


def _f32(v):
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        raise ParseError("expected a number")
    try:
        return _f32_bits(float(v))
    except OverflowError:  # beyond f32: `as f32` saturates to infinity
        return _f32_bits(float("inf") if v > 0 else float("-inf"))


def _obj(v):
    if not isinstance(v, dict):
        raise ParseError("expected a map")
    return v


def _seq(v, f):
    if not isinstance(v, list):
        raise ParseError("expected a sequence")
    return [f(e) for e in v]


def _opt(d, key, f):
    v = d.get(key)
    return None if v is None else f(v)


def _req(d, key, f):
    if key not in d:
        raise ParseError("missing field `%s`" % key)
    return f(d[key])


def _default_false(d, key):
    return _bool(d[key]) if key in d else False


def boost_part(v):
    d = _obj(v)
    fun = _opt(d, "boost_fun", _str)
    if fun is not None and fun not in ("Log2", "Log10", "Multiply", "Add", "Replace"):
        raise ParseError("unknown variant")
    return {"path": _req(d, "path", _str), "boost_fun": fun, "param": _opt(d, "param", _f32),
            "skip_when_score": _opt(d, "skip_when_score", lambda s: _seq(s, _f32)), "expression": _opt(d, "expression", _str)}


def options(v):
    d = _obj(v)
    return {"explain": _default_false(d, "explain"), "top": _opt(d, "top", _uint), "skip": _opt(d, "skip", _uint),
            "boost": _opt(d, "boost", lambda s: _seq(s, boost_part))}


def snippet_info(v):
    """src/search/request/snippet_info.rs:1-13: every field has a serde default, so a missing key is fine and a null is not.  The canonical dump only
    records that the object was there."""
    d = _obj(v)
    if "num_words_around_snippet" in d:
        n = d["num_words_around_snippet"]
        if isinstance(n, bool) or not isinstance(n, _Int) or not -(1 << 63) <= n < (1 << 63):
            raise ParseError("expected an i64")
    for key in ("snippet_start_tag", "snippet_end_tag", "snippet_connector"):
        if key in d:
            _str(d[key])
    if "max_snippets" in d:
        _uint(d["max_snippets"], 32)
    return True


def search_part(v):
    d = _obj(v)
    return {"path": _req(d, "path", _str), "terms": _req(d, "terms", lambda s: _seq(s, _str)),
            "levenshtein_distance": _opt(d, "levenshtein_distance", lambda x: _uint(x, 32)),
            "starts_with": _default_false(d, "starts_with"), "is_regex": _default_false(d, "is_regex"),
            "token_value": _opt(d, "token_value", boost_part), "boost": _opt(d, "boost", _f32), "ignore_case": _opt(d, "ignore_case", _bool),
            "snippet": _opt(d, "snippet", _bool), "snippet_info": bool(_opt(d, "snippet_info", snippet_info)),
            "top": _opt(d, "top", _uint), "skip": _opt(d, "skip", _uint), "options": _opt(d, "options", options)}


def search_request(v):
    d = _obj(v)
    if len(d) != 1:
        raise ParseError("expected a map with a single key")
    (tag, body), = d.items()
    if tag == "search":
        return {"search": search_part(body)}
    if tag not in ("or", "and"):
        raise ParseError("unknown variant")
    b = _obj(body)
    return {tag: {"queries": _req(b, "queries", lambda s: _seq(s, search_request)), "options": _opt(b, "options", options)}}


def facet(v):
    d = _obj(v)
    return {"field": _req(d, "field", _str), "top": _opt(d, "top", _uint) if "top" in d else 10}


def phrase_boost(v):
    d = _obj(v)
    return {"search1": _req(d, "search1", search_part), "search2": _req(d, "search2", search_part)}


def request(v):
    d = _obj(v)
    select = _opt(d, "select", lambda s: _seq(s, _str))
    return {"search_req": _opt(d, "search_req", search_request), "suggest": _opt(d, "suggest", lambda s: _seq(s, search_part)),
            "boost": _opt(d, "boost", lambda s: _seq(s, boost_part)), "boost_term": _opt(d, "boost_term", lambda s: _seq(s, search_part)),
            "facets": _opt(d, "facets", lambda s: _seq(s, facet)), "phrase_boosts": _opt(d, "phrase_boosts", lambda s: _seq(s, phrase_boost)),
            "select": select is not None, "filter": _opt(d, "filter", search_request),
            "top": _opt(d, "top", _uint) if "top" in d else 10, "skip": _opt(d, "skip", _uint),
            "why_found": _default_false(d, "why_found"), "text_locality": _default_false(d, "text_locality"), "explain": _default_false(d, "explain")}


def canonical(text):
    """-> the canonical dump (str); raises ParseError where serde_json would return an error"""
    return json.dumps(request(_loads(text)), ensure_ascii=False, separators=(",", ":"))
