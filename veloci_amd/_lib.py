"""ctypes binding of the C ABI (include/veloci_amd.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback for the query path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("VQ_LIB") or os.path.join(_HERE, "libveloci_amd.so")
_LIB = None

VQ_OK = 0
ERR_NAMES = {1: "InvalidRequest", 2: "FstNotFound", 3: "IndexNotFound", 4: "Unsupported", 5: "Device", 6: "InvalidArgument", 7: "Json"}

# every symbol include/veloci_amd.h declares
SYMBOLS = [
    "vq_last_error", "vq_index_builder_new", "vq_index_builder_free", "vq_index_add_fst", "vq_index_add_token_to_anchor_score",
    "vq_index_add_key_value_store", "vq_index_add_phrase_pair_to_anchor", "vq_index_add_boost", "vq_index_set_column_meta", "vq_index_build",
    "vq_index_free", "vq_index_set_stream", "vq_index_set_streams", "vq_index_set_allreduce", "vq_index_device_bytes", "vq_request_parse", "vq_request_free", "vq_request_to_json", "vq_request_has_facets", "vq_request_page_after", "vq_result_is_page", "vq_debug_to_lowercase", "vq_debug_normalize_text", "vq_debug_sort_unique_u32", "vq_debug_compile", "vq_result_num_hits",
    "vq_result_execution_time_ns", "vq_result_len", "vq_result_ids", "vq_result_scores", "vq_result_num_facets", "vq_result_facet_field",
    "vq_result_facet_len", "vq_result_facet_value", "vq_result_facet_count", "vq_result_to_json", "vq_result_why_found_terms_json", "vq_result_why_found_info_json", "vq_result_explain_json", "vq_result_free", "vq_search", "vq_search_json",
    "vq_highlight_json", "vq_highlight_text", "vq_suggest_json", "vq_suggest_len", "vq_suggest_text", "vq_suggest_score", "vq_suggest_term_id", "vq_suggest_free",
    "vq_search_batch", "vq_search_batch_flat", "vq_search_batch_partial", "vq_partial_bytes", "vq_partial_device_ptr", "vq_partial_hist_bytes",
    "vq_partial_hist_device_ptr", "vq_merge_partials", "vq_merge_partials_flat", "vq_partial_free",
    "vq_search_batch_partial_at", "vq_partial_slots", "vq_index_partial_arena_ptr", "vq_partial_total_bytes", "vq_merge_partials_flat_strided",
    "vq_comm_unique_id", "vq_comm_init", "vq_comm_init_custom", "vq_comm_destroy", "vq_shard_step_begin", "vq_shard_step_end", "vq_shard_step_free", "vq_shard_step_flat",
    "vq_profile_read", "vq_profile_enable", "vq_profile_json", "vq_debug_div100_mismatches", "vq_debug_facet_select", "vq_index_speculative_reruns", "vq_version",
]
COMM_ID_BYTES = 128
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLREDUCE_U32_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class VelociError(RuntimeError):
    """Mirrors the reference's VelociError (src/error.rs:5-43): `code` is the C-ABI error code."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code
        self.kind = ERR_NAMES.get(code, str(code))


def lib_path():
    return _SO


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise ImportError(f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  veloci_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (SONAME libamdhip64.so.7, same as
    # /opt/rocm's).  Importing torch first makes the loader resolve this library's NEEDED entry to that copy,
    # so torch.distributed (RCCL) and these kernels share devices, streams and memory.  Loading this library
    # first and torch later would put two runtimes in the process and the second one finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_SO)
    vp, cp, u32, u64, sz, i = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_size_t, C.c_int
    sig = {
        "vq_last_error": (cp, []),
        "vq_version": (cp, []),
        "vq_index_builder_new": (vp, [u32, u32, u32]),
        "vq_index_builder_free": (None, [vp]),
        "vq_index_add_fst": (i, [vp, cp, u32, vp, vp]),
        "vq_index_add_token_to_anchor_score": (i, [vp, cp, u32, vp, vp, vp, vp]),
        "vq_index_add_key_value_store": (i, [vp, cp, u32, u32, vp, vp]),
        "vq_index_add_phrase_pair_to_anchor": (i, [vp, cp, u64, vp, vp, vp, vp]),
        "vq_index_add_boost": (i, [vp, cp, u32, u32, vp, vp]),
        "vq_index_set_column_meta": (i, [vp, cp, i, i]),
        "vq_index_build": (i, [vp, i, C.POINTER(vp)]),
        "vq_index_free": (None, [vp]),
        "vq_index_set_stream": (i, [vp, vp]),
        "vq_index_set_streams": (i, [vp, vp, vp]),
        "vq_index_set_allreduce": (i, [vp, vp, vp]),
        "vq_index_device_bytes": (u64, [vp]),
        "vq_request_parse": (i, [cp, sz, C.POINTER(vp)]),
        "vq_request_free": (None, [vp]),
        "vq_request_to_json": (cp, [vp]),
        "vq_request_has_facets": (i, [vp]),
        "vq_request_page_after": (i, [vp, C.c_float, u32, C.POINTER(vp)]),
        "vq_result_is_page": (i, [vp]),
        "vq_debug_to_lowercase": (sz, [cp, sz, cp, sz]),
        "vq_debug_normalize_text": (sz, [cp, sz, cp, sz]),
        "vq_debug_sort_unique_u32": (sz, [vp, sz]),
        "vq_debug_compile": (i, [vp, vp]),
        "vq_result_num_hits": (u64, [vp]),
        "vq_result_execution_time_ns": (u64, [vp]),
        "vq_result_len": (sz, [vp]),
        "vq_result_ids": (C.POINTER(u32), [vp]),
        "vq_result_scores": (C.POINTER(C.c_float), [vp]),
        "vq_result_num_facets": (sz, [vp]),
        "vq_result_facet_field": (cp, [vp, sz]),
        "vq_result_facet_len": (sz, [vp, sz]),
        "vq_result_facet_value": (cp, [vp, sz, sz]),
        "vq_result_facet_count": (u64, [vp, sz, sz]),
        "vq_result_to_json": (cp, [vp]),
        "vq_result_why_found_terms_json": (cp, [vp]),
        "vq_result_why_found_info_json": (cp, [vp]),
        "vq_result_explain_json": (cp, [vp]),
        "vq_suggest_json": (i, [vp, cp, sz, C.POINTER(vp)]),
        "vq_highlight_json": (i, [vp, cp, sz, C.POINTER(vp)]),
        "vq_highlight_text": (sz, [cp, sz, C.POINTER(cp), C.POINTER(sz), sz, cp, sz, i, cp, sz]),
        "vq_suggest_len": (sz, [vp]),
        "vq_suggest_text": (cp, [vp, sz]),
        "vq_suggest_score": (C.c_float, [vp, sz]),
        "vq_suggest_term_id": (u32, [vp, sz]),
        "vq_suggest_free": (None, [vp]),
        "vq_result_free": (None, [vp]),
        "vq_search": (i, [vp, vp, C.POINTER(vp)]),
        "vq_search_json": (i, [vp, cp, sz, C.POINTER(vp)]),
        "vq_search_batch": (i, [vp, C.POINTER(vp), sz, C.POINTER(vp), C.POINTER(i)]),
        "vq_search_batch_flat": (i, [vp, C.POINTER(vp), sz, sz, vp, vp, vp, vp, vp]),
        "vq_search_batch_partial": (i, [vp, C.POINTER(vp), sz, C.POINTER(vp)]),
        "vq_partial_bytes": (sz, [vp]),
        "vq_partial_device_ptr": (vp, [vp]),
        "vq_partial_hist_bytes": (sz, [vp]),
        "vq_partial_hist_device_ptr": (vp, [vp]),
        "vq_profile_json": (cp, [vp, i]),
        "vq_debug_div100_mismatches": (u32, []),
        "vq_debug_facet_select": (i, [vp, u32, u32, u32, vp, vp]),
        "vq_index_speculative_reruns": (u64, [vp]),
        "vq_merge_partials": (i, [vp, vp, vp, u32, C.POINTER(vp), C.POINTER(i)]),
        "vq_merge_partials_flat": (i, [vp, vp, vp, u32, sz, vp, vp, vp, vp, vp]),
        "vq_partial_free": (None, [vp]),
        "vq_search_batch_partial_at": (i, [vp, C.POINTER(vp), sz, i, sz, C.POINTER(vp)]),
        "vq_partial_slots": (i, []),
        "vq_index_partial_arena_ptr": (vp, [vp]),
        "vq_partial_total_bytes": (sz, [vp]),
        "vq_merge_partials_flat_strided": (i, [vp, vp, vp, u32, sz, sz, vp, vp, vp, vp, vp]),
        "vq_comm_unique_id": (i, [vp]),
        "vq_comm_init": (i, [vp, i, i, vp]),
        "vq_comm_init_custom": (i, [vp, i, i, ALLGATHER_FN, ALLREDUCE_U32_FN, vp]),
        "vq_comm_destroy": (i, [vp]),
        "vq_shard_step_begin": (i, [vp, vp, sz, C.POINTER(vp)]),
        "vq_shard_step_end": (i, [vp, sz, vp, vp, vp, vp, vp]),
        "vq_shard_step_free": (None, [vp]),
        "vq_shard_step_flat": (i, [vp, vp, sz, sz, vp, vp, vp, vp, vp]),
        "vq_profile_read": (i, [vp, i, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64)]),
        "vq_profile_enable": (i, [vp, i]),
    }
    for name, (rt, at) in sig.items():
        f = getattr(L, name)
        f.restype = rt
        f.argtypes = at
    _LIB = L
    return L


def check(rc):
    if rc != VQ_OK:
        raise VelociError(rc, lib().vq_last_error().decode("utf-8", "replace"))
