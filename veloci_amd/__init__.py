"""veloci_amd — MI355X-native query execution for the veloci search engine.

Only the hot path: search::Request -> search::SearchResult, executed by hand-written gfx950 kernels
behind the C ABI of include/veloci_amd.h.  Importing this package does not touch the GPU; the HIP
library is loaded on first use and there is no CPU fallback.
"""
from ._lib import VelociError, lib, lib_path  # noqa: F401
from .index import Index, IndexData, csr_from_lists  # noqa: F401
from .search import Hit, PartialBatch, Request, RequestBatch, SearchResult, highlight, highlight_text, search, search_batch, search_batch_flat, suggest  # noqa: F401

__all__ = ["VelociError", "lib", "lib_path", "Index", "IndexData", "csr_from_lists", "Hit", "PartialBatch", "Request", "RequestBatch", "SearchResult", "search",
           "search_batch", "search_batch_flat", "suggest", "highlight", "highlight_text"]
