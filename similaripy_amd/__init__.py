"""similaripy_amd — MI355X-native top-k sparse similarity, drop-in for similaripy's hot path.

Public surface mirrors similaripy/__init__.py:8-36.
Compute happens only in libsimilaripy_hip.so (hand-written HIP for gfx950); importing the
package needs no GPU, calling a similarity function does.
"""
from . import cython_code  # noqa: F401  (get_num_threads, where the reference's tests look for it)
from . import multi_gpu  # noqa: F401  (one process -> one worker per GPU)
from .normalization import bm25, bm25plus, normalize, tfidf
from .similarity import (
    asymmetric_cosine,
    cosine,
    dice,
    dot_product,
    jaccard,
    p3alpha,
    rp3beta,
    s_plus,
    tversky,
)



def device_count() -> int:
    """Usable HIP devices (0 when there is none)."""
    from . import _abi
    return _abi.device_count()


def get_num_threads() -> int:
    """Counterpart of ``similaripy.cython_code.utils.get_num_threads`` (utils.pyx:18-25: omp_get_max_threads(), the width of the
    reference's row loop): the number of GPUs one call can shard its target rows over (sp_device_count, include/sp_knn.h)."""
    return device_count()


def device_cache_trim() -> int:
    """Give the device buffers the library keeps between host-mode calls (operands, outputs, workspace; at most
    SIMILARIPY_AMD_DEVICE_CACHE_MB, default 16 GiB, per GPU) back to the driver.  Returns the bytes released."""
    from . import _abi
    return _abi.device_cache_trim()


__version__ = "0.1.0"

__all__ = [
    "__version__",
    "normalize",
    "bm25",
    "bm25plus",
    "tfidf",
    "dot_product",
    "cosine",
    "asymmetric_cosine",
    "jaccard",
    "dice",
    "tversky",
    "p3alpha",
    "rp3beta",
    "s_plus",
    "device_cache_trim",
    "device_count",
    "get_num_threads",
]
