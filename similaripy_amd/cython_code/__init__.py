"""Name-compatible home of the one helper the reference's own tests reach into (`similaripy.cython_code.utils.get_num_threads`,
tests/test_similarity.py:384-390).  Nothing here is Cython: the compute lives in libsimilaripy_hip.so."""
from . import utils  # noqa: F401
