"""`get_num_threads` — utils.pyx:18-25 returns omp_get_max_threads(), the width of the reference's parallel row loop
(s_plus.h:313, 337).  Here that loop is sharded over GPUs: the answer is the number of usable HIP devices (sp_device_count)."""
from __future__ import annotations


def get_num_threads() -> int:
    from .. import _abi
    return _abi.device_count()
