"""flex_amd -- MI355X-native SpMM engine behind guohaoqiang/Flex's bench loop.

The product is the C-ABI library ``flex_amd/lib/libflex_spmm.so`` (sources in
``flex_amd/csrc``, declarations in ``include/flex_spmm.h``) plus the C++ host mirror
of the reference's DataLoader/Mat/run interface (``flex_amd/csrc/host``).  This Python
package is only the ctypes binding that tests and bench.py drive it through; it has
no compute path of its own and raises if the HIP library is missing.
"""
from .binding import (  # noqa: F401
    FLEX_ORDER_NATURAL,
    FLEX_ORDER_RCM,
    FLEX_ORDER_CLUSTER,
    FLEX_ORDER_GORDER,
    FLEX_PLAN_STATS,
    FLEX_PLAN_AUTOTUNE,
    FLEX_PLAN_ROW_RANGE,
    FLEX_PLAN_XCD_INTERLEAVE,
    FlexError,
    HostCsr,
    Plan,
    build,
    csv_load,
    csv_save,
    csr_load_bin,
    csr_save_bin,
    csr_fingerprint,
    hbm_probe,
    perm_load,
    perm_save,
    mtx_load,
    fill_dense_rand,
    gather_rows,
    lib,
    lib_path,
    order_rcm,
    order_cluster,
    order_deg,
    order_dfs,
    order_gorder,
    order_rabbit,
    synth_preset,
    perm_csr,
    set_host_threads,
    shard_rows,
    synth_graph,
    SYNTH_PRESETS,
)
from .multigpu import RowShard, broadcast_dense, make_shard  # noqa: F401
