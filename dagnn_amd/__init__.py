"""dagnn_amd - MI355X-native (gfx950) implementation of DAGNN's layer-by-layer message-passing path.

Drop-in modules with the reference's constructor / state_dict / forward(G) contracts:

    from dagnn_amd import DAGNN, ASTNodeEncoder        # ogbg-code/model/dagnn.py, ogbg-code/utils.py
    from dagnn_amd import DAGNN_NA, DAGNN_BN           # dvae/dagnn.py (DAGNN), dvae/dagnn_bn.py
    from dagnn_amd import DataParallel                 # ogbg-code/tg/data_parallel.py (list[Batch] caller)

The hot path runs in libdagnn_hip.so (hand-written HIP, C ABI in include/dagnn_hip.h); importing
this package does not need a GPU, calling `forward` does.
"""
from .constants import *  # noqa: F401,F403
from .data import (GraphBatch, GraphData, augment_edge2, collate_sharded, collate_with_plan,  # noqa: F401
                   shard_by_nodes)
from .host_plan import attach_plan, build_plan_host  # noqa: F401
from .dvae import DAGNN_BN, DAGNN_NA  # noqa: F401
from .model import DAGNN, ASTNodeEncoder  # noqa: F401
from .train import GradBucket  # noqa: F401
from .data_parallel import DataParallel  # noqa: F401

__version__ = "0.1.0"
