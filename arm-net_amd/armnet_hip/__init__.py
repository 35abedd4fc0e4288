"""armnet_hip — Python binding of the MI355X-native ARM-Net hot path (C ABI: include/armnet_hip.h)."""
from . import native  # noqa: F401
from .block import ArmBlockParams, arm_block_forward, embedding_forward, entmax_forward  # noqa: F401
from .sharded import HipShardOps, RowShardedTable, shard_rows, sharded_arm_block  # noqa: F401
