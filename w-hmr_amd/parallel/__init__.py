from .grad_reducer import GradReducer, shard_batch, broadcast_buffers      # noqa: F401
from .sync_bn import SyncGroup, convert_sync_batchnorm, revert_sync_batchnorm      # noqa: F401
