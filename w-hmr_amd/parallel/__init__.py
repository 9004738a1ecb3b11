from .grad_reducer import GradReducer, shard_batch, broadcast_buffers      # noqa: F401
