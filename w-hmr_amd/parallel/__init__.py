from .grad_reducer import GradReducer, shard_batch      # noqa: F401
