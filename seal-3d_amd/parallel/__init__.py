from .dist import RayShardedDP, init_from_env, shard_slice  # noqa: F401
