from .dist import RayShardedDP, free_port, init_from_env, launched_by_torchrun, shard_slice, spawn_ranks  # noqa: F401
