from focusflow_official_amd.model import FF_RAFT_FUSION  # noqa: F401
