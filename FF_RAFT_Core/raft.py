from focusflow_official_amd.raft_net import RAFT  # noqa: F401
