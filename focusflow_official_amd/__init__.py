"""focusflow_official_amd — MI355X-native FF-RAFT hot path behind the reference's module API."""
from .model import FF_RAFT_FUSION  # noqa: F401
from .raft_net import RAFT  # noqa: F401
from .corr_block import CorrBlock  # noqa: F401
from .update_block import BasicUpdateBlock  # noqa: F401
from .cce import BasicParallelFusionLayer  # noqa: F401
