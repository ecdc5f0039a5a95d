from focusflow_official_amd.update_block import BasicMotionEncoder, BasicUpdateBlock, FlowHead, SepConvGRU  # noqa: F401
