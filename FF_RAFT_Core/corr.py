from focusflow_official_amd.corr_block import CorrBlock  # noqa: F401
