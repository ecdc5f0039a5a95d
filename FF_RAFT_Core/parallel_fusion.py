from focusflow_official_amd.cce import BasicParallelFusionLayer, FusionUnit  # noqa: F401
