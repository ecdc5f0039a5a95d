"""`from PWCNet_Core.ff_pwcnet import FF_PWCNET` (core/models/ff-pwcnet/train.py:19, evaluate.py) -> the HIP module."""
from focusflow_official_amd.pwcnet import FF_PWCNET, Decoder, Extractor, Refiner  # noqa: F401
from focusflow_official_amd.cce import FusionUnit  # noqa: F401
