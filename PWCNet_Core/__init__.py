"""Import-path shim: lets the reference's core/models/ff-pwcnet/train.py / evaluate.py
(`from PWCNet_Core.ff_pwcnet import FF_PWCNET`, train.py:19) pick up the MI355X path unchanged.  See INTEGRATION.md."""
