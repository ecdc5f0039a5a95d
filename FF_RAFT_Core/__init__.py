"""Import-path shim: lets the reference's train.py / evaluate.py (`from FF_RAFT_Core.ff_raft import
FF_RAFT_FUSION`, train.py:19) pick up the MI355X path unchanged.  See INTEGRATION.md."""
