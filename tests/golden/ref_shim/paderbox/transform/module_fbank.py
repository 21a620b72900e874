"""Stand-in for paderbox.transform.module_fbank (absent third-party dep): delegates to the oracle's
restatement.  Filterbank parity to the real paderbox is UNPINNED (oracle/features_np.py header)."""
from oracle.features_np import get_fbanks, hz2mel, mel2hz  # noqa: F401
