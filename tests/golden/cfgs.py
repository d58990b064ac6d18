"""Config builders live in the package (``unibev_amd/configs.py``); re-exported for the golden
generator and the tests."""
from unibev_amd.configs import PC_RANGE, decoder_cfg, head_cfg, transformer_cfg  # noqa: F401
