"""Model graph configs (the reference's `models/v8/*.yaml` graphs as Python dict builders)."""
from .v8 import CONFIGS, get_config  # noqa: F401
