from .get_model import get_model  # noqa: F401
from .pwclite import PWCLite  # noqa: F401
from .pwclite_uflow import PWCLiteUflow  # noqa: F401
from .uflow_model import PWCFlow  # noqa: F401
