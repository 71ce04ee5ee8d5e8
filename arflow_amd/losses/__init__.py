from .get_loss import get_loss  # noqa: F401
from .uflow_loss import UFlowLoss  # noqa: F401
from .flow_loss import unFlowLoss  # noqa: F401
from .fullres_loss import FullResLoss  # noqa: F401
from .mv_loss import MvLoss  # noqa: F401
