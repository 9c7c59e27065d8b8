"""MI355X-native hot path of gauravchak/two_tower_models: the reference's nn.Module
API over hand-written gfx950 HIP kernels (libtt_hotpath.so, C ABI in include/)."""
from . import parallel
from .baseline_mips_module import BaselineMIPSModule
from .graphs import GraphedTrainStep
from .optim import DenseExactAdam
from .two_tower_base_retrieval import TwoTowerBaseRetrieval
from .two_tower_with_debiasing import TwoTowerWithDebiasing
from .two_tower_with_position_debiased_weights import TwoTowerWithPositionDebiasedWeights
from .two_tower_with_user_debiased_weights import TwoTowerWithUserDebiasedWeights
from .two_tower_with_user_history_encoder import TwoTowerWithUserHistoryEncoder
from .user_history_encoder import UserHistoryEncoder

__all__ = [
    "BaselineMIPSModule", "DenseExactAdam", "GraphedTrainStep", "TwoTowerBaseRetrieval", "TwoTowerWithDebiasing",
    "TwoTowerWithPositionDebiasedWeights", "TwoTowerWithUserDebiasedWeights", "TwoTowerWithUserHistoryEncoder",
    "UserHistoryEncoder", "parallel",
]
