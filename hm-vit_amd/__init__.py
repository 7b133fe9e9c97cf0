"""hm-vit_amd: MI355X-native HM-ViT fusion hot path (HIP kernels behind the reference's
fuse-module API).  Import as ``hmvit_amd`` (see hmvit_amd.py at the repository root: the
directory name carries a hyphen)."""
from . import _lib  # noqa: F401  (raises ImportError when libhmvit.so is not built)
from .fusion import HeteroFusion, HeteroFusionBlock  # noqa: F401
from .pointpillar import PointPillar  # noqa: F401
from .decoder import HeteroDecoder, NaiveCompressor  # noqa: F401
from .model import BevformerPointPillarHetero  # noqa: F401
from .postprocess import VoxelPostprocessor, quad_iou, caluclate_tp_fp, calculate_ap, voc_ap  # noqa: F401
from .voxelizer import SpVoxelPreprocessor  # noqa: F401
from .cvt import BEVEmbedding, CrossAttention, CrossViewAttention  # noqa: F401
from .camera import ResnetEncoder, CrossViewModule, CvtCameraEncoder  # noqa: F401
from .fax import CrossViewSwapAttention, FAXModule, FaxCameraEncoder  # noqa: F401
from .train import PointPillarLoss, make_optimizer, train_step  # noqa: F401  (training: loss / optimiser / one step; loop in .trainer)

# the precision mode that is held to the reference's own fp32 tolerance (1e-4) and that bench.py reports as the headline
REFERENCE_PRECISION = "split"

__all__ = ["HeteroFusion", "HeteroFusionBlock", "PointPillar", "HeteroDecoder", "NaiveCompressor", "BevformerPointPillarHetero",
           "VoxelPostprocessor", "quad_iou", "caluclate_tp_fp", "calculate_ap", "voc_ap", "SpVoxelPreprocessor", "BEVEmbedding", "CrossAttention", "CrossViewAttention",
           "ResnetEncoder", "CrossViewModule", "CvtCameraEncoder", "CrossViewSwapAttention", "FAXModule", "FaxCameraEncoder",
           "PointPillarLoss", "make_optimizer", "train_step"]
