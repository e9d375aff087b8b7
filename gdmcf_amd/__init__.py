"""gdmcf_amd -- MI355X-native (gfx950) hot path of GDMCF: the per-batch forward/reverse Gaussian
diffusion path behind the reference's own Python API.  See DESIGN.md / include/gdmcf_hip.h."""
from . import _lib  # noqa: F401
from .DNN import DNN, timestep_embedding  # noqa: F401
from .gaussian_diffusion import GaussianDiffusion, GaussianDiffusionDiscrete, ModelMeanType  # noqa: F401
from .optim import FusedAdamW  # noqa: F401
from .evaluate_utils import computeTopNAccuracy, computeTopNAccuracy_device, masked_topk, print_results  # noqa: F401
from .lightgcn import LightGCN  # noqa: F401
from .onehot import DNNOneHot  # noqa: F401
from .onehot_embedding import DNNOneHotEmbedding  # noqa: F401
from .onehot_gcn import DNNOneHotEmbeddingGCN  # noqa: F401
from . import checkpoint, data_utils, driver, parallel  # noqa: F401

__all__ = ["DNN", "timestep_embedding", "GaussianDiffusion", "GaussianDiffusionDiscrete", "ModelMeanType", "FusedAdamW", "computeTopNAccuracy",
           "computeTopNAccuracy_device", "masked_topk", "print_results", "LightGCN", "DNNOneHot", "DNNOneHotEmbedding", "DNNOneHotEmbeddingGCN"]
