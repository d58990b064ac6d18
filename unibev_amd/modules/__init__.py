from .bricks import (FFN, BaseModule, BaseTransformerLayer, DetrTransformerDecoderLayer,  # noqa
                     LearnedPositionalEncoding, MultiheadAttention, TransformerLayerSequence)
from .decoder import DetectionTransformerDecoder  # noqa
from .deform_attn import (CustomMSDeformableAttention, MSDeformableAttention3DImg,  # noqa
                          MSDeformableAttention3DPts, MultiScaleDeformableAttention)
from .head import UniBEV_Head, inverse_sigmoid  # noqa
from .encoders import ImgEncoder, ImgLayer, PtsEncoder, PtsLayer  # noqa
from .sca import SpatialCrossAttentionImg, SpatialCrossAttentionPts  # noqa
from .transformer import UniBEVTransformer  # noqa
from .voxel import HardSimpleVFE, Voxelization, extract_pts_feat, sparse_to_dense, voxelize_batch  # noqa
from .sparse_encoder import (SparseBasicBlock, SparseConv3d, SparseConvTensor, SparseEncoder,  # noqa
                             SparseSequential, SubMConv3d, make_sparse_convmodule)
from .grid_mask import GridMask  # noqa
from .backbones import (FPN, SECOND, SECONDFPN, ModulatedDeformConv2dPack, ResNet,  # noqa
                        extract_img_feat)
from .detector import UniBEV  # noqa
