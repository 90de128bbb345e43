from .droid_transformer import DenseNetwork, FullCrossAttentionEncoder, FullTransformerEncoder, MLPBlock  # noqa: F401
from .epic import EPiC_encoder, EPiC_layer  # noqa: F401
from .mdma import MDMA, Block  # noqa: F401
from .losses import ConditionalFlowMatchingLoss, DroidLoss, FlowMatchingLoss  # noqa: F401
from .time_emb import CosineEncoding, cosine_encoding  # noqa: F401
