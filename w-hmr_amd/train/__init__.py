from .vit_autograd import ViTFn, vit_forward_train, vit_backward      # noqa: F401
from .deconv_autograd import DeconvBNReLUFn, deconv_forward_train, deconv_backward      # noqa: F401
from .smpl_autograd import SMPLFn, smpl_forward_train, smpl_backward      # noqa: F401
from .maf_autograd import MAFSampleFn      # noqa: F401
