from .vit_autograd import ViTFn, vit_forward_train, vit_backward      # noqa: F401
from .deconv_autograd import DeconvBNReLUFn, deconv_forward_train, deconv_backward      # noqa: F401
from .smpl_autograd import SMPLFn, smpl_forward_train, smpl_backward      # noqa: F401
from .maf_autograd import MAFSampleFn      # noqa: F401
from .heads_autograd import LinearFn, ConvNHWCFn, DownsampleFn      # noqa: F401
from .whmr_train import whmr_forward_train, regressor_train, tz_head_train      # noqa: F401
from .graph_step import capture_train_step      # noqa: F401
