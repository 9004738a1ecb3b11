from .vit_autograd import ViTFn, vit_forward_train, vit_backward      # noqa: F401
