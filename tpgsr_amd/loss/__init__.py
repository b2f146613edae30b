from . import image_loss, gradient_loss  # noqa: F401
