from . import image_loss, gradient_loss, semantic_loss  # noqa: F401
