from .unet_discriminators import ResBlock, UNetDiscriminator  # noqa: F401
